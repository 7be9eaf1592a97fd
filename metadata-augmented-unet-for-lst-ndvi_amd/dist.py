"""Data-parallel training support: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU code at all; this is new behaviour with reference
semantics: N ranks x per-rank batch b with synchronised BatchNorm statistics and averaged gradients
equals the reference on one device with batch N*b.

Two collectives per step family:
  * gradients: all parameters' gradients live in ONE flat fp32 arena; it is cut into buckets in
    reverse registration order (= backward arrival order) and each bucket is all-reduced
    asynchronously as soon as its last gradient has arrived, overlapping the rest of backward.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages (default 32 MiB) keep
    every link busy; tiny per-tensor messages would be latency-bound.  The convolution weight
    gradients (99.9 % of the bytes) are WRITTEN INTO the arena by the kernel that produces them
    (``functional.ConvBNReLU.backward`` unpacks the split-K sum straight into the parameter's arena slot and
    autograd adopts that view as ``p.grad``: no copy); the few small tensors left enter through one
    multi-tensor copy per bucket.  The collective averages (ReduceOp.AVG on RCCL).
  * BatchNorm: per layer one all-reduce of [sum | sum of squares] (2C fp64) in forward and one of
    [sum dz | sum dz*xhat] in backward -- see ``functional.ConvBNReLU``; enabled by
    ``model.set_sync_bn(group)``.

Parameters that receive no gradient (``temporal_encoder.*`` when ``temporal_embeddings=False``,
SURVEY D4) keep ``grad is None`` exactly as in the reference; their arena slots stay zero.
"""
from __future__ import annotations

import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def all_reduce_sum(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of ``t`` in place.  RCCL ("nccl") reduces device tensors directly; under the
    "gloo" rehearsal backend (CPU tests, or several ranks sharing one GPU) a device tensor is staged
    through host memory -- correctness rehearsal only, never the measured path."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
        return None
    return dist.all_reduce(t, group=group, async_op=async_op)


class _Bucket:
    __slots__ = ("lo", "hi", "params", "pending", "handle", "launched", "streams")

    def __init__(self):
        self.lo = self.hi = 0
        self.params: List[torch.nn.Parameter] = []
        self.pending = 0
        self.handle = None
        self.launched = False
        self.streams = []


def _grad_sync_gone():
    from . import functional as F_
    F_._GRAD_SYNC_ACTIVE[0] -= 1


class GradSync:
    """Bucketed, backward-overlapped gradient averaging over a process group.

    Usage per step (mirrors src/train.py:252-256 with the collective inserted):
        sync.begin(); loss.backward(); sync.finish(); optimizer.step(); optimizer.zero_grad()
    """

    def __init__(self, module: torch.nn.Module, group=None, bucket_bytes: int = 32 << 20):
        self.group = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
        self.world = dist.get_world_size(self.group) if self.group is not None else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("GradSync: module has no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._slot = {}
        self.buckets: List[_Bucket] = []
        # reverse registration order ~ order in which autograd produces the gradients
        off = total
        cur = _Bucket()
        cur.hi = total
        for p in reversed(self.params):
            off -= p.numel()
            self._slot[id(p)] = (off, off + p.numel(), cur)
            cur.params.append(p)
            cur.lo = off
            if (cur.hi - cur.lo) * 4 >= bucket_bytes:
                self.buckets.append(cur)
                cur = _Bucket()
                cur.hi = off
        if cur.params:
            self.buckets.append(cur)
        # the arena slot of every parameter, as a tensor of the parameter's shape: a kernel that produces the gradient may write
        # it there directly (functional.ConvBNReLU.backward does, for the convolution weights)
        for p in self.params:
            lo, hi, _ = self._slot[id(p)]
            p._mau_grad_slot = self.flat[lo:hi].view_as(p)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._active = False
        self._avg = self.group is not None and dist.get_backend(self.group) == "nccl"
        self._counted = None
        if self.group is not None:
            from . import functional as F_
            F_._GRAD_SYNC_ACTIVE[0] += 1          # (the weight-gradient side stream stays off while buckets are launched inside backward)
            self._counted = weakref.finalize(self, _grad_sync_gone)      # remove() or garbage collection, whichever comes first

    # ------------------------------------------------------------------ #
    def begin(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.handle = None
            b.launched = False
            b.streams = []
        self._active = True

    def _launch(self, b: _Bucket):
        """All gradients of the bucket have arrived: move them into the arena with ONE multi-tensor copy (instead of a
        small kernel per parameter), re-point ``p.grad`` at the arena slots, start the bucket's all-reduce."""
        b.launched = True
        if b.streams:
            # gradients of this bucket were produced on other streams too (the TemporalEncoder's backward runs on its side
            # stream): the launching stream waits for them before the copy / the collective read the arena
            cur = torch.cuda.current_stream()
            for s in b.streams:
                if s != cur:
                    cur.wait_stream(s)
        have = [p for p in b.params if p.grad is not None and p.grad.data_ptr() != p._mau_grad_slot.data_ptr()]
        if have:
            views = [p._mau_grad_slot for p in have]
            torch._foreach_copy_(views, [p.grad for p in have])
            for p, v in zip(have, views):
                p.grad = v
        if self.group is not None:
            seg = self.flat[b.lo:b.hi]
            if self._avg:                                  # RCCL averages in the collective itself
                b.handle = dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:                                          # gloo has no AVG: pre-scale, then SUM
                seg.mul_(1.0 / self.world)
                b.handle = all_reduce_sum(seg, self.group, async_op=True)

    def _on_grad(self, p: torch.nn.Parameter):
        if not self._active:
            return
        b = self._slot[id(p)][2]
        if p.is_cuda:
            s = torch.cuda.current_stream(p.device)
            if s not in b.streams:
                b.streams.append(s)
            ws = getattr(p, "_mau_grad_stream", None)       # the gradient is still being written on the weight-gradient stream
            if ws is not None and ws not in b.streams:
                b.streams.append(ws)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def finish(self):
        """Launch what is left (buckets holding never-used parameters), then wait for everything."""
        if not self._active:
            return
        for b in self.buckets:
            if not b.launched:
                # slots of parameters without a gradient this step must not carry stale values
                for p in b.params:
                    if p.grad is None:
                        lo, hi, _ = self._slot[id(p)]
                        self.flat[lo:hi].zero_()
                self._launch(b)
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
        self._active = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._counted is not None:
            self._counted()                       # (a finalizer runs once)
            self._counted = None
        for p in self.params:
            if hasattr(p, "_mau_grad_slot"):
                del p._mau_grad_slot


def init_process_group_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("MAU_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())      # rehearsal: several ranks may share one GPU
    return rank, local, world
