"""Data-parallel training support: one process per GPU, RCCL over xGMI ("gloo" in the CPU tests).

The reference has no multi-GPU code at all; this is new behaviour with reference
semantics: N ranks x per-rank batch b with synchronised BatchNorm statistics and averaged gradients
equals the reference on one device with batch N*b.

How the collectives are issued (round 4).  ``torch.distributed`` is the rendezvous (ranks, store, barrier); the
data-path collectives are **direct ``ncclAllReduce`` calls into librccl on the stream we name** (``RcclComm``, a ctypes
binding of the RCCL that torch itself ships), not ``ProcessGroupNCCL`` calls.  ProcessGroupNCCL runs every collective on
ITS OWN stream: each call is an event hand-over there and back (the waiting kernel sits behind two events), a Work object, a
watchdog entry -- 30-40 us of host time and two stream switches, 36 times per step for the latency-bound BatchNorm messages;
and all collectives of a group are serialised on that one stream, so a 0.5-16 KB BatchNorm message that the backward
pass is waiting for queues behind a 32 MiB gradient bucket issued before it.  Here:
  * **SyncBN** (18 forward + 18 backward messages of [sums | pixel count], 2C+1 fp64): ``ncclAllReduce`` on the COMPUTE
    stream itself, between the kernel that produces the sums and the one that consumes them -- stream order is the only
    synchronisation, no events, and under hipGraph capture the collective is just one more kernel node.  (The backward
    sums cannot ride with the gradient buckets, as round 3's notes hoped: BatchNorm's backward needs the GLOBAL
    sum(dz), sum(dz*xhat) before it can form dz of that layer -- the very next kernel.)
  * **gradients**: all parameters' gradients live in ONE flat fp32 arena cut into ~32 MiB buckets in reverse registration
    order (= backward arrival order); when a bucket's last gradient arrives its all-reduce (``ncclAvg``) is issued on a
    COMMUNICATION stream through a SECOND communicator: that stream waits (events) for the streams that produced the
    bucket's gradients -- the compute stream and the weight-gradient side stream -- and the compute stream never waits for
    anything until ``finish()``, one join per step.  The weight-gradient side stream therefore stays ON under data parallelism
    (round 3 switched it off: every bucket launch made the main stream wait for it).  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): few large messages keep every link busy.  The convolution weight gradients
    (99.9 % of the bytes) are WRITTEN INTO the arena by the kernel that produces them; the few small tensors enter
    through one multi-tensor copy per bucket.
Under the "gloo" backend (CPU tests; several ranks sharing one GPU in the rehearsals) there is no RCCL communicator:
the same messages go through ``torch.distributed`` (device tensors staged through host memory) -- correctness rehearsal,
never the measured path.

Parameters that receive no gradient (``temporal_encoder.*`` when ``temporal_embeddings=False``,
SURVEY D4) keep ``grad is None`` exactly as in the reference; their arena slots stay zero.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

# --------------------------------------------------------------------------- #
# librccl, called directly (rccl.h: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy)
# --------------------------------------------------------------------------- #
NCCL_SUM, NCCL_AVG = 0, 4                      # ncclRedOp_t
_NCCL_DTYPE = {torch.float32: 7, torch.float64: 8, torch.bfloat16: 9, torch.float16: 6, torch.int32: 2, torch.int64: 4}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]              # NCCL_UNIQUE_ID_BYTES


_RCCL = [None]


def _rccl():
    """The RCCL torch was built with (``torch/lib/librccl.so``: already mapped into the process, so this adds no second
    copy of the library), else the ROCm installation's."""
    if _RCCL[0] is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"]
        err = None
        for path in cands:
            try:
                lib = C.CDLL(path)
                break
            except OSError as e:
                err = e
        else:
            raise ImportError(f"librccl not found ({err}); data-parallel runs over RCCL need it")
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommAbort.argtypes = [C.c_void_p]
        _RCCL[0] = lib
    return _RCCL[0]


def _nccl_check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"RCCL {what} failed: {_rccl().ncclGetErrorString(status).decode()} (ncclResult_t {status})")


class RcclComm:
    """One RCCL communicator over the ranks of a ``torch.distributed`` group (which is used ONLY to hand rank 0's
    ``ncclUniqueId`` to the others).  ``all_reduce`` enqueues ``ncclAllReduce`` on the stream it is given and returns:
    ordering against kernels is stream order, exactly like a kernel launch through the C ABI.  Collectives of ONE communicator
    must be issued in the same order on every rank and on one stream at a time -- SyncBN (compute stream) and the gradient
    buckets (communication stream) therefore own a communicator each."""

    def __init__(self, group=None, device: Optional[torch.device] = None):
        group = group if group is not None else dist.group.WORLD
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        lib = _rccl()
        uid = _UniqueId()
        if self.rank == 0:
            _nccl_check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            box = [bytes(uid.internal)] if self.rank == 0 else [None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group,
                                       device=self.device if dist.get_backend(group) == "nccl" else None)
            C.memmove(C.byref(uid), box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _nccl_check(lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._fin = weakref.finalize(self, RcclComm._destroy, self._comm.value)

    @staticmethod
    def _destroy(handle):
        try:
            _rccl().ncclCommDestroy(C.c_void_p(handle))
        except Exception:            # interpreter shutdown: the runtime may already be gone
            pass

    def all_reduce(self, t: torch.Tensor, op: int = NCCL_SUM, stream: Optional[int] = None):
        """In-place all-reduce of a contiguous device tensor on ``stream`` (a raw ``hipStream_t``; default: torch's current)."""
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("RcclComm.all_reduce needs a contiguous device tensor")
        if stream is None:
            stream = torch.cuda.current_stream(t.device).cuda_stream
        _nccl_check(_rccl().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_DTYPE[t.dtype], op, self._comm, stream),
                    "ncclAllReduce")

    def destroy(self):
        self._fin()

    def abort(self):
        """``ncclCommAbort``: ends a collective that will never complete (a dead or mis-ordered peer) so that the process can
        leave; the communicator is unusable afterwards."""
        if self._fin.detach() is not None:
            try:
                _rccl().ncclCommAbort(self._comm)
            except Exception:
                pass


# (id(group), tag, device) -> (group, communicator).  The entry holds the group OBJECT: while it is here CPython cannot hand its
# id to another group, and a lookup checks identity -- a communicator built over a destroyed group is never returned for a new one.
_COMMS = {}


def direct_rccl(direct: Optional[bool] = None) -> bool:
    """Whether the data-path collectives are issued as direct RCCL calls: the caller's explicit choice, else ``MAU_RCCL_DIRECT``
    (1 / 0), else yes."""
    if direct is not None:
        return bool(direct)
    return os.environ.get("MAU_RCCL_DIRECT", "1") != "0"


def rccl_comm(group, tag: str, direct: Optional[bool] = None) -> Optional[RcclComm]:
    """The process's communicator ``tag`` ("bn" / "grad") over ``group``, created on first use by EVERY rank of the group
    (creation is itself a collective); ``None`` when the group does not run on RCCL (gloo rehearsals, CPU tests) or the direct
    path is off (``direct=False`` / ``MAU_RCCL_DIRECT=0``: everything through ProcessGroupNCCL, as in round 3)."""
    if group is None or not torch.cuda.is_available() or dist.get_backend(group) != "nccl":
        return None
    if not direct_rccl(direct):
        return None
    key = (id(group), tag, torch.cuda.current_device())
    ent = _COMMS.get(key)
    if ent is None or ent[0] is not group:
        ent = _COMMS[key] = (group, RcclComm(group))
    return ent[1]


def destroy_comms(abort: bool = False):
    """Destroy (or abort) every directly-driven communicator of the process and forget them.  Call it BEFORE
    ``dist.destroy_process_group()``: the communicators were built over the group's ranks and must not outlive it."""
    for _, c in list(_COMMS.values()):
        c.abort() if abort else c.destroy()
    _COMMS.clear()


class CollectiveWatchdog:
    """A directly-issued ``ncclAllReduce`` has no timeout of its own (ProcessGroupNCCL's watchdog never sees it): a peer that
    died, or two communicators' kernels started in different orders on two ranks, is a silent hang.  The training loop calls
    ``kick()`` once per finished step (after a host read-back, so "finished" means the GPU was there) and once per validation batch;
    checkpoint writing runs under ``paused()``.  When no kick arrives for ``timeout_s`` the watchdog thread says so, aborts the
    communicators (``ncclCommAbort``, in a helper thread bounded by ``MAU_DIST_ABORT_TIMEOUT_S``: the abort may block too) and ends
    the process with a non-zero code -- the launcher then ends the other ranks.  ``close()`` (or leaving the ``with`` block) ends the
    thread: a caller that catches an exception of the loop must not be killed by a stale watchdog later."""

    def __init__(self, timeout_s: Optional[float] = None, exit_code: int = 86, _exit=os._exit):
        import threading
        import time
        self.timeout_s = float(os.environ.get("MAU_DIST_TIMEOUT_S", "600")) if timeout_s is None else float(timeout_s)
        self._exit, self._code, self._time = _exit, exit_code, time
        self.abort_timeout_s = float(os.environ.get("MAU_DIST_ABORT_TIMEOUT_S", "20"))
        self._last = time.monotonic()
        self._paused = 0
        self._stop = threading.Event()
        self.fired = False
        self._thread = threading.Thread(target=self._watch, daemon=True, name="mau-collective-watchdog")
        self._thread.start()

    def kick(self):
        self._last = self._time.monotonic()

    def paused(self):
        """Context manager for stretches that are not training steps and issue no collective of ours to wait for -- a validation pass,
        checkpoint writing, the first fetch of an epoch: the clock stands still inside and restarts on exit."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            self._paused += 1
            try:
                yield self
            finally:
                self._paused -= 1
                self.kick()
        return _cm()

    def _watch(self):
        while not self._stop.wait(min(1.0, self.timeout_s / 4)):
            if self._paused > 0:
                self._last = self._time.monotonic()
                continue
            if self._time.monotonic() - self._last > self.timeout_s:
                import sys
                import threading
                self.fired = True
                print(f"mau_amd.dist: no training step finished for {self.timeout_s:.0f} s -- a collective is hung; aborting the RCCL "
                      f"communicators and leaving with code {self._code}", file=sys.stderr, flush=True)
                # ncclCommAbort can itself block (a peer that never answers): it runs in a helper thread under a bound, the exit does not wait for it
                t = threading.Thread(target=destroy_comms, kwargs=dict(abort=True), daemon=True, name="mau-comm-abort")
                t.start()
                t.join(self.abort_timeout_s)
                self._exit(self._code)
                return

    def close(self):
        self._stop.set()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


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
    __slots__ = ("lo", "hi", "params", "pending", "handle", "launched", "streams", "side_streams")

    def __init__(self):
        self.lo = self.hi = 0
        self.params: List[torch.nn.Parameter] = []
        self.pending = 0
        self.handle = None
        self.launched = False
        self.streams = []            # streams on which autograd accumulated a gradient of the bucket (the backward nodes' streams)
        self.side_streams = []       # weight-gradient streams still WRITING arena slots of the bucket (functional: deferred join)


class GradSync:
    """Bucketed, backward-overlapped gradient averaging over a process group.

    Usage per step (mirrors src/train.py:252-256 with the collective inserted):
        sync.begin(); loss.backward(); sync.finish(); optimizer.step(); optimizer.zero_grad()
    """

    def __init__(self, module: torch.nn.Module, group=None, bucket_bytes: int = 32 << 20, direct: Optional[bool] = None):
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
        # RCCL called directly on a communication stream of our own (module docstring); None = through torch.distributed
        self.comm = rccl_comm(self.group, "grad", direct) if dev.type == "cuda" else None
        self.comm_stream = torch.cuda.Stream(device=dev) if self.comm is not None else None
        self._comm_used = False
        self._dirty = set()           # ids of parameters whose arena slot has held a gradient (the arena starts zeroed)

    # ------------------------------------------------------------------ #
    def begin(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.handle = None
            b.launched = False
            b.streams = []
            b.side_streams = []
        self._active = True
        self._comm_used = False

    def _launch(self, b: _Bucket):
        """All gradients of the bucket have arrived: move the small ones into the arena with ONE multi-tensor copy (the
        convolution weights' are born there), re-point ``p.grad`` at the arena slots, start the bucket's all-reduce."""
        b.launched = True
        cur = torch.cuda.current_stream() if self.flat.is_cuda else None
        # gradients accumulated on other streams (the TemporalEncoder's backward runs on its side stream): the launching stream
        # waits for those before the copy reads them
        for s in b.streams:
            if s != cur:
                cur.wait_stream(s)
        if self.comm is None:                          # torch.distributed issues the collective behind the launching stream:
            for s in b.side_streams:                   # that stream must have seen the weight-gradient streams' writes too
                if s != cur:
                    cur.wait_stream(s)
        have = [p for p in b.params if p.grad is not None and p.grad.data_ptr() != p._mau_grad_slot.data_ptr()]
        if have:
            views = [p._mau_grad_slot for p in have]
            torch._foreach_copy_(views, [p.grad for p in have])
            for p, v in zip(have, views):
                p.grad = v
        if self.group is None:
            return
        seg = self.flat[b.lo:b.hi]
        if self.comm is not None:
            # the COMMUNICATION stream waits for the launching stream (which has joined every accumulation stream and ran the
            # copy) and for the weight-gradient streams still writing slots of this bucket; the compute stream just goes on
            cs = self.comm_stream
            cs.wait_stream(cur)
            for s in b.side_streams:
                if s != cur:
                    cs.wait_stream(s)
            self.comm.all_reduce(seg, NCCL_AVG, cs.cuda_stream)
            self._comm_used = True
        elif self._avg:                                # ProcessGroupNCCL (MAU_RCCL_DIRECT=0): averages in the collective itself
            b.handle = dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:                                          # gloo has no AVG: pre-scale, then SUM
            seg.mul_(1.0 / self.world)
            b.handle = all_reduce_sum(seg, self.group, async_op=True)

    def _on_grad(self, p: torch.nn.Parameter):
        if not self._active:
            return
        b = self._slot[id(p)][2]
        self._dirty.add(id(p))
        if p.is_cuda:
            s = torch.cuda.current_stream(p.device)
            if s not in b.streams:
                b.streams.append(s)
            ws = getattr(p, "_mau_grad_stream", None)       # the gradient is still being written on the weight-gradient stream
            if ws is not None and ws not in b.side_streams:
                b.side_streams.append(ws)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def finish(self):
        """Launch what is left (buckets holding never-used parameters), then make the caller's stream wait for everything."""
        if not self._active:
            return
        for b in self.buckets:
            if not b.launched:
                # slots of parameters without a gradient this step must not carry stale values (a slot that never held a
                # gradient is still zero: no fill kernel per step for the parameters that are never used)
                for p in b.params:
                    if p.grad is None and id(p) in self._dirty:
                        lo, hi, _ = self._slot[id(p)]
                        self.flat[lo:hi].zero_()
                        self._dirty.discard(id(p))
                self._launch(b)
        if self._comm_used:                           # ONE join per step: the optimizer reads the averaged arena
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
        self._active = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
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
