"""The inner training step of the reference (src/train.py:243-256) as ONE hipGraph.

    step = GraphedTrainStep(model, optimizer, criterion)
    for batch in loader:
        loss = step(maps, temp_series, metadata, targets)        # forward + criterion + backward + optimizer.step

A train step of the U-Net is ~190 kernel launches of 2 us .. 400 us; launched one by one from Python the
short ones (weight packs, slab reductions, BatchNorm finalizes, split-K sums) leave the GPU waiting for
the host.  The C ABI allocates nothing and never synchronises, the optimizer (mau_amd.AdamW, or torch's fused
AdamW made capturable) reads its step count from device memory, and every reduction keeps its workspace in torch's caching allocator -- so the whole step is
capturable: the first ``warmup`` calls run eagerly (they are ordinary training steps on the batches they
are given; they also warm the allocator and set the kernels' LDS attributes), the next call captures
forward, criterion, backward and the optimizer step into a graph and replays it, every later call only
refreshes the static input buffers and replays.  Results are bit-identical to the eager step
(``tests/test_gpu_model.py::test_graphed_train_step_matches_eager``).

Data parallel: with ``grad_sync=dist.GradSync(...)`` the bucketed gradient all-reduces and the model's SyncBN all-reduces are
captured too (RCCL kernels become graph nodes; torch's ProcessGroupNCCL joins its internal stream to the capture).  Measured on one
GPU under a 1-rank RCCL group: eager data-parallel step 14.4 ms, captured 13.5 ms, plain captured step 13.0 ms (same box) -- the
data-parallel path's extra cost is mostly host work that the graph removes.  ``bench.py`` and ``train.run`` keep the data-parallel
step EAGER unless ``MAU_DP_GRAPH=1``: a capture that goes wrong at N > 1 cannot be rehearsed on the one-GPU boxes this was built on,
and a hung rank costs the whole measurement.  Not captured: anything whose shapes change between calls.  A failed capture raises -- it is never retried or silently replaced.
"""
from __future__ import annotations

import gc
from typing import Callable, Optional, Sequence

import torch

from . import functional as F_
from .optim import AdamW as _AdamW


def _make_capturable(optimizer: torch.optim.Optimizer):
    """Adam-family optimizers keep their step count on the host unless ``capturable``: switch the flag and move counts
    that already exist (a resumed optimizer) to the parameters' device.  Optimizers without the flag (SGD) need nothing."""
    for group in optimizer.param_groups:
        if "capturable" in group and not group["capturable"]:
            group["capturable"] = True
            for p in group["params"]:
                st = optimizer.state.get(p)
                if st and torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                    st["step"] = st["step"].to(p.device, dtype=torch.float32)


def _grad_accumulator(p: torch.Tensor):
    with torch.enable_grad():
        return p.view_as(p).grad_fn.next_functions[0][0]


def live_autograd_graph_params(params) -> list:
    """Indices of the parameters whose AccumulateGrad node is being KEPT ALIVE by something -- an autograd graph of an earlier
    step that still exists (a kept ``loss`` / ``outputs`` with grad_fn, a reference cycle).  A parameter owns its accumulator only
    weakly: with no graph alive the node dies as soon as the probe lets go of it and the second look finds a fresh one; a node
    that survives (its ``metadata`` dict, which lives with the C++ node, still carries the mark) is held by a graph."""
    params = [p for p in params if p.requires_grad and p.is_leaf]
    mark = object()
    for p in params:
        _grad_accumulator(p).metadata["mau_capture_probe"] = mark
    held = []
    for i, p in enumerate(params):
        node = _grad_accumulator(p)
        if node.metadata.get("mau_capture_probe") is mark:
            del node.metadata["mau_capture_probe"]
            held.append(i)
    return held


class GraphedTrainStep:
    """``criterion(outputs, targets)`` returns the reference's loss dict (``{'total': ...}``, src/utils/losses.py) or a
    scalar tensor.  ``clip_grad_norm``: max norm of ``torch.nn.utils.clip_grad_norm_`` (src/train.py:253-254), 0 = off.
    ``copy_inputs=False``: the tensors of the capturing call ARE the static buffers -- later calls must pass the same
    tensors (a resident synthetic batch, or buffers the loader fills in place); the default copies every batch in."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, criterion: Callable, warmup: int = 3,
                 clip_grad_norm: float = 0.0, copy_inputs: bool = True, grad_sync=None):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.warmup = max(1, int(warmup))
        self.clip = float(clip_grad_norm)
        self.copy_inputs = copy_inputs
        self.grad_sync = grad_sync                 # dist.GradSync: its bucketed all-reduces (and the model's SyncBN ones) become graph nodes
        self.calls = 0
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._in: Optional[Sequence[torch.Tensor]] = None
        self.loss: Optional[torch.Tensor] = None
        self.outputs: Optional[torch.Tensor] = None
        self._groups, self._self_packing, self._expect_gen = [], False, -1
        _make_capturable(optimizer)

    # ------------------------------------------------------------------ #
    def _loss_of(self, outputs, targets):
        losses = self.criterion(outputs, targets)
        return losses.get("total") if isinstance(losses, dict) else losses

    def _eager(self, maps, temp_series, metadata, targets):
        outputs = self.model(maps, temp_series, metadata)                      # src/train.py:245
        loss = self._loss_of(outputs, targets)                                 # :247-249
        if self.grad_sync is not None:
            self.grad_sync.begin()
        loss.backward()                                                        # :252
        if self.grad_sync is not None:
            self.grad_sync.finish()
        if self.clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)  # :253-254
        self.optimizer.step()                                                  # :255
        self.optimizer.zero_grad(set_to_none=True)                             # :256
        return loss.detach(), outputs.detach()

    def _capture(self, batch):
        if self.copy_inputs:
            self._in = [t.detach().clone() for t in batch]
        else:
            self._in = list(batch)
        self.optimizer.zero_grad(set_to_none=True)        # the captured backward ASSIGNS the gradients (static buffers of the graph)
        # An autograd graph of an earlier step that is still alive (a reference cycle in user code, a kept loss) keeps its
        # AccumulateGrad nodes, bound to the stream they were created on; the captured backward would re-use them and make the engine
        # synchronise with that stream -- on ROCm 7.2 the capture then dies in hipStreamEndCapture instead of raising.
        gc.collect()
        held = live_autograd_graph_params(self.model.parameters())
        if held:
            raise RuntimeError(
                f"GraphedTrainStep: an autograd graph of an earlier step is still alive (it holds the AccumulateGrad nodes of {len(held)} "
                "parameters, created on another stream) -- typically a kept `loss` / `outputs` tensor that still has a grad_fn, or a "
                "reference cycle through an autograd ctx.  Capturing now would make the engine synchronise the capture with that "
                "stream and end in hipStreamEndCapture taking the process down.  Drop or .detach() those tensors before the "
                "capturing call (the step itself returns detached tensors).")
        # Who re-packs the convolution weights?  torch's optimizers: the captured forward starts with the (captured) multi-tensor
        # re-pack.  optim.AdamW writes the packs together with the update: the captured step then contains NO separate pack -- the
        # forward of replay k reads what the optimizer of replay k - 1 (or of the last warm-up step) wrote.
        self._groups = [m._pack_group for m in self.model.modules() if hasattr(m, "_pack_group")]
        self._self_packing = isinstance(self.optimizer, _AdamW)
        if not self._self_packing:
            F_.mark_params_updated()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # Under a process group another thread of this process talks to the runtime while we capture: ProcessGroupNCCL's watchdog
        # polls the events of earlier collectives (hipEventQuery).  In the default "global" capture mode that call is an error in ANY
        # thread while a capture is open -- the watchdog dies and takes the process with it (seen as a sporadic SIGABRT, depending on
        # whether the last warm-up step's collectives had been reaped yet).  "thread_local" restricts the check to the capturing thread.
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        try:
            with torch.cuda.graph(graph, capture_error_mode=mode):
                outputs = self.model(self._in[0], self._in[1], self._in[2])
                loss = self._loss_of(outputs, self._in[3])
                if self.grad_sync is not None:
                    self.grad_sync.begin()
                loss.backward()
                if self.grad_sync is not None:
                    self.grad_sync.finish()
                if self.clip > 0:
                    torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
                self.optimizer.step()
        except Exception as e:      # the stream / allocator state after a broken capture is not trustworthy: no silent fallback
            raise RuntimeError(f"GraphedTrainStep: capturing the train step failed ({type(e).__name__}: {e}); "
                               "run the step eagerly in a fresh process") from e
        self.graph, self.loss, self.outputs = graph, loss.detach(), outputs.detach()

    # ------------------------------------------------------------------ #
    def __call__(self, maps, temp_series, metadata, targets):
        """One training step on the batch; returns the loss (0-dim device tensor, overwritten by the next call)."""
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            loss, self.outputs = self._eager(maps, temp_series, metadata, targets)
            return loss
        batch = (maps, temp_series, metadata, targets)
        if self.graph is None:
            self._capture(batch)
        elif self.copy_inputs:
            for dst, src in zip(self._in, batch):
                if dst.shape != src.shape:
                    raise ValueError(f"GraphedTrainStep was captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
                dst.copy_(src, non_blocking=True)
        elif any(a.data_ptr() != b.data_ptr() or a.shape != b.shape for a, b in zip(self._in, batch)):
            raise ValueError("GraphedTrainStep(copy_inputs=False): pass the tensors of the captured call (fill them in place)")
        if self._self_packing:
            # the graph holds no pack launch of its own.  If something outside changed parameters since the last replay -- another
            # optimizer's step or mark_params_updated() (generation), load_state_dict / an in-place write (tensor versions) -- the
            # packs are brought up to date before it runs; otherwise ensure() finds them fresh and launches nothing
            for pg in self._groups:
                for code in list(pg._state):
                    pg.ensure(code)
        self.graph.replay()
        # the replay changed every parameter behind Python's back: eager forwards that follow (validation) must re-pack -- unless
        # the captured optimizer already wrote the packs of the new weights
        F_.mark_params_updated()
        if self._self_packing:
            for pg in self._groups:
                pg.fresh_after_step = F_._GENERATION[0]
        self._expect_gen = F_._GENERATION[0]
        return self.loss
