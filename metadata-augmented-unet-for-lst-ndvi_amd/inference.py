"""Inference session for the app / evaluation callers.

The reference's callers run ``model.eval()`` + ``torch.no_grad()`` forwards on fixed shapes
(``app/model_utils.py:102-109``: one 1x23x512x512 tile per click; ``test/metadata_sensitivity.py:294-311``:
B=50 sweeps).  At B=1 the forward is ~45 short kernels and host launch overhead dominates, so the
whole forward is captured ONCE into a hipGraph on a side stream and replayed per call (the C ABI does
no allocation or synchronisation, which is what makes it capturable).  Same kernels, same results.
"""
from __future__ import annotations

import torch


class GraphedInference:
    """``GraphedInference(model, maps, temp_series, metadata)(maps, temp_series, metadata) -> output``.

    The example inputs fix shapes/dtypes; later calls may pass different VALUES of the same shapes.
    The session freezes the model (``freeze_inference``): packed weights and folded BatchNorm coefficients are
    computed once during warm-up and the captured graph holds only the per-call kernels, so
    reload weights -> build a new session (cheap: a few forwards).
    """

    def __init__(self, model: torch.nn.Module, maps: torch.Tensor, temp_series: torch.Tensor, metadata: torch.Tensor,
                 warmup: int = 2, clone_output: bool = True):
        if not maps.is_cuda:
            raise RuntimeError("GraphedInference needs inputs on the MI355X ('cuda') device; there is no CPU fallback")
        self.model = model.eval()
        self.clone_output = clone_output                 # False: __call__ returns the session's own output buffer, valid until the next call
        if hasattr(self.model, "freeze_inference"):
            self.model.freeze_inference(True)           # packed weights / folded BN coefficients computed once, outside the graph
        self._in = [maps.detach().clone(), temp_series.detach().clone(), metadata.detach().clone()]
        side = torch.cuda.Stream(device=maps.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):            # packs weights, sets kernel attributes, warms the allocator
                self.model(*self._in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._out = self.model(*self._in)

    @property
    def inputs(self):
        """The session's own (maps, temp_series, metadata) device buffers: a caller that fills THESE (``sess.inputs[0].copy_(host_tile)``)
        and passes them to ``__call__`` pays no device-to-device copy at all."""
        return tuple(self._in)

    @torch.no_grad()
    def __call__(self, maps: torch.Tensor, temp_series: torch.Tensor, metadata: torch.Tensor) -> torch.Tensor:
        srcs = (maps, temp_series, metadata)
        for dst, src in zip(self._in, srcs):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedInference was captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
        todo = [(d, s) for d, s in zip(self._in, srcs) if not (s.is_cuda and s.data_ptr() == d.data_ptr())]
        if len(todo) > 1 and all(s.is_cuda and s.dtype == d.dtype and s.device == d.device for d, s in todo):
            # ONE multi-tensor launch for the three small inputs: at B = 1 each separate copy costs ~6 us of stream time, 2 % of the
            # 0.7 ms forward (the copies were 3 of the session's launches)
            torch._foreach_copy_([d for d, _ in todo], [s for _, s in todo], non_blocking=True)
        else:
            for d, s in todo:                                # (host tensors, or a dtype conversion: plain copies)
                d.copy_(s, non_blocking=True)
        self.graph.replay()
        return self._out.clone() if self.clone_output else self._out
