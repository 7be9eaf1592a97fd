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
                 warmup: int = 2):
        if not maps.is_cuda:
            raise RuntimeError("GraphedInference needs inputs on the MI355X ('cuda') device; there is no CPU fallback")
        self.model = model.eval()
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

    @torch.no_grad()
    def __call__(self, maps: torch.Tensor, temp_series: torch.Tensor, metadata: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self._in, (maps, temp_series, metadata)):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedInference was captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self._out.clone()
