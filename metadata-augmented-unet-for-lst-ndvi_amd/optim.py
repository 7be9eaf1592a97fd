"""``mau_amd.AdamW``: ``torch.optim.AdamW`` (the reference's optimizer, src/train.py:213-214; conf/config.yaml:41,48,52) with the
update of every 3x3 convolution weight of a network and the re-pack of the updated weights into the matrix-core layouts done by
ONE kernel (``mau_adamw_pack_step``, csrc/optim.hip).

Same constructor arguments, same update rule, same ``state_dict`` layout (``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}``:
an ``optimizer_state_dict`` written here loads into ``torch.optim.AdamW`` and vice versa).  What differs is the traffic:
torch's fused AdamW streams p, g, m, v (28 bytes per parameter), then the forward of the next step re-reads every weight twice to
build its two packs; here a workgroup owns a 64 x 64 x 9 block of a layer, applies AdamW and writes both packs from LDS
(32 bytes per parameter, one launch: 0.37 -> 0.2 ms per step of the U-Net).  The weight gradients are produced straight into a
flat arena (``functional.ConvBNReLU.backward`` writes the split-K sum into the parameter's ``_mau_grad_slot``; autograd adopts
that view as ``p.grad``), so the kernel's table of addresses is built once.  All other parameters (BatchNorm, biases, encoders,
head: 0.03 % of the model) go through torch's fused multi-tensor AdamW kernel.  The step count lives on the device: the step is
capturable into a hipGraph (``train_graph.GraphedTrainStep``).
"""
from __future__ import annotations

import ctypes
from typing import List

import torch

from . import functional as F_
from ._lib import call, lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("AdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self._arena = None

    # ------------------------------------------------------------------ #
    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)      # on the device: capturable
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not st["step"].is_cuda and p.is_cuda:                                  # a state_dict written by torch's AdamW
            st["step"] = st["step"].to(p.device, dtype=torch.float32)
        return st

    def _conv_params(self, group) -> List[torch.Tensor]:
        return [p for p in group["params"] if p.is_cuda and p.dim() == 4 and p.shape[2:] == (3, 3) and p.dtype == torch.float32
                and getattr(p, "_mau_group", None) is not None and p.is_contiguous()]

    def _ensure_slots(self, convs):
        """Stable gradient addresses: every convolution weight gets a slot of a flat arena (unless dist.GradSync already gave it
        one); the backward writes the weight gradient there and autograd adopts the view as p.grad."""
        missing = [p for p in convs if getattr(p, "_mau_grad_slot", None) is None]
        if missing:
            self._arena = torch.zeros(sum(p.numel() for p in missing), dtype=torch.float32, device=missing[0].device)
            off = 0
            for p in missing:
                p._mau_grad_slot = self._arena[off:off + p.numel()].view_as(p)
                off += p.numel()

    def _table(self, convs, pg, code):
        states = [self.state[p] for p in convs]
        st = pg._state.get(code) if pg is not None and code is not None else None
        key = (tuple(p.data_ptr() for p in convs), tuple(p._mau_grad_slot.data_ptr() for p in convs),
               tuple(s["exp_avg"].data_ptr() for s in states), code, None if st is None else tuple(t.data_ptr() for t in st["wf"]))
        tb = self._tables.get(id(pg))
        if tb is None or tb["key"] != key:
            nbytes = lib.mau_adamw_pack_desc_bytes()
            host = ctypes.create_string_buffer(nbytes * len(convs))
            nxt = ctypes.c_int(0)
            index = {id(p): i for i, p in enumerate(pg.params)} if st is not None else {}
            for i, (p, s) in enumerate(zip(convs, states)):
                j = index.get(id(p))
                wf = st["wf"][j].data_ptr() if j is not None else None
                wd = st["wd"][j].data_ptr() if j is not None else None
                call("mau_adamw_pack_desc_fill", ctypes.addressof(host), i, p.data_ptr(), p._mau_grad_slot.data_ptr(), s["exp_avg"].data_ptr(),
                     s["exp_avg_sq"].data_ptr(), wf, wd, p.shape[0], p.shape[1], nxt.value, ctypes.addressof(nxt))
            tb = self._tables[id(pg)] = {"key": key, "table": torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(convs[0].device),
                                         "tiles": nxt.value, "packs": st is not None}
        return tb

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = float(group["lr"]), group["betas"], group["eps"], group["weight_decay"]
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            for p in live:
                if p.grad.is_sparse:
                    raise RuntimeError("AdamW does not support sparse gradients")
                self._init_state(p)
            torch._foreach_add_([self.state[p]["step"] for p in live], 1.0)
            convs = [p for p in self._conv_params(group) if p.grad is not None]
            fused_ids = set()
            by_group = {}
            for p in convs:
                by_group.setdefault(id(p._mau_group), (p._mau_group, []))[1].append(p)
            for pg, ps in by_group.values():
                self._ensure_slots(ps)
                stray = [p for p in ps if p.grad.data_ptr() != p._mau_grad_slot.data_ptr()]
                if stray:                                   # (a gradient that did not come through the arena: accumulated, cloned ...)
                    torch._foreach_copy_([p._mau_grad_slot for p in stray], [p.grad for p in stray])
                # the packs of the ONE precision the network has been run in ride along (several: the next forward re-packs)
                codes = list(pg._state.keys())
                code = codes[0] if len(codes) == 1 and len(ps) == len(pg.params) else None
                tb = self._table(ps, pg, code)
                call("mau_adamw_pack_step", tb["table"].data_ptr(), len(ps), tb["tiles"], code if code is not None else F_.MAU_F32,
                     self.state[ps[0]]["step"].data_ptr(), lr, b1, b2, eps, wd, F_._stream())
                if tb["packs"]:
                    pg.fresh_after_step = F_._GENERATION[0] + 1      # (the global step post-hook bumps the generation once, after this returns)
                fused_ids.update(id(p) for p in ps)
            rest = [p for p in live if id(p) not in fused_ids]
            if rest:
                torch._fused_adamw_(rest, [p.grad for p in rest], [self.state[p]["exp_avg"] for p in rest],
                                    [self.state[p]["exp_avg_sq"] for p in rest], [], [self.state[p]["step"] for p in rest],
                                    lr=lr, beta1=b1, beta2=b2, weight_decay=wd, eps=eps, amsgrad=False, maximize=False)
        return loss
