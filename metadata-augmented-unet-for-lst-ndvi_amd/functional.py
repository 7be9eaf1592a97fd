"""Autograd functions of the hot path; every forward/backward is a call into libmau_hip.so.

Internal activations are "NHWC-ld" tensors: ``torch`` tensors of shape (N, H, W, ld) with
``ld = roundup(C, 8)``, dtype ``torch.bfloat16`` (throughput mode), ``torch.float16`` (the same
kernels on fp16 operands, inference) or ``torch.float32`` (parity mode); channels [C, ld) are always zero.  PyTorch only provides device
memory, streams and the autograd graph; no arithmetic of the path runs in torch.

Reference: src/model.py (VGGBlock :9-21, pool :218, up/_upsample_match :219,243-246,
fuse_embeddings :248-259, head :284-292), src/utils/losses.py:27-39.
"""
from __future__ import annotations

import contextlib
import dataclasses

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import MAU_BF16, MAU_F16, MAU_F32, call, lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MAU_F32
    if dt == torch.bfloat16:
        return MAU_BF16
    if dt == torch.float16:
        return MAU_F16
    raise TypeError(f"unsupported activation dtype {dt}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on '{t.device}'. This is the MI355X-native HIP path; it has no CPU "
            "fallback. Move the module and its inputs to a 'cuda' (ROCm) device.")


def _ld(t: torch.Tensor) -> int:
    """Pixel stride (elements) of an NHWC-ld tensor or channel-sliced view of one."""
    n, h, w, _ = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != w * ld or (n > 1 and t.stride(0) != h * w * ld) or ld % 8:
        raise RuntimeError(f"not an NHWC-ld tensor: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def _as_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return t if it is NHWC-ld addressable (16-byte aligned base), else a contiguous copy."""
    try:
        _ld(t)
        if t.data_ptr() % 16 == 0:
            return t
    except RuntimeError:
        pass
    return t.contiguous()


@dataclass
class Act:
    """An internal activation: NHWC-ld tensor + its logical channel count.  ``nchw=True``: the network's INPUT as the reference's
    collate_fn delivers it -- (N, C, H, W) fp32 -- handed to the first convolution untouched (``mau_conv3x3_first_fwd`` reads it
    directly: no layout kernel)."""
    t: torch.Tensor
    C: int
    nchw: bool = False

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def H(self):
        return self.t.shape[2 if self.nchw else 1]

    @property
    def W(self):
        return self.t.shape[3 if self.nchw else 2]


# --------------------------------------------------------------------------- #
# weight packing
# --------------------------------------------------------------------------- #
_DT = {MAU_F32: torch.float32, MAU_BF16: torch.bfloat16, MAU_F16: torch.float16}
_GENERATION = [0]


_ZERO_ARENA = {"gen": -1, "buf": None, "cur": 0}


def _zero_bias_grad(n, dev):
    """A zero vector for an identically-zero bias gradient: a slice of one zero-filled arena per optimizer step (one fill
    kernel per step instead of one per layer: 18 launches of ~4 us each in the U-Net).  A region is handed out once and never
    again, so the slices behave like independent tensors (they may become ``param.grad`` and be scaled or accumulated into)."""
    z = _ZERO_ARENA
    need = (n + 63) // 64 * 64
    if z["gen"] != _GENERATION[0] or z["buf"] is None or z["buf"].device != dev or z["cur"] + need > z["buf"].numel():
        z["buf"] = torch.zeros(max(8192, need), dtype=torch.float32, device=dev)
        z["cur"] = 0
        z["gen"] = _GENERATION[0]
    out = z["buf"][z["cur"]:z["cur"] + n]
    z["cur"] += need
    return out


def mark_params_updated(*_args, **_kwargs):
    """Invalidate every cached weight pack.  Registered as a GLOBAL optimizer-step post hook (below), so any
    ``torch.optim`` step -- including the fused AdamW kernel, which updates parameters in place WITHOUT bumping
    ``Tensor._version`` -- is followed by a re-pack on the next forward.  Call it by hand after writing to a
    parameter through ``.data`` or a raw pointer (in-place tensor ops, ``copy_`` and ``load_state_dict`` bump
    ``_version`` and are detected by themselves)."""
    _GENERATION[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook  # noqa: E402

_reg_step_hook(mark_params_updated)


_TICKETS = {}


def _tickets(dev) -> torch.Tensor:
    """Zeroed uint32 ticket buffer of the single-launch reductions (``mau_reduce_rows_*``, ``mau_bn_stats_finalize_train``),
    one per (device, stream): a launch leaves it zeroed, launches on one stream are ordered, and launches that may run
    concurrently on two streams never share one."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream())
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(lib.mau_reduce_tickets_elems(), dtype=torch.int32, device=dev)
    return t


class PackGroup:
    """The 3x3 convolution weights of ONE network: an optimizer step changes all of them, so all forward and data-gradient
    packs are rebuilt by ONE launch (``mau_conv3x3_pack_weights_multi``; 18 launches -> 1 per step in the U-Net) into
    PERSISTENT buffers -- re-packed in place, so the addresses a captured hipGraph or a frozen inference session holds stay
    valid and always see the current weights.  Keyed like the per-parameter cache: (optimizer-step generation, every
    weight's ``_version``, every storage address, dtype)."""

    def __init__(self):
        self.params: List[torch.Tensor] = []
        self._state = {}
        self.fresh_after_step = -1        # optimizer-step generation whose packs optim.AdamW wrote together with the update

    def add(self, w: torch.Tensor):
        if not any(w is q for q in self.params):
            self.params.append(w)
            self._state = {}
        w._mau_group = self

    def _build(self, code: int, ptrs):
        import ctypes
        dev = self.params[0].device
        dt = _DT[code]
        wf = [torch.empty(lib.mau_conv3x3_packed_elems(code, w.shape[0], w.shape[1]), dtype=dt, device=dev) for w in self.params]
        wd = [torch.empty(lib.mau_conv3x3_packed_elems(code, w.shape[1], w.shape[0]), dtype=dt, device=dev) for w in self.params]
        nbytes = lib.mau_conv3x3_pack_desc_bytes()
        host = ctypes.create_string_buffer(nbytes * len(self.params))
        nxt = ctypes.c_int(0)
        for i, w in enumerate(self.params):
            call("mau_conv3x3_pack_desc_fill", ctypes.addressof(host), i, w.data_ptr(), wf[i].data_ptr(), wd[i].data_ptr(), code,
                 w.shape[0], w.shape[1], nxt.value, ctypes.addressof(nxt))
        table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        return {"ptrs": ptrs, "wf": wf, "wd": wd, "table": table, "tiles": nxt.value, "key": None}

    def ensure(self, code: int):
        ptrs = tuple(w.data_ptr() for w in self.params)
        st = self._state.get(code)
        if st is None or st["ptrs"] != ptrs:                      # first use, or the module moved (.to(), .cuda())
            st = self._state[code] = self._build(code, ptrs)
        key = (_GENERATION[0], tuple(w._version for w in self.params))
        if st["key"] != key:
            # (optim.AdamW has already written the packs of this generation while it updated the weights: nothing to launch)
            if not (self.fresh_after_step == _GENERATION[0] and st["key"] is not None and st["key"][1] == key[1] and len(self._state) == 1):
                call("mau_conv3x3_pack_weights_multi", st["table"].data_ptr(), len(self.params), st["tiles"], code, _stream())
            st["key"] = key
            for w, f, d in zip(self.params, st["wf"], st["wd"]):
                w._mau_pack = [(key[0], w._version, w.data_ptr(), code), f, d]
        return st


def pack_conv_weights(w: torch.Tensor, code: int, forward: bool = True, dgrad: bool = False):
    """OIHW fp32 master weights -> (forward pack, data-gradient pack) in the activation dtype.

    The packs are cached on the parameter and keyed by (optimizer-step generation, ``w._version``, storage address,
    dtype): they are rebuilt once per optimizer step (``mark_params_updated``), not once per forward.  A version-only
    key served stale weights from step 2 on with fused AdamW (``tests/test_gpu_model.py::
    test_multi_step_training_tracks_oracle`` is the regression test); the generation counter closes that hole.
    A weight that belongs to a network (``PackGroup``) is re-packed together with all the others in one launch; a
    stand-alone weight by one launch of its own.  Either way the buffers are persistent and re-packed IN PLACE.
    """
    cout, cin = w.shape[0], w.shape[1]
    key = (_GENERATION[0], w._version, w.data_ptr(), code)
    cache = getattr(w, "_mau_pack", None)
    if cache is not None and cache[0] == key and (cache[1] is not None or not forward) and (cache[2] is not None or not dgrad):
        return (cache[1] if forward else None), (cache[2] if dgrad else None)
    group = getattr(w, "_mau_group", None)
    if group is not None and _PACK_MULTI:
        group.ensure(code)
        cache = w._mau_pack
        return (cache[1] if forward else None), (cache[2] if dgrad else None)
    dt = _DT[code]
    if cache is None:
        cache = [None, None, None]
        try:
            w._mau_pack = cache
        except (AttributeError, RuntimeError):      # a tensor that refuses attributes: no caching
            pass

    def buf(old, n):                                # persistent: same dtype / device / size -> re-packed in place
        if old is not None and old.dtype == dt and old.device == w.device and old.numel() == n:
            return old
        return torch.empty(n, dtype=dt, device=w.device)

    stale = cache[0] != key
    need_f = forward and (stale or cache[1] is None or cache[1].dtype != dt)
    need_d = dgrad and (stale or cache[2] is None or cache[2].dtype != dt)
    wf = buf(cache[1], lib.mau_conv3x3_packed_elems(code, cout, cin)) if need_f else None
    wd = buf(cache[2], lib.mau_conv3x3_packed_elems(code, cin, cout)) if need_d else None
    if need_f or need_d:
        call("mau_conv3x3_pack_weights", w.detach().data_ptr(), wf.data_ptr() if need_f else None, wd.data_ptr() if need_d else None,
             code, cout, cin, _stream())
    if stale:                                       # what was not re-packed now is out of date
        cache[1] = wf
        cache[2] = wd
    else:
        if need_f:
            cache[1] = wf
        if need_d:
            cache[2] = wd
    cache[0] = key
    return (cache[1] if forward else None), (cache[2] if dgrad else None)


# --------------------------------------------------------------------------- #
# layout at the boundary
# --------------------------------------------------------------------------- #
class ToNHWC(torch.autograd.Function):
    """(B,C,H,W) fp32 as delivered by collate_fn (src/dataset.py:99-106) -> NHWC-ld."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, dt: torch.dtype):
        _require_cuda(x, "UrbanPredictor.forward(maps)")
        x = x.contiguous().float()
        N, Cc, H, W = x.shape
        ld = pad8(Cc)
        out = torch.empty((N, H, W, ld), dtype=dt, device=x.device)
        call("mau_nchw_to_nhwc", x.data_ptr(), out.data_ptr(), dtype_code(dt), N, Cc, H, W, ld, _stream())
        ctx.C = Cc
        return out

    @staticmethod
    def backward(ctx, g):
        g = _as_nhwc(g)
        N, H, W, _ = g.shape
        out = torch.empty((N, ctx.C, H, W), dtype=torch.float32, device=g.device)
        call("mau_nhwc_to_nchw", g.data_ptr(), out.data_ptr(), dtype_code(g.dtype), N, ctx.C, H, W, _ld(g), _stream())
        return out, None


def to_nchw(a: Act) -> torch.Tensor:
    """Debug/test helper: NHWC-ld -> (N,C,H,W) fp32 (no autograd)."""
    t = _as_nhwc(a.t.detach())
    N, H, W, _ = t.shape
    out = torch.empty((N, a.C, H, W), dtype=torch.float32, device=t.device)
    call("mau_nhwc_to_nchw", t.data_ptr(), out.data_ptr(), dtype_code(t.dtype), N, a.C, H, W, _ld(t), _stream())
    return out


# --------------------------------------------------------------------------- #
# conv3x3 + BatchNorm2d + ReLU
# --------------------------------------------------------------------------- #
@dataclass
class BNState:
    """Non-tensor configuration of one conv-bn-relu (module buffers are passed as tensors)."""
    training: bool
    C0: int                               # logical channels of the (first) tensor source
    momentum: float = BN_MOMENTUM
    eps: float = BN_EPS
    group: object = None                  # torch.distributed process group for SyncBN (None = local BN)
    comm: object = None                   # dist.RcclComm of that group (RCCL called directly on the compute stream) or None
    world: int = 1
    grad_enabled: bool = True             # torch.is_grad_enabled() at call time (invisible inside Function.forward)
    frozen: object = None                 # dict of a frozen inference session (packed weights, scale, shift) or None
    C1: int = 0                           # logical channels of the second tensor source (virtual concat), 0 = none
    pool: bool = False                    # also return maxpool2x2(output) (encoder blocks: skip + next level)
    out_view: object = None               # preallocated NHWC-ld view the activation is written into (U-Net++ row buffers)
    head: object = None                   # None | True (tanh on channel 0 when out_channels == 2) | False (bare 1x1): return final(activation)
    up_to: object = None                  # None | (H, W): return the bilinear (align_corners=True) resize of the activation
    first: bool = False                   # x is the network input (N, C, H, W) fp32, C <= 8: the first-layer kernel reads it as it is
    dtype: object = None                  # activation dtype of the network (first=True: x itself is fp32)
    dgrad_first: bool = False             # backward: enqueue the data gradient before the weight-gradient branch's kernel (model._Runtime)


def _all_reduce_(t: torch.Tensor, st: BNState):
    """SUM all-reduce of a BatchNorm message, in place, ordered on the CURRENT stream between the kernel that wrote ``t`` and the one
    that reads it.  ``st.comm`` (dist.RcclComm, set by ``model.set_sync_bn``): ``ncclAllReduce`` enqueued on this very stream -- no
    stream hand-over, no host wait, a plain kernel node under graph capture.  Without a communicator (gloo rehearsal / CPU tests,
    ``MAU_RCCL_DIRECT=0``) the group's torch.distributed backend is used."""
    if st.comm is not None:
        st.comm.all_reduce(t, 0, _stream())
    elif st.group is not None:             # an explicitly given group is used even when it has one rank
        from .dist import all_reduce_sum
        all_reduce_sum(t, st.group)


# Weight gradients on a side stream.  "1": beside the data gradient of the same layer, joined before the block's backward returns
# (two MFMA kernels compete for the same pipes: ~1 %, profiles/r1).  "2": NOT joined per layer -- the weight-gradient chain (MFMA-bound)
# runs beside everything the main stream does next, notably the HBM-bound BatchNorm-backward passes of the following layers; one
# join when the backward pass ends (autograd engine callback).  Same kernels, same arithmetic: bit-identical.  Measured (one box,
# B=32 U-Net step): eager 13.80 -> 12.93 ms (the second stream fills the launch gaps of the first), hipGraph 12.93 -> 12.85 ms
# (a 227-register weight-gradient wave leaves no room for a BatchNorm wave on its SIMD: the kernels share the chip by CU, not by
# issue slot).  The eager figure depends on the host: on a box with a slow / busy CPU the extra stream bookkeeping made the eager
# step SLOWER (17-18 vs 14-16 ms); the captured step does not care.  Default "2"; "0" = one stream.
# Test hooks (plain module attributes; the tests flip them with monkeypatch to compare a fused path with the launches it replaces, bit for
# bit).  They were environment switches while the A/B measurements ran (EXPERIMENTS.md); every one is decided, none is read from the
# environment any more.
# (MAU_OVERLAP_WGRAD is the one environment variable left here: 0 = everything on one stream, the form a per-kernel profile needs --
#  scripts/profile.sh, scripts/step_trace.py)
_OVERLAP_WGRAD = int(os.environ.get("MAU_OVERLAP_WGRAD", "2") or 0)      # weight gradients on a side stream: 0 = off, 1 = joined per layer, 2 = ONE join when the backward pass ends
_FUSED_REDUCE = True        # single-launch slab reductions (tickets)
_FUSED_BN = True            # BatchNorm passes fused with pool / head / upsample
_FUSED_UP = True            # (the upsample member of the above, separately switchable)
_PACK_MULTI = True          # one multi-tensor weight-pack launch per optimizer step (functional.PackGroup)
_INFER_POOL_FUSED = True    # inference: an encoder block's max-pool out of the convolution's epilogue (mau_conv3x3_fwd_pool)
_FIRST_WGRAD = True         # the first layer's weight gradient on its own kernel (csrc/conv3x3_first.hip)
_SIDE_STREAMS = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def _join_side_when_backward_ends(dev):
    """Queue the join of the weight-gradient stream: the engine runs the callback on the thread that called backward(), after the
    last node -- whatever follows on that thread's stream (any optimizer, a gradient clip, a collective) sees finished gradients.
    One callback per deferred launch, no state kept between passes (a backward pass that raises never runs its callbacks: a
    "queued already" flag would then stay set and silently drop the join of every later pass); waits after the first are no-ops."""
    def join():
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))

    torch.autograd.Variable._execution_engine.queue_callback(join)


# launch order inside ConvBNReLU.backward: None = the network's choice (BNState.dgrad_first), "0" / "1" = forced (test hook)
_DGRAD_FIRST = None


def _conv_fwd(x, x1, st, emb, emb_ws, E, wpk, bias, post, y, Cout, slab, code, N, H, W, stream):
    """One launch of the implicit-GEMM convolution over cat([x, x1, broadcast(emb)], channels)."""
    scale, shift = post if post is not None else (None, None)
    call("mau_conv3x3_fwd2", x.data_ptr(), _ld(x), st.C0, x1.data_ptr() if x1 is not None else None,
         _ld(x1) if x1 is not None else 0, st.C1 if x1 is not None else 0, emb.data_ptr() if E else None,
         emb_ws.data_ptr() if E else None, E, wpk.data_ptr(), bias.data_ptr() if bias is not None else None,
         scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None,
         y.data_ptr(), _ld(y), Cout, slab.data_ptr() if slab is not None else None, code, N, H, W, stream)


class ConvBNReLU(torch.autograd.Function):
    """relu(bn(conv3x3(cat([x, x1, broadcast(emb)], 1)))) -- one half of VGGBlock.forward (src/model.py:18-21) with the
    decoder's channel concat (src/model.py:279-282) and fuse_embeddings (:248-259) as loader sources, never materialised.

    What happens to the activation decides the variant (``BNState``), so that a full-resolution tensor is never written just
    to be read once by a streaming operator:
      * ``pool``  -- also returns nn.MaxPool2d(2,2) of the activation (src/model.py:268-271), written in the same pass; the
                     backward recomputes "skip gradient + pool gradient" inside both BatchNorm passes (no ``da`` tensor);
      * ``head``  -- returns final(activation) with tanh on channel 0 (src/model.py:284-292) instead of the activation: the
                     1x1 head reads the raw conv output and applies BatchNorm + ReLU on the fly, forward and backward;
      * ``up_to`` -- returns up(activation) (bilinear x2, align_corners=True, src/model.py:219,279-282) instead of the
                     activation: the resize applies BatchNorm + ReLU to the four corners it loads.
    ``functional._FUSED_BN = False`` (a test hook) runs the same variants through the separate kernels, bit-identical."""

    @staticmethod
    def forward(ctx, x, x1, emb, weight, bias, gamma, beta, rmean, rvar, nbt, hw, hb, st: BNState):
        _require_cuda(x, "conv3x3")
        first = st.first
        if first:
            # the network's first convolution on the input tensor as delivered (src/model.py:222, src/dataset.py:99-106)
            if x1 is not None or emb is not None or x.dim() != 4 or x.shape[1] != st.C0 or st.C0 > lib.mau_conv3x3_first_max_channels():
                raise RuntimeError("conv3x3 (first layer): expects one (N, C <= 8, H, W) fp32 input")
            xin = x.contiguous().float()
            N, _, H, W = xin.shape
            act_dt = st.dtype
        else:
            x = _as_nhwc(x)
            if x1 is not None:
                x1 = _as_nhwc(x1)
            N, H, W, _ = x.shape
            act_dt = x.dtype
        code = dtype_code(act_dt)
        dev = x.device
        Cout, Cin = weight.shape[0], weight.shape[1]
        E = 0 if emb is None else emb.shape[1]
        C1 = st.C1 if x1 is not None else 0
        if st.C0 + C1 + E != Cin:
            raise RuntimeError(f"conv3x3: input has {st.C0}+{C1}+{E} channels, weight expects {Cin}")
        if emb is not None:
            emb = emb.contiguous().float()
        ldy = pad8(Cout)
        stream = _stream()
        needs = ctx.needs_input_grad
        inference = (not st.training) and (not st.grad_enabled or not any(needs))
        emb_ws = torch.empty((N, E), dtype=act_dt, device=dev) if E else None
        f32 = dict(dtype=torch.float32, device=dev)
        dst = st.out_view                               # where the activation goes: a slot of a row buffer, or a fresh tensor
        if dst is not None and (tuple(dst.shape) != (N, H, W, ldy) or dst.dtype != act_dt or Cout % 64 != 0):
            raise RuntimeError("conv3x3: out_view must be an (N,H,W,Cout) NHWC-ld view of the activation dtype with Cout % 64 == 0")
        if sum((st.pool, st.head is not None, st.up_to is not None)) > 1 or ((st.head is not None or st.up_to is not None) and dst is not None):
            raise RuntimeError("conv3x3: pool / head / up_to are exclusive, and head / up_to write no activation (no out_view)")
        Co = tanh0 = 0
        w2 = None
        if st.head is not None:
            Co = hw.shape[0]
            tanh0 = 1 if (Co == 2 and st.head) else 0
            w2 = hw.detach().reshape(Co, Cout).contiguous().float()

        def first_fwd(post, y_, slab_, x8_):
            """conv0_0.conv1 straight from the fp32 NCHW input and the fp32 master weights (no layout kernel, no weight pack)"""
            scale_, shift_ = post if post is not None else (None, None)
            call("mau_conv3x3_first_fwd", xin.data_ptr(), st.C0, weight.detach().data_ptr(), bias.detach().data_ptr() if bias is not None else None,
                 scale_.data_ptr() if scale_ is not None else None, shift_.data_ptr() if shift_ is not None else None, y_.data_ptr(), _ld(y_),
                 Cout, slab_.data_ptr() if slab_ is not None else None, x8_.data_ptr() if x8_ is not None else None, code, N, H, W, stream)

        def pooled_of(a):
            pl = torch.empty((N, H // 2, W // 2, ldy), dtype=act_dt, device=dev)
            call("mau_maxpool2x2_fwd", a.data_ptr(), _ld(a), pl.data_ptr(), ldy, code, N, H, W, Cout, stream)
            return pl

        def head_of(a):                                 # final 1x1 conv (+ tanh on channel 0) of a materialised activation
            out = torch.empty((N, Co, H, W), **f32)
            call("mau_head_fwd", a.data_ptr(), _ld(a), w2.data_ptr(), hb.detach().data_ptr(), out.data_ptr(), tanh0, code, N, H * W,
                 Cout, Co, stream)
            return out

        def up_of(a):                                   # bilinear resize of a materialised activation
            Hu, Wu = st.up_to
            up = torch.empty((N, Hu, Wu, ldy), dtype=act_dt, device=dev)
            call("mau_resize_bilinear_fwd", a.data_ptr(), _ld(a), H, W, up.data_ptr(), ldy, 0, code, N, Hu, Wu, Cout, stream)
            return up

        if inference:
            # eval-mode BN + ReLU are a fixed per-channel affine map -> folded into the conv epilogue.  In a frozen
            # session (UrbanPredictor.freeze_inference) the packed weights and the folded coefficients are computed once.
            y = dst if dst is not None else torch.empty((N, H, W, ldy), dtype=act_dt, device=dev)
            fz = st.frozen if st.frozen is not None else {}
            fkey = (_GENERATION[0], weight._version, gamma._version, rmean._version, rvar._version)
            if "wf" not in fz or fz["wf"].dtype != act_dt or fz.get("key") != fkey:
                # (an optimizer step -- of ANY model: the generation counter is global --, mark_params_updated() or an in-place
                #  write re-derives the copies.)  Everything is refreshed IN PLACE: the packs are persistent buffers and
                # scale / shift are allocated once per session, so a live GraphedInference graph, which holds these
                # addresses, replays with the new values instead of reading freed memory.
                fz["key"] = fkey
                fz["wf"] = weight if first else pack_conv_weights(weight, code, forward=True, dgrad=False)[0]     # (first layer: the master weights themselves)
                if "scale" not in fz or fz["scale"].device != dev or fz["scale"].numel() != Cout:
                    fz["scale"], fz["shift"] = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
                call("mau_bn_coeffs_eval", gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(),
                     st.eps, fz["scale"].data_ptr(), fz["shift"].data_ptr(), None, None, Cout, stream)
            pl = None
            if first:
                first_fwd((fz["scale"], fz["shift"]), y, None, None)
            elif st.pool and _INFER_POOL_FUSED and x1 is None and not E and H >= 2 and W >= 2:
                # an encoder block's second convolution: the pooled tensor comes out of the same launch (the 16x16x32 tilings take the
                # 2x2 maxima from the registers they store; other tilings run the pooling kernel behind the convolution inside the call)
                pl = torch.empty((N, H // 2, W // 2, ldy), dtype=act_dt, device=dev)
                call("mau_conv3x3_fwd_pool", x.data_ptr(), _ld(x), st.C0, fz["wf"].data_ptr(), bias.data_ptr() if bias is not None else None,
                     fz["scale"].data_ptr(), fz["shift"].data_ptr(), y.data_ptr(), _ld(y), Cout, pl.data_ptr(), ldy, code, N, H, W, stream)
            else:
                _conv_fwd(x, x1, st, emb, emb_ws, E, fz["wf"], bias, (fz["scale"], fz["shift"]), y, Cout, None, code, N, H, W, stream)
            if st.head is not None:
                out = head_of(y)
                ctx.mark_non_differentiable(out)
                return out
            if st.up_to is not None:
                up = up_of(y)
                ctx.mark_non_differentiable(up)
                return up
            ctx.mark_non_differentiable(y)
            if st.pool:
                if pl is None:
                    pl = pooled_of(y)
                ctx.mark_non_differentiable(pl)
                return y, pl
            return y
        y = torch.empty((N, H, W, ldy), dtype=act_dt, device=dev)
        need_dx = needs[0] or (x1 is not None and needs[1]) or (E > 0 and needs[2])
        if first:
            if need_dx:
                raise RuntimeError("conv3x3 (first layer): no gradient w.r.t. the network input on this path")
            wf = wd = None
            # the input in the activation type as NHWC-8: what this layer's weight gradient reads; a by-product of the same pass
            x = torch.empty((N, H, W, 8), dtype=act_dt, device=dev) if (st.grad_enabled and needs[3]) else None
        else:
            wf, wd = pack_conv_weights(weight, code, forward=True, dgrad=st.grad_enabled and need_dx)
        scale, shift = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        mean, invstd = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        npix = N * H * W
        if st.training:
            tiles = lib.mau_conv3x3_first_rows(N, H, W) if first else lib.mau_conv3x3_num_pixel_tiles(code, N, H, W, Cout)
            cpad = (Cout + 63) // 64 * 64
            slab = torch.empty((tiles, 2 * cpad), **f32)
            if first:
                first_fwd(None, y, slab, x)
            else:
                _conv_fwd(x, x1, st, emb, emb_ws, E, wf, bias, None, y, Cout, slab, code, N, H, W, stream)
            nbt_ptr = nbt.data_ptr() if nbt is not None else None
            tk = _tickets(dev).data_ptr() if _FUSED_REDUCE else None
            if st.group is None:
                # single GPU: slab -> fp64 partials -> second level + finalize by the last workgroup, ONE launch
                ws = torch.empty(lib.mau_bn_stats_ws_elems(tiles, Cout), dtype=torch.float64, device=dev)
                call("mau_bn_stats_finalize_train", slab.data_ptr(), tiles, float(npix), gamma.data_ptr(), beta.data_ptr(),
                     rmean.data_ptr(), rvar.data_ptr(), nbt_ptr, st.momentum, st.eps, scale.data_ptr(), shift.data_ptr(),
                     mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), tk, Cout, stream)
            else:
                # SyncBN: ONE collective per layer carries [sum(y) | sum(y^2) | local pixel count]; the count is summed
                # with the statistics, so ranks with different local batch sizes still agree on the global moments
                sums = torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev)
                ws = torch.empty(lib.mau_reduce_rows_ws_elems(tiles, 2 * Cout), dtype=torch.float64, device=dev)
                call("mau_bn_stats_sums_f64", slab.data_ptr(), tiles, Cout, sums.data_ptr(), ws.data_ptr(), _tickets(dev).data_ptr(),
                     float(npix), stream)
                _all_reduce_(sums, st)
                call("mau_bn_finalize_train", sums.data_ptr(), 0.0, gamma.data_ptr(), beta.data_ptr(),
                     rmean.data_ptr(), rvar.data_ptr(), nbt_ptr, st.momentum, st.eps, scale.data_ptr(), shift.data_ptr(),
                     mean.data_ptr(), invstd.data_ptr(), Cout, stream)
        else:
            call("mau_bn_coeffs_eval", gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(),
                 st.eps, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), Cout, stream)
            if first:
                first_fwd(None, y, None, x)
            else:
                _conv_fwd(x, x1, st, emb, emb_ws, E, wf, bias, None, y, Cout, None, code, N, H, W, stream)

        # ---- the activation stage ----
        fused = _FUSED_BN
        fuse_head = fused and st.head is not None and Cout <= lib.mau_head_bn_max_channels() and H * W >= 32
        fuse_up = fused and _FUSED_UP and st.up_to is not None and H <= st.up_to[0] and W <= st.up_to[1]
        a = pl = out = up = argidx = None
        if fuse_head:                                    # the head reads y and applies BatchNorm + ReLU on the fly: no activation tensor
            out = torch.empty((N, Co, H, W), **f32)
            call("mau_head_bn_fwd", y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), w2.data_ptr(), hb.detach().data_ptr(),
                 out.data_ptr(), tanh0, code, N, H * W, Cout, Co, stream)
        elif fuse_up:                                    # the resize applies BatchNorm + ReLU to the corners it loads
            Hu, Wu = st.up_to
            up = torch.empty((N, Hu, Wu, ldy), dtype=act_dt, device=dev)
            call("mau_resize_bilinear_bn_fwd", y.data_ptr(), ldy, H, W, scale.data_ptr(), shift.data_ptr(), up.data_ptr(), ldy, 0, code,
                 N, Hu, Wu, Cout, stream)
        else:
            a = dst if dst is not None else torch.empty_like(y)
            lda = _ld(a)
            if st.pool and H >= 2 and W >= 2:
                pl = torch.empty((N, H // 2, W // 2, ldy), dtype=act_dt, device=dev)
                # 2 bits per (window, channel): which pixel holds the maximum -- the routing of the pool's backward (fused with the
                # BatchNorm backward passes, no activation is re-read for it)
                argidx = torch.empty((N, H // 2, W // 2, ldy // 8), dtype=torch.int16, device=dev) if fused else None
                call("mau_bn_relu_apply_pool", y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), a.data_ptr(), lda,
                     pl.data_ptr(), ldy, argidx.data_ptr() if argidx is not None else None, code, N, H, W, Cout, stream)
            else:
                call("mau_bn_relu_apply", y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), a.data_ptr(), lda, code,
                     npix, Cout, stream)
            if st.head is not None:
                out = head_of(a)
            elif st.up_to is not None:
                up = up_of(a)
        # (the context must not hold the output: st.out_view IS the returned activation, and node -> ctx -> st -> out_view -> grad_fn
        #  -> node is a reference cycle that only Python's cycle collector frees -- a whole step's graph, its AccumulateGrad nodes
        #  included, then outlives the step; a hipGraph capture that meets those stale nodes dies in hipStreamEndCapture)
        ctx.st = st if st.out_view is None else dataclasses.replace(st, out_view=None)
        ctx.E = E
        ctx.C1 = C1
        ctx.wd = wd                           # data-gradient pack of THIS forward's weights
        ctx.pooled = pl is not None
        ctx.head = (Co, tanh0, tuple(hw.shape), fuse_head) if st.head is not None else None
        ctx.up = (st.up_to, fuse_up) if st.up_to is not None else None
        # the activation is saved only where the backward still reads it: the unfused pool (arg-max) and the unfused head (dW)
        keep_a = a if ((pl is not None and not fused) or (st.head is not None and not fuse_head)) else None
        ctx.save_for_backward(x, x1, emb, weight, y, scale, shift, mean, invstd, keep_a, w2, out, argidx)
        if st.head is not None:
            return out
        if st.up_to is not None:
            return up
        if st.pool:
            return a, pl
        return a

    @staticmethod
    def backward(ctx, da, dpl=None):
        x, x1, emb, weight, y, scale, shift, mean, invstd, a, w2, out, argidx = ctx.saved_tensors
        st: BNState = ctx.st
        E, C1 = ctx.E, ctx.C1
        N, H, W, ldy = y.shape
        code = dtype_code(y.dtype)
        dev = y.device
        Cout, Cin = weight.shape[0], weight.shape[1]
        npix = N * H * W
        stream = _stream()
        f32 = dict(dtype=torch.float32, device=dev)
        needs = ctx.needs_input_grad
        sync = st.training and st.group is not None
        rows = lib.mau_bn_bwd_rows(npix)
        slab = torch.empty((rows, 2 * Cout), **f32)
        sums = torch.empty(2 * Cout + 1, dtype=torch.float64, device=dev)
        g32 = torch.empty(2 * Cout, **f32)                   # [dbeta | dgamma]: the LOCAL sums, rounded to fp32 by the reducer
        dy = torch.empty_like(y)
        dhw = dhb = None
        cptrs = (scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())

        def finish_sums():
            """slab -> sums (fp64) + g32 (fp32), one launch; data parallel: all-reduce with the pixel count appended.
            Returns (sums for the apply pass, count argument)."""
            ws = torch.empty(lib.mau_reduce_rows_ws_elems(rows, 2 * Cout), dtype=torch.float64, device=dev)
            call("mau_reduce_rows_f64_f32", slab.data_ptr(), rows, 2 * Cout, 2 * Cout, sums.data_ptr(), g32.data_ptr(), ws.data_ptr(),
                 _tickets(dev).data_ptr() if (_FUSED_REDUCE or sync) else None, float(npix) if sync else 0.0, stream)
            if not st.training:
                return torch.zeros_like(sums), float(npix)       # eval-mode BN is a fixed affine map
            if sync:                                             # the global count travels with the sums (see forward)
                _all_reduce_(sums, st)
                return sums, 0.0
            return sums, float(npix)

        fused_pool = ctx.pooled and dpl is not None and argidx is not None
        if ctx.head is not None and ctx.head[3]:
            # ---- head + BatchNorm backward, two passes over y: da = W_head^T dz is recomputed from dout in both ----
            Co, tanh0, hshape, _ = ctx.head
            dout = da.contiguous().float()
            hrows, rowlen = lib.mau_head_bwd_rows(N, H * W), lib.mau_head_bwd_rowlen(Cout, Co)
            hslab = torch.empty((hrows, rowlen), **f32)
            call("mau_head_bn_bwd_reduce", y.data_ptr(), ldy, *cptrs, w2.data_ptr(), out.data_ptr(), dout.data_ptr(), slab.data_ptr(), Cout,
                 hslab.data_ptr(), tanh0, code, N, H * W, Cout, Co, stream)
            sums_apply, count = finish_sums()
            red = torch.empty(rowlen, **f32)
            hws = torch.empty(lib.mau_reduce_rows_ws_elems(hrows, rowlen), dtype=torch.float64, device=dev)
            call("mau_reduce_rows_f32", hslab.data_ptr(), hrows, rowlen, rowlen, red.data_ptr(), hws.data_ptr(),
                 _tickets(dev).data_ptr() if _FUSED_REDUCE else None, stream)
            red = red.view(Co, pad8(Cout) + 8)
            dhw = red[:, :Cout].reshape(hshape).contiguous()
            dhb = red[:, pad8(Cout)].contiguous()
            call("mau_head_bn_bwd_apply", y.data_ptr(), ldy, *cptrs, sums_apply.data_ptr(), count, w2.data_ptr(), out.data_ptr(),
                 dout.data_ptr(), dy.data_ptr(), ldy, tanh0, code, N, H * W, Cout, Co, stream)
        elif fused_pool:
            # ---- pool + skip + BatchNorm backward, two passes over (y, dpl, dskip): no da tensor ----
            dpl = _as_nhwc(dpl)
            dsk = _as_nhwc(da) if da is not None else None
            pargs = (y.data_ptr(), ldy, dpl.data_ptr(), _ld(dpl), argidx.data_ptr(), dsk.data_ptr() if dsk is not None else None,
                     _ld(dsk) if dsk is not None else 0)
            call("mau_pool_bn_bwd_reduce", *pargs, *cptrs, slab.data_ptr(), Cout, code, N, H, W, Cout, stream)
            sums_apply, count = finish_sums()
            call("mau_pool_bn_bwd_apply", *pargs, *cptrs, sums_apply.data_ptr(), count, dy.data_ptr(), ldy, code, N, H, W, Cout, stream)
        else:
            if ctx.head is not None:
                # unfused head: da, dW, db from the saved activation
                Co, tanh0, hshape, _ = ctx.head
                dout = da.contiguous().float()
                da = torch.empty((N, H, W, ldy), dtype=y.dtype, device=dev)
                hrows, rowlen = lib.mau_head_bwd_rows(N, H * W), lib.mau_head_bwd_rowlen(Cout, Co)
                hslab = torch.empty((hrows, rowlen), **f32)
                call("mau_head_bwd", a.data_ptr(), _ld(a), w2.data_ptr(), out.data_ptr(), dout.data_ptr(), da.data_ptr(), ldy,
                     hslab.data_ptr(), tanh0, code, N, H * W, Cout, Co, stream)
                red = torch.empty(rowlen, **f32)
                hws = torch.empty(lib.mau_reduce_rows_ws_elems(hrows, rowlen), dtype=torch.float64, device=dev)
                call("mau_reduce_rows_f32", hslab.data_ptr(), hrows, rowlen, rowlen, red.data_ptr(), hws.data_ptr(),
                     _tickets(dev).data_ptr() if _FUSED_REDUCE else None, stream)
                red = red.view(Co, pad8(Cout) + 8)
                dhw = red[:, :Cout].reshape(hshape).contiguous()
                dhb = red[:, pad8(Cout)].contiguous()
            elif ctx.up is not None:
                # the upsample's adjoint produces da (its forward read y through BatchNorm + ReLU, or the activation)
                (Hu, Wu), _ = ctx.up
                g = _as_nhwc(da)
                da = torch.empty((N, H, W, ldy), dtype=y.dtype, device=dev)
                call("mau_resize_bilinear_bwd", g.data_ptr(), _ld(g), 0, Hu, Wu, da.data_ptr(), ldy, code, N, H, W, Cout, stream)
            elif ctx.pooled and dpl is not None:
                # the block's output fed the pool AND the skip connection: one pass adds the two gradients (PoolSkip's job)
                dpl = _as_nhwc(dpl)
                act = a
                if act is None:                          # (fused build, but this backward runs unfused: recompute the activation)
                    act = torch.empty_like(y)
                    call("mau_bn_relu_apply", y.data_ptr(), ldy, scale.data_ptr(), shift.data_ptr(), act.data_ptr(), ldy, code, npix, Cout, stream)
                dsum = torch.empty((N, H, W, ldy), dtype=y.dtype, device=dev)
                if da is None:
                    call("mau_maxpool2x2_bwd", act.data_ptr(), _ld(act), dpl.data_ptr(), _ld(dpl), dsum.data_ptr(), ldy, code, N, H, W, Cout, stream)
                else:
                    da = _as_nhwc(da)
                    call("mau_maxpool2x2_bwd_add", act.data_ptr(), _ld(act), dpl.data_ptr(), _ld(dpl), da.data_ptr(), _ld(da),
                         dsum.data_ptr(), ldy, code, N, H, W, Cout, stream)
                da = dsum
            elif da is None:
                da = torch.zeros((N, H, W, ldy), dtype=y.dtype, device=dev)
            da = _as_nhwc(da)
            # --- BN + ReLU backward: two passes over (da, y) ---
            call("mau_bn_relu_bwd_reduce", da.data_ptr(), _ld(da), y.data_ptr(), ldy, *cptrs, slab.data_ptr(), Cout, code, npix, Cout, stream)
            sums_apply, count = finish_sums()
            call("mau_bn_relu_bwd_apply", da.data_ptr(), _ld(da), y.data_ptr(), ldy, *cptrs, sums_apply.data_ptr(), count, dy.data_ptr(), ldy,
                 code, npix, Cout, stream)
        dgamma = g32[Cout:]
        dbeta = g32[:Cout]
        # --- weight gradient: independent of the data gradient given dy ---
        dw = None
        need_dx = needs[0] or (x1 is not None and needs[1]) or (E and needs[2])
        # (under dist.GradSync the buckets' all-reduces run on a communication stream that waits for the side stream by itself:
        #  the compute stream is not held up, so the overlap stays on -- round 3 had to switch it off there)
        overlap = _OVERLAP_WGRAD
        side = _side_stream(dev) if (needs[3] and overlap and (need_dx or overlap >= 2)) else None
        deferred = False
        dfull = None
        if (st.dgrad_first if _DGRAD_FIRST is None else _DGRAD_FIRST == "1") and need_dx and needs[3] and side is not None:
            # the main chain's data gradient is enqueued BEFORE the branch's weight gradient (both need only dy: whichever is
            # enqueued first takes the chip, the other follows when its workgroups retire); the branch still waits only for dy --
            # the event is recorded in front of the data gradient.  Which order is faster depends on the network (model._Runtime).
            dy_ready = torch.cuda.Event()
            dy_ready.record()
            wd = ctx.wd if ctx.wd is not None else pack_conv_weights(weight, code, forward=False, dgrad=True)[1]
            dfull = torch.empty((N, H, W, pad8(Cin)), dtype=y.dtype, device=dev)
            call("mau_conv3x3_fwd", dy.data_ptr(), ldy, Cout, None, None, 0, wd.data_ptr(), None, None, None, dfull.data_ptr(),
                 pad8(Cin), Cin, None, code, N, H, W, stream)
            side.wait_event(dy_ready)
        if needs[3]:
            if side is not None and dfull is None:
                side.wait_stream(torch.cuda.current_stream())           # dy is complete
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                # (workspaces of the side stream's kernels belong to ITS allocator pool: freed here, re-used there)
                acc = torch.empty(max(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, Cout, Cin),
                                      lib.mau_conv3x3_first_wgrad_ws_elems(N, H, W, Cout) if st.first else 0), **f32)
                emb_ws = torch.empty((N, E), dtype=y.dtype, device=dev) if E else None
            # data parallel (dist.GradSync): the split-K sum is written straight into the parameter's slot of the gradient arena
            # and autograd adopts the returned view as weight.grad -- no copy into the arena later.  Only for the FIRST
            # gradient of a step: an accumulating backward (weight.grad already set) must not overwrite what it adds to.
            slot = getattr(weight, "_mau_grad_slot", None)
            dw = slot.view_as(weight) if (slot is not None and weight.is_leaf and weight.grad is None and slot.device == dev) else torch.empty_like(weight)
            wstream = side.cuda_stream if side is not None else stream
            # deferred join: only where the gradient is ASSIGNED (arena slot adopted as weight.grad, or a fresh tensor) -- an
            # accumulating AccumulateGrad would add on the main stream to what the side stream is still writing
            # (and only for a parameter: the gradient of a derived weight -- EmbFold's W_eff -- is read by ITS backward on the main stream)
            deferred = side is not None and overlap >= 2 and weight.is_leaf and weight.grad is None
            if st.first and _FIRST_WGRAD:
                # conv0_0.conv1: K = pixels, N = 9 taps x 8 channels straight from the forward's NHWC-8 by-product (csrc/conv3x3_first.hip)
                call("mau_conv3x3_first_wgrad", x.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), acc.data_ptr(), Cin, Cout, code, N, H, W, wstream)
            else:
                call("mau_conv3x3_wgrad2", x.data_ptr(), _ld(x), st.C0, x1.data_ptr() if x1 is not None else None,
                     _ld(x1) if x1 is not None else 0, C1, emb.data_ptr() if E else None,
                     emb_ws.data_ptr() if E else None, E, dy.data_ptr(), ldy, Cout, acc.data_ptr(), code, N, H, W, wstream)
                call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), lib.mau_conv3x3_wgrad_splits(code, N, H, W, Cout, Cin),
                     dw.data_ptr(), Cout, Cin, wstream)
            if deferred:
                for t in (x, x1, emb, dy, dw):                           # read / written over there after this function has returned
                    if t is not None:
                        t.record_stream(side)
                weight._mau_grad_stream = side                           # (dist.GradSync: the bucket's launch waits for it)
                _join_side_when_backward_ends(dev)
        # --- data gradient (same implicit-GEMM kernel, rotated/transposed weight pack) ---
        dx = dx1 = demb = None
        if need_dx:
            wd = ctx.wd if ctx.wd is not None else pack_conv_weights(weight, code, forward=False, dgrad=True)[1]
            ldd = pad8(Cin)
            if dfull is None:
                dfull = torch.empty((N, H, W, ldd), dtype=y.dtype, device=dev)
                call("mau_conv3x3_fwd", dy.data_ptr(), ldy, Cout, None, None, 0, wd.data_ptr(), None, None, None, dfull.data_ptr(),
                     ldd, Cin, None, code, N, H, W, stream)
            Ct = st.C0 + C1
            if E and needs[2]:
                demb = torch.empty((N, E), **f32)
                ws = torch.empty(lib.mau_bcast_bwd_ws_elems(N, H * W, E), **f32)
                call("mau_bcast_bwd", dfull.data_ptr(), ldd, Ct, demb.data_ptr(), ws.data_ptr(), code, N, H * W, E, stream)
            if C1:
                # the two tensors' gradients are channel slices of one buffer (C0 % 16 == 0; with E > 0 also C1 % 8 == 0)
                if needs[0]:
                    dx = dfull[..., :st.C0]
                if needs[1]:
                    dx1 = dfull[..., st.C0:st.C0 + pad8(C1)]
            elif needs[0]:
                dx = dfull[..., :pad8(st.C0)] if E else dfull           # C0 % 8 == 0 is enforced by the kernel when E > 0
        if side is not None and not deferred:
            torch.cuda.current_stream().wait_stream(side)
        dbias = None
        if needs[4]:
            if st.training:
                # conv bias followed by train-mode BN has an identically zero gradient (the batch mean absorbs it)
                dbias = _zero_bias_grad(Cout, dev)
            else:
                # eval-mode BN is the fixed affine map z = scale*y + shift: d/dbias = sum(dy) = scale * sum(dz)
                dbias = scale * dbeta
        return dx, dx1, demb, dw, dbias, dgamma, dbeta, None, None, None, dhw, dhb, None


# --------------------------------------------------------------------------- #
# MaxPool2d(2,2)
# --------------------------------------------------------------------------- #
class MaxPool2x2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2), src/model.py:218."""

    @staticmethod
    def forward(ctx, x, C):
        x = _as_nhwc(x)
        N, H, W, _ = x.shape
        ldy = pad8(C)
        y = torch.empty((N, H // 2, W // 2, ldy), dtype=x.dtype, device=x.device)
        call("mau_maxpool2x2_fwd", x.data_ptr(), _ld(x), y.data_ptr(), ldy, dtype_code(x.dtype), N, H, W, C, _stream())
        ctx.save_for_backward(x)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _as_nhwc(dy)
        N, H, W, _ = x.shape
        dx = torch.empty((N, H, W, pad8(ctx.C)), dtype=x.dtype, device=x.device)
        call("mau_maxpool2x2_bwd", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), pad8(ctx.C),
             dtype_code(x.dtype), N, H, W, ctx.C, _stream())
        return dx, None


class PoolSkip(torch.autograd.Function):
    """(maxpool2x2(x), x): an encoder activation feeds the next level through the pool AND the decoder through a skip
    connection (src/model.py:268-271 with :279-282).  Autograd would add the two gradients of x in an extra pass over
    the full-resolution tensor; here the skip gradient is added while the pool backward writes dx."""

    @staticmethod
    def forward(ctx, x, C):
        x = _as_nhwc(x)
        N, H, W, _ = x.shape
        ldy = pad8(C)
        y = torch.empty((N, H // 2, W // 2, ldy), dtype=x.dtype, device=x.device)
        call("mau_maxpool2x2_fwd", x.data_ptr(), _ld(x), y.data_ptr(), ldy, dtype_code(x.dtype), N, H, W, C, _stream())
        ctx.save_for_backward(x)
        ctx.C = C
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        (x,) = ctx.saved_tensors
        if dy is None:
            return dskip, None
        dy = _as_nhwc(dy)
        N, H, W, _ = x.shape
        dx = torch.empty((N, H, W, pad8(ctx.C)), dtype=x.dtype, device=x.device)
        if dskip is None:
            call("mau_maxpool2x2_bwd", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), pad8(ctx.C),
                 dtype_code(x.dtype), N, H, W, ctx.C, _stream())
        else:
            dskip = _as_nhwc(dskip)
            call("mau_maxpool2x2_bwd_add", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dskip.data_ptr(), _ld(dskip),
                 dx.data_ptr(), pad8(ctx.C), dtype_code(x.dtype), N, H, W, ctx.C, _stream())
        return dx, None


# --------------------------------------------------------------------------- #
# decoder input: cat([skips..., resize(low)]) on channels
# --------------------------------------------------------------------------- #
def _resize_into(src, Cs, h, w, dst, choff, N, H, W, code, stream):
    if choff % 8 == 0:
        call("mau_resize_bilinear_fwd", src.data_ptr(), _ld(src), h, w, dst.data_ptr(), _ld(dst), choff, code, N, H, W, Cs, stream)
    else:   # tiny-channel models only: resize to a scratch tensor, then element-granular copy
        tmp = torch.empty((N, H, W, pad8(Cs)), dtype=dst.dtype, device=dst.device)
        call("mau_resize_bilinear_fwd", src.data_ptr(), _ld(src), h, w, tmp.data_ptr(), pad8(Cs), 0, code, N, H, W, Cs, stream)
        call("mau_copy_channels", tmp.data_ptr(), pad8(Cs), dst.data_ptr(), _ld(dst), choff, 0, code, N * H * W, Cs, stream)


class RowPrefix(torch.autograd.Function):
    """torch.cat([x_0, ..., x_{k-1}], 1) for activations that ALREADY sit side by side in one row buffer (U-Net++: the
    nodes x^{i,0..j-1} of a row are the leading inputs of node x^{i,j}, src/model.py:136-177): the result is a view of the
    buffer -- no copy -- and the backward hands every input its channel slice of the gradient, also as views."""

    @staticmethod
    def forward(ctx, C, *xs):
        base = xs[0]
        ld, esz = base.stride(2), base.element_size()
        for j, t in enumerate(xs):
            if t.data_ptr() != base.data_ptr() + j * C * esz or t.stride() != base.stride() or t.shape != base.shape:
                raise RuntimeError("RowPrefix: inputs are not adjacent channel slots of one row buffer")
        N, H, W, _ = base.shape
        ctx.C, ctx.k = C, len(xs)
        return base.as_strided((N, H, W, C * len(xs)), base.stride())

    @staticmethod
    def backward(ctx, g):
        g = _as_nhwc(g)
        return (None, *[g[..., j * ctx.C:(j + 1) * ctx.C] for j in range(ctx.k)])


class Fanout(torch.autograd.Function):
    """``Fanout.apply(x, k)`` -> k aliases of ``x``, one per reader.  Forward: views, no copy.  Backward: the readers' gradients
    summed by ONE kernel (``mau_sum_tensors``: fp32 sums in reader order, one rounding) -- autograd's own accumulation of the
    gradients of a shared activation is k - 1 generic strided ``at::add`` launches (the slices of the data-gradient buffers are
    not contiguous), 3 tensors of traffic and a rounding each: 16 of them, 0.85 ms, in a U-Net++ step (src/model.py:136-177: every
    row slot is read by the later nodes of its row and by the node above it)."""

    @staticmethod
    def forward(ctx, x, k: int):
        ctx.k = k
        ctx.meta = (x.shape, x.dtype, x.device)
        return tuple(x.as_strided(x.shape, x.stride()) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        import ctypes
        shape, dtype, dev = ctx.meta
        live = [_as_nhwc(g) for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        N, H, W, C = shape
        ld = pad8(C)
        kmax = lib.mau_sum_tensors_max()
        out = torch.empty((N, H, W, ld), dtype=dtype, device=dev)
        while True:                                          # (more readers than one launch takes: fold the first kmax, go on)
            part = live[:kmax]
            ptrs = (ctypes.c_void_p * len(part))(*[g.data_ptr() for g in part])
            lds = (ctypes.c_int * len(part))(*[_ld(g) for g in part])
            call("mau_sum_tensors", ctypes.addressof(ptrs), ctypes.addressof(lds), len(part), out.data_ptr(), ld, dtype_code(dtype),
                 N * H * W, C, _stream())
            live = [out] + live[kmax:]
            if len(live) == 1:
                break
            out = torch.empty((N, H, W, ld), dtype=dtype, device=dev)
        res = live[0]
        return (res if ld == C else res[..., :C]), None


class EmbFold(torch.autograd.Function):
    """The broadcast embedding of a 3x3 convolution as a rank-one term (csrc/embfold.hip).

    ``W_eff = EmbFold.apply(weight, emb, Ct, Ep)``: ``weight`` (Cout, Ct + E, 3, 3) convolves Ct tensor channels followed by the
    E channels of ``emb`` (N, E) broadcast over the pixels (fuse_embeddings / the U-Net++ nodes' emb_map, src/model.py:248-259,
    :111-121).  ``W_eff`` (Cout, Ct + Ep, 3, 3) = [weight[:, :Ct] | T] with T[co][i] = sum_e weight[co][Ct + e] * emb[i][e] gives the
    SAME output when the convolution is run over the tensors and ``emb_identity(N, Ep)`` -- Ep = roundup(N, 16) indicator channels
    instead of E constant ones (16 or 32 instead of 128: a third of the MACs of a U-Net++ decoder node, forward, data gradient and
    weight gradient alike).  The backward maps dW_eff to dW and demb; nothing else in the path knows about the fold."""

    @staticmethod
    def forward(ctx, weight, emb, Ct: int, Ep: int):
        _require_cuda(weight, "EmbFold")
        w = weight.detach().contiguous().float()
        e = emb.detach().contiguous().float()
        Cout, Cf = w.shape[0], w.shape[1]
        N, E = e.shape
        if Cf != Ct + E or Ep < N:
            raise RuntimeError(f"EmbFold: weight has {Cf} input channels, expected {Ct}+{E}; Ep={Ep} must cover the batch of {N}")
        weff = torch.empty((Cout, Ct + Ep, 3, 3), dtype=torch.float32, device=w.device)
        call("mau_emb_fold_fwd", w.data_ptr(), e.data_ptr(), weff.data_ptr(), Cout, Ct, E, N, Ep, _stream())
        ctx.save_for_backward(weight, emb)
        ctx.dims = (Cout, Ct, E, N, Ep)
        return weff

    @staticmethod
    def backward(ctx, dweff):
        weight, emb = ctx.saved_tensors
        Cout, Ct, E, N, Ep = ctx.dims
        dweff = dweff.contiguous().float()
        w = weight.detach().contiguous().float()
        e = emb.detach().contiguous().float()
        dw = demb = None
        if ctx.needs_input_grad[0]:
            # (like ConvBNReLU: the FIRST gradient of a step is written straight into the parameter's slot of the gradient arena)
            slot = getattr(weight, "_mau_grad_slot", None)
            dw = slot.view_as(weight) if (slot is not None and weight.is_leaf and weight.grad is None and slot.device == w.device) else torch.empty_like(w)
        ws = None
        if ctx.needs_input_grad[1]:
            demb = torch.empty((N, E), dtype=torch.float32, device=w.device)
            ws = torch.empty(lib.mau_emb_fold_ws_elems(Cout, N, E), dtype=torch.float32, device=w.device)
        call("mau_emb_fold_bwd", w.data_ptr(), e.data_ptr(), dweff.data_ptr(), dw.data_ptr() if dw is not None else None,
             demb.data_ptr() if demb is not None else None, ws.data_ptr() if ws is not None else None, Cout, Ct, E, N, Ep, _stream())
        return dw, demb, None, None


_EMB_IDENTITY = {}
_EMB_FOLD = True            # the broadcast embedding folded into roundup(N, 16) indicator channels (training; test hook)
_EMB_FOLD_MIN_WORK = 1 << 20      # batch x pixels from which folding pays (test hook)


def emb_identity(N: int, Ep: int, dev) -> torch.Tensor:
    """The (N, Ep) identity "embedding" of a folded convolution (constant: cached per shape and device)."""
    key = (N, Ep, str(dev))
    t = _EMB_IDENTITY.get(key)
    if t is None:
        t = torch.zeros((N, Ep), dtype=torch.float32, device=dev)
        t[torch.arange(N), torch.arange(N)] = 1.0
        _EMB_IDENTITY[key] = t
    return t


def fold_embedding(weight: torch.Tensor, emb: torch.Tensor, Ct: int, pixels: int):
    """(W_eff, identity embedding) if folding pays for this layer, else None: fewer 16-channel stages than the E broadcast channels,
    and at least MAU_EMB_FOLD_MIN_WORK = 2^20 pixels in the batch.  What is saved grows with batch x pixels, what it costs -- deriving
    and re-packing the weight every step, mapping its gradient back: five small launches per layer -- does not: measured on the
    U-Net++ at B=16, folding the four full-resolution nodes takes 0.65 ms off the 19.8 ms step, the nodes below add nothing
    (+-0.05 ms), and the U-Net's 16 x 16 bottleneck LOSES 0.2 ms.  Ep = roundup(N, 16), one more stage where that makes the stage
    count even (the 16x16x32 multiply loop walks stage pairs)."""
    N, E = emb.shape
    Ep = (N + 15) // 16 * 16
    if ((Ct + Ep) // 16) % 2 == 1 and Ct % 16 == 0:
        Ep += 16
    if not _EMB_FOLD or Ep + 16 > E or Ct % 8 != 0 or N * pixels < _EMB_FOLD_MIN_WORK:
        return None
    return EmbFold.apply(weight, emb, Ct, Ep), emb_identity(N, Ep, emb.device)


class UpsampleTo(torch.autograd.Function):
    """_upsample_match(up(low), skip) as a tensor of its own: the decoder convolution then reads [skip, up] as two loader
    sources (virtual concat).  ``two_step=True`` = the U-Net's x2 upsample followed by a second resize only if the sizes
    differ (src/model.py:219,243-246); ``False`` = U-Net++'s direct resize to the target (src/model.py:111-121)."""

    @staticmethod
    def forward(ctx, low, C_low, two_step, H, W):
        low = _as_nhwc(low)
        N, h, w, _ = low.shape
        code = dtype_code(low.dtype)
        stream = _stream()
        ld = pad8(C_low)
        src, sh, sw = low, h, w
        mid = None
        if two_step and (2 * h, 2 * w) != (H, W):
            mid = torch.empty((N, 2 * h, 2 * w, ld), dtype=low.dtype, device=low.device)
            call("mau_resize_bilinear_fwd", low.data_ptr(), _ld(low), h, w, mid.data_ptr(), ld, 0, code, N, 2 * h, 2 * w, C_low, stream)
            src, sh, sw = mid, 2 * h, 2 * w
        out = torch.empty((N, H, W, ld), dtype=low.dtype, device=low.device)
        call("mau_resize_bilinear_fwd", src.data_ptr(), _ld(src), sh, sw, out.data_ptr(), ld, 0, code, N, H, W, C_low, stream)
        ctx.meta = (C_low, (h, w), (H, W), mid is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        C_low, (h, w), (H, W), two = ctx.meta
        g = _as_nhwc(g)
        N = g.shape[0]
        code = dtype_code(g.dtype)
        stream = _stream()
        ld = pad8(C_low)
        gsrc, gld, sH, sW = g, _ld(g), H, W
        if two:
            dmid = torch.empty((N, 2 * h, 2 * w, ld), dtype=g.dtype, device=g.device)
            call("mau_resize_bilinear_bwd", g.data_ptr(), gld, 0, H, W, dmid.data_ptr(), ld, code, N, 2 * h, 2 * w, C_low, stream)
            gsrc, gld, sH, sW = dmid, ld, 2 * h, 2 * w
        dlow = torch.empty((N, h, w, ld), dtype=g.dtype, device=g.device)
        call("mau_resize_bilinear_bwd", gsrc.data_ptr(), gld, 0, sH, sW, dlow.data_ptr(), ld, code, N, h, w, C_low, stream)
        return dlow, None, None, None, None


class ConcatUp(torch.autograd.Function):
    """torch.cat([skip_0, ..., skip_{k-1}, up(low)], 1) with up = bilinear(align_corners=True).

    ``two_step=True`` reproduces the U-Net decoder (src/model.py:243-246,279-282): x2 upsample, then a
    second resize to the skip's size only if the sizes differ.  ``two_step=False`` is U-Net++'s
    direct resize to the target size (src/model.py:111-121).
    """

    @staticmethod
    def forward(ctx, low, C_low, two_step, skip_Cs, *skips):
        low = _as_nhwc(low)
        skips = [_as_nhwc(s) for s in skips]
        N, H, W, _ = skips[0].shape
        h, w = low.shape[1], low.shape[2]
        code = dtype_code(low.dtype)
        stream = _stream()
        Ctot = sum(skip_Cs) + C_low
        ld = pad8(Ctot)
        out = torch.empty((N, H, W, ld), dtype=low.dtype, device=low.device)
        choff = 0
        npix = N * H * W
        for s, cs in zip(skips, skip_Cs):
            call("mau_copy_channels", s.data_ptr(), _ld(s), out.data_ptr(), ld, choff, 0, code, npix, cs, stream)
            choff += cs
        mid = None
        src, sh, sw = low, h, w
        if two_step and (2 * h, 2 * w) != (H, W):
            mid = torch.empty((N, 2 * h, 2 * w, pad8(C_low)), dtype=low.dtype, device=low.device)
            call("mau_resize_bilinear_fwd", low.data_ptr(), _ld(low), h, w, mid.data_ptr(), pad8(C_low), 0, code, N, 2 * h, 2 * w, C_low, stream)
            src, sh, sw = mid, 2 * h, 2 * w
        _resize_into(src, C_low, sh, sw, out, choff, N, H, W, code, stream)
        if ld > Ctot:   # zero the pad channels
            out[..., Ctot:].zero_()
        ctx.meta = (C_low, tuple(skip_Cs), (h, w), (H, W), mid is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        C_low, skip_Cs, (h, w), (H, W), two = ctx.meta
        g = _as_nhwc(g)
        N = g.shape[0]
        code = dtype_code(g.dtype)
        stream = _stream()
        ldg = _ld(g)
        npix = N * H * W
        grads = []
        choff = 0
        for cs in skip_Cs:
            if cs % 8 == 0 and choff % 8 == 0:
                grads.append(g[..., choff:choff + cs])                     # zero-copy channel slice
            else:
                t = torch.empty((N, H, W, pad8(cs)), dtype=g.dtype, device=g.device)
                call("mau_copy_channels", g.data_ptr() + choff * g.element_size(), ldg,
                     t.data_ptr(), pad8(cs), 0, pad8(cs), code, npix, cs, stream)
                grads.append(t)
            choff += cs
        if choff % 8 == 0:
            gsrc, goff, gld = g, choff, ldg
        else:
            gsrc = torch.empty((N, H, W, pad8(C_low)), dtype=g.dtype, device=g.device)
            call("mau_copy_channels", g.data_ptr() + choff * g.element_size(), ldg, gsrc.data_ptr(), pad8(C_low), 0,
                 pad8(C_low), code, npix, C_low, stream)
            goff, gld = 0, pad8(C_low)
        if two:
            dmid = torch.empty((N, 2 * h, 2 * w, pad8(C_low)), dtype=g.dtype, device=g.device)
            call("mau_resize_bilinear_bwd", gsrc.data_ptr(), gld, goff, H, W, dmid.data_ptr(), pad8(C_low), code, N, 2 * h, 2 * w, C_low, stream)
            gsrc, goff, gld, sH, sW = dmid, 0, pad8(C_low), 2 * h, 2 * w
        else:
            sH, sW = H, W
        dlow = torch.empty((N, h, w, pad8(C_low)), dtype=g.dtype, device=g.device)
        call("mau_resize_bilinear_bwd", gsrc.data_ptr(), gld, goff, sH, sW, dlow.data_ptr(), pad8(C_low), code, N, h, w, C_low, stream)
        return (dlow, None, None, None, *grads)


class BcastCat(torch.autograd.Function):
    """Materialised cat([x, emb[:, :, None, None].expand(...)], 1) (src/model.py:248-259).

    Only used when channel counts do not fit the fused loader's 8-channel granularity
    (tiny test models); production shapes take the fused path inside mau_conv3x3_fwd.
    """

    @staticmethod
    def forward(ctx, x, C, emb):
        x = _as_nhwc(x)
        emb = emb.contiguous().float()
        N, H, W, _ = x.shape
        E = emb.shape[1]
        ld = pad8(C + E)
        code = dtype_code(x.dtype)
        out = torch.empty((N, H, W, ld), dtype=x.dtype, device=x.device)
        stream = _stream()
        call("mau_copy_channels", x.data_ptr(), _ld(x), out.data_ptr(), ld, 0, 0, code, N * H * W, C, stream)
        call("mau_bcast_fill", emb.data_ptr(), out.data_ptr(), ld, C, ld, code, N, H * W, E, stream)
        ctx.meta = (C, E)
        return out

    @staticmethod
    def backward(ctx, g):
        C, E = ctx.meta
        g = _as_nhwc(g)
        N, H, W, _ = g.shape
        code = dtype_code(g.dtype)
        stream = _stream()
        dx = torch.empty((N, H, W, pad8(C)), dtype=g.dtype, device=g.device)
        call("mau_copy_channels", g.data_ptr(), _ld(g), dx.data_ptr(), pad8(C), 0, pad8(C), code, N * H * W, C, stream)
        demb = torch.empty((N, E), dtype=torch.float32, device=g.device)
        ws = torch.empty(lib.mau_bcast_bwd_ws_elems(N, H * W, E), dtype=torch.float32, device=g.device)
        call("mau_bcast_bwd", g.data_ptr(), _ld(g), C, demb.data_ptr(), ws.data_ptr(), code, N, H * W, E, stream)
        return dx, None, demb


# --------------------------------------------------------------------------- #
# head
# --------------------------------------------------------------------------- #
class Head(torch.autograd.Function):
    """final 1x1 conv + tanh on channel 0 when out_channels == 2 (src/model.py:284-292 / :187-193).
    ``activate=False``: the bare 1x1 conv of the deep-supervision heads (src/model.py:180-185)."""

    @staticmethod
    def forward(ctx, a, C, weight, bias, activate=True):
        a = _as_nhwc(a)
        N, H, W, _ = a.shape
        Co = weight.shape[0]
        tanh0 = 1 if (Co == 2 and activate) else 0
        out = torch.empty((N, Co, H, W), dtype=torch.float32, device=a.device)
        w2 = weight.detach().reshape(Co, C).contiguous().float()
        call("mau_head_fwd", a.data_ptr(), _ld(a), w2.data_ptr(), bias.detach().data_ptr(), out.data_ptr(), tanh0,
             dtype_code(a.dtype), N, H * W, C, Co, _stream())
        ctx.save_for_backward(a, w2, out)
        ctx.meta = (C, Co, tanh0, tuple(weight.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w2, out = ctx.saved_tensors
        C, Co, tanh0, wshape = ctx.meta
        N, H, W, _ = a.shape
        dout = dout.contiguous().float()
        da = torch.empty((N, H, W, pad8(C)), dtype=a.dtype, device=a.device)
        rows, rowlen = lib.mau_head_bwd_rows(N, H * W), lib.mau_head_bwd_rowlen(C, Co)
        slab = torch.empty((rows, rowlen), dtype=torch.float32, device=a.device)   # (the 7 unwritten columns per output channel are never read back)
        stream = _stream()
        call("mau_head_bwd", a.data_ptr(), _ld(a), w2.data_ptr(), out.data_ptr(), dout.data_ptr(), da.data_ptr(), pad8(C),
             slab.data_ptr(), tanh0, dtype_code(a.dtype), N, H * W, C, Co, stream)
        red = torch.empty(rowlen, dtype=torch.float32, device=a.device)
        ws = torch.empty(lib.mau_reduce_rows_ws_elems(rows, rowlen), dtype=torch.float64, device=a.device)
        call("mau_reduce_rows_f32", slab.data_ptr(), rows, rowlen, rowlen, red.data_ptr(), ws.data_ptr(),
             _tickets(a.device).data_ptr() if _FUSED_REDUCE else None, stream)
        red = red.view(Co, pad8(C) + 8)
        dw = red[:, :C].reshape(wshape).contiguous()
        db = red[:, pad8(C)].contiguous()
        return da, None, dw, db, None


# --------------------------------------------------------------------------- #
# MetadataEncoder
# --------------------------------------------------------------------------- #
class MetaMLP(torch.autograd.Function):
    """Linear(F,32) -> ReLU -> Linear(32,D), src/model.py:38-48."""

    @staticmethod
    def forward(ctx, md, w0, b0, w2, b2):
        _require_cuda(md, "MetadataEncoder")
        md = md.contiguous().float()
        N, F = md.shape
        Hd, D = w0.shape[0], w2.shape[0]
        hidden = torch.empty((N, Hd), dtype=torch.float32, device=md.device)
        emb = torch.empty((N, D), dtype=torch.float32, device=md.device)
        call("mau_meta_mlp_fwd", md.data_ptr(), w0.detach().data_ptr(), b0.detach().data_ptr(), w2.detach().data_ptr(),
             b2.detach().data_ptr(), hidden.data_ptr(), emb.data_ptr(), N, F, Hd, D, _stream())
        ctx.save_for_backward(md, w0, w2, hidden)
        return emb

    @staticmethod
    def backward(ctx, demb):
        md, w0, w2, hidden = ctx.saved_tensors
        demb = demb.contiguous().float()
        N, F = md.shape
        Hd, D = w0.shape[0], w2.shape[0]
        dw0, db0 = torch.empty_like(w0), torch.empty(Hd, dtype=torch.float32, device=md.device)
        dw2, db2 = torch.empty_like(w2), torch.empty(D, dtype=torch.float32, device=md.device)
        ws = torch.empty((N, Hd), dtype=torch.float32, device=md.device)
        call("mau_meta_mlp_bwd", md.data_ptr(), w0.detach().data_ptr(), w2.detach().data_ptr(), hidden.data_ptr(),
             demb.data_ptr(), dw0.data_ptr(), db0.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), N, F, Hd, D,
             _stream())
        return None, dw0, db0, dw2, db2


# --------------------------------------------------------------------------- #
# TemporalEncoder recurrence
# --------------------------------------------------------------------------- #
class LSTMLast(torch.autograd.Function):
    """h_T of nn.LSTM(input_size=1, hidden_size=H, batch_first=True) over x (B,T) (src/model.py:26,30-31: ``_, (h_n, _) =
    self.lstm(x.unsqueeze(-1))``; ``h_n[-1]``), the whole sequence in one persistent launch per direction.
    Parameters in torch's layout: weight_ih_l0 (4H,1), weight_hh_l0 (4H,H), bias_ih_l0, bias_hh_l0 (4H)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        _require_cuda(x, "TemporalEncoder")
        x = x.contiguous().float()
        B, T = x.shape
        H = w_hh.shape[1]
        dev = x.device
        need = any(ctx.needs_input_grad[1:])
        h = torch.empty((B, H), dtype=torch.float32, device=dev)
        gates = torch.empty((B, T, 4 * H), dtype=torch.float32, device=dev) if need else None
        cells = torch.empty((B, T, H), dtype=torch.float32, device=dev) if need else None
        wi, wh = w_ih.detach().contiguous().float(), w_hh.detach().contiguous().float()
        call("mau_lstm_fwd", x.data_ptr(), wi.data_ptr(), wh.data_ptr(), b_ih.detach().data_ptr(), b_hh.detach().data_ptr(),
             h.data_ptr(), gates.data_ptr() if need else None, cells.data_ptr() if need else None, B, T, H, _stream())
        if need:
            ctx.save_for_backward(x, wh, gates, cells)
        ctx.shapes = (tuple(w_ih.shape), B, T, H)
        return h

    @staticmethod
    def backward(ctx, dh):
        x, wh, gates, cells = ctx.saved_tensors
        wshape, B, T, H = ctx.shapes
        dev = x.device
        dh = dh.contiguous().float()
        f32 = dict(dtype=torch.float32, device=dev)
        dw_ih, dw_hh = torch.empty(4 * H, **f32), torch.empty((4 * H, H), **f32)
        db_ih, db_hh = torch.empty(4 * H, **f32), torch.empty(4 * H, **f32)
        ws = torch.empty(lib.mau_lstm_bwd_ws_elems(B, T, H), **f32)
        call("mau_lstm_bwd", x.data_ptr(), wh.data_ptr(), gates.data_ptr(), cells.data_ptr(), dh.data_ptr(), dw_ih.data_ptr(),
             dw_hh.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(), ws.data_ptr(), B, T, H, _stream())
        return None, dw_ih.view(wshape), dw_hh, db_ih, db_hh            # (the series is data: no gradient w.r.t. x)


class Linear(torch.autograd.Function):
    """nn.Linear (TemporalEncoder.fc, src/model.py:27,34) on the HIP path: out = x w^T + b."""

    @staticmethod
    def forward(ctx, x, w, b):
        _require_cuda(x, "Linear")
        x = x.contiguous().float()
        N, F = x.shape
        D = w.shape[0]
        out = torch.empty((N, D), dtype=torch.float32, device=x.device)
        wc = w.detach().contiguous().float()
        call("mau_linear_fwd", x.data_ptr(), wc.data_ptr(), b.detach().data_ptr() if b is not None else None, out.data_ptr(), N, F, D, _stream())
        ctx.save_for_backward(x, wc)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wc = ctx.saved_tensors
        N, F = x.shape
        D = wc.shape[0]
        dout = dout.contiguous().float()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty_like(wc), torch.empty(D, dtype=torch.float32, device=x.device)
        call("mau_linear_bwd", x.data_ptr(), wc.data_ptr(), dout.data_ptr(), dx.data_ptr() if dx is not None else None, dw.data_ptr(),
             db.data_ptr(), N, F, D, _stream())
        return dx, dw, (db if ctx.has_bias else None)


# --------------------------------------------------------------------------- #
# MSE criterion
# --------------------------------------------------------------------------- #
class MSELoss(torch.autograd.Function):
    """F.mse_loss(outputs, targets) (src/utils/losses.py:33) with its gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, out, tgt):
        _require_cuda(out, "compute_loss_mse")
        out = out.contiguous().float()
        tgt = tgt.contiguous().float()
        n = out.numel()
        partial = torch.empty(lib.mau_mse_blocks(n), dtype=torch.float64, device=out.device)
        loss = torch.empty(1, dtype=torch.float32, device=out.device)
        dout = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        call("mau_mse_fwd_bwd", out.data_ptr(), tgt.data_ptr(), partial.data_ptr(), loss.data_ptr(),
             dout.data_ptr() if dout is not None else None, n, _stream())
        ctx.save_for_backward(dout)                      # kept across backward calls (retain_graph works)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g if d is not None else None), None


class L1GradientLoss(torch.autograd.Function):
    """(mean|o-t|, gradient_loss(o,t)) of src/utils/losses.py:5-25,68.  The two outputs are independent autograd values, as
    in the reference: backward returns ``g_l1 * d l1/d out + g_grad * d gradient_loss/d out`` for WHATEVER weights the
    caller combines them with (the two partial gradients are produced by two passes of the same kernel)."""

    @staticmethod
    def forward(ctx, out, tgt):
        _require_cuda(out, "gradient_loss")
        out = out.contiguous().float()
        tgt = tgt.contiguous().float()
        B, Cc, H, W = out.shape
        n = out.numel()
        need = ctx.needs_input_grad[0]
        partial = torch.empty(3 * lib.mau_l1_gradient_blocks(n), dtype=torch.float64, device=out.device)
        terms = torch.empty(3, dtype=torch.float32, device=out.device)
        d_l1 = torch.empty_like(out) if need else None
        d_gr = torch.empty_like(out) if need else None
        call("mau_l1_gradient_loss", out.data_ptr(), tgt.data_ptr(), partial.data_ptr(), terms.data_ptr(),
             d_l1.data_ptr() if need else None, 1.0, 0.0, B, Cc, H, W, _stream())
        if need:
            terms2 = torch.empty(3, dtype=torch.float32, device=out.device)
            call("mau_l1_gradient_loss", out.data_ptr(), tgt.data_ptr(), partial.data_ptr(), terms2.data_ptr(),
                 d_gr.data_ptr(), 0.0, 1.0, B, Cc, H, W, _stream())
        ctx.save_for_backward(d_l1, d_gr)
        return terms[0], terms[1] + terms[2]

    @staticmethod
    def backward(ctx, g_l1, g_grad):
        d_l1, d_gr = ctx.saved_tensors
        if d_l1 is None:
            return None, None
        return g_l1 * d_l1 + g_grad * d_gr, None
