"""Drop-in mirror of the reference's ``src/model.py`` module surface, running on libmau_hip.

Same class names, constructor signatures, ``forward(maps, temp_series, metadata)`` order,
parameter/buffer names (``state_dict`` is key- and shape-identical, so reference ``.pth``
checkpoints load with ``strict=True``) and error behaviour:

    UrbanPredictor            src/model.py:295-329
    UrbanPredictor_unet       src/model.py:195-292
    UrbanPredictor_unetpp     src/model.py:51-193
    VGGBlock                  src/model.py:9-21
    MetadataEncoder           src/model.py:38-48
    TemporalEncoder           src/model.py:23-34   (the LSTM recurrence is one persistent HIP launch per direction)

The ``torch.nn`` layers inside are parameter containers created in the reference's order (so the
same ``torch.manual_seed`` gives the same initial weights); their ``forward`` is never used for
the hot path -- all arithmetic runs in the HIP kernels behind ``functional``.

Extras that the reference does not have (all optional, defaults keep reference behaviour):
  * ``model.set_precision("bf16" | "fp16" | "fp32")`` -- bf16 throughput mode (default; env ``MAU_PRECISION``),
    the same kernels on fp16 operands (``v_mfma_f32_32x32x16_f16``, fp32 accumulation; inference, BASELINE
    configs[4]), or the fp32 parity mode;
  * ``model.set_sync_bn(process_group)``     -- BatchNorm statistics all-reduced over RCCL.
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import functional as F_
from .functional import Act, BNState

_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def _default_precision() -> str:
    p = os.environ.get("MAU_PRECISION", "bf16").lower()
    if p not in _DTYPES:
        raise ValueError(f"MAU_PRECISION must be one of {list(_DTYPES)}, got {p!r}")
    return p


class _Runtime:
    """Per-model execution settings shared by all blocks of one network."""

    def __init__(self):
        self.precision = _default_precision()
        self.group = None
        self.world = 1
        self.comm = None                  # dist.RcclComm for the SyncBN messages (None: local BN, or a group that is not on RCCL)
        # launch order inside a block's backward (functional.ConvBNReLU.backward): False = the weight-gradient branch's kernel is
        # enqueued before the main chain's data gradient, True = after it.  Measured per network ("launch order"):
        # the U-Net is 0.6 % faster with False, the U-Net++ 1.2 % faster with True; MAU_BWD_DGRAD_FIRST=0/1 overrides both.
        self.dgrad_first = False

    @property
    def dtype(self) -> torch.dtype:
        return _DTYPES[self.precision]


class VGGBlock(nn.Module):
    """(Conv3x3 -> BN -> ReLU) x 2, src/model.py:9-21."""

    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, middle_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(middle_channels)
        self.conv2 = nn.Conv2d(middle_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self._rt: Optional[_Runtime] = None
        self._group = None                # functional.PackGroup of the network this block belongs to (all weights re-packed in one launch)
        self._frozen = None               # [dict, dict] while a frozen inference session is active
        # a state-dict load through ANY container (nn.Module recursion calls _load_from_state_dict, not load_state_dict)
        # drops the frozen copies: they would keep serving the old weights and BatchNorm coefficients
        self._register_load_state_dict_pre_hook(self._drop_frozen)

    def _drop_frozen(self, *_args, **_kwargs):
        if self._frozen is not None:
            self._frozen = [{}, {}]

    def _half(self, x: Act, emb, conv: nn.Conv2d, bn: nn.BatchNorm2d, x1: Optional[Act] = None, pool: bool = False, out_view=None,
              head: Optional[nn.Conv2d] = None, head_act: bool = True, up_to=None, weight: Optional[torch.Tensor] = None):
        rt = self._rt or _Runtime()
        if weight is None and self._group is not None and getattr(conv.weight, "_mau_group", None) is not self._group:
            self._group.add(conv.weight)          # (a deep copy of the network carries new Parameter objects)
        st = BNState(training=self.training and bn.training, C0=x.C, momentum=bn.momentum, eps=bn.eps,
                     group=rt.group, comm=rt.comm, world=rt.world, grad_enabled=torch.is_grad_enabled(),
                     frozen=None if self._frozen is None else self._frozen[0 if conv is self.conv1 else 1],
                     C1=0 if x1 is None else x1.C, pool=pool, out_view=out_view,
                     head=None if head is None else bool(head_act), up_to=up_to, first=x.nchw, dtype=rt.dtype if x.nchw else None,
                     dgrad_first=rt.dgrad_first)
        # (``weight``: a tensor DERIVED from conv.weight that the convolution runs on instead -- functional.EmbFold's W_eff)
        t = F_.ConvBNReLU.apply(x.t, None if x1 is None else x1.t, emb, conv.weight if weight is None else weight, conv.bias, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                None if head is None else head.weight, None if head is None else head.bias, st)
        if head is not None:
            return t                                # (N, out_channels, H, W) fp32: the module's output
        if pool:
            return Act(t[0], conv.out_channels), Act(t[1], conv.out_channels)
        return Act(t, conv.out_channels)

    def _fold(self, emb: Optional[torch.Tensor], Ct: int, pixels: int, dtype):
        """Training: the broadcast embedding of conv1 folded into roundup(N, 16) indicator channels (functional.EmbFold) -- returns
        (embedding the loader broadcasts, weight conv1 runs on).  Eval keeps the plain form: a frozen inference session caches packs
        of the PARAMETERS, and the folded weight changes with every batch's embedding."""
        # (16-bit modes only: fp32 is the parity mode and keeps the reference's association of the sums)
        if emb is None or dtype == torch.float32 or not (self.training and torch.is_grad_enabled()):
            return emb, None
        folded = F_.fold_embedding(self.conv1.weight, emb, Ct, pixels)
        if folded is None:
            return emb, None
        return folded[1], folded[0]

    def forward(self, x: Act, emb: Optional[torch.Tensor] = None, x1: Optional[Act] = None, pool: bool = False, out_view=None,
                head: Optional[nn.Conv2d] = None, head_act: bool = True, up_to=None):
        """``x1``: second input tensor, channel-concatenated after ``x`` by the conv loader (never materialised);
        ``pool=True`` returns ``(block output, maxpool2x2(block output))``; ``out_view``: preallocated NHWC view the block's
        output is written into (a slot of a U-Net++ row buffer); ``head``: the network's final 1x1 conv -- the block returns
        ``final(block output)`` (tanh on channel 0 when ``head_act`` and out_channels == 2) and the block output itself is never
        written; ``up_to=(H, W)``: the block returns the bilinear (align_corners=True) resize of its output instead of it."""
        emb, w1 = self._fold(emb, x.C + (0 if x1 is None else x1.C), x.H * x.W, torch.float32 if x.nchw else x.t.dtype)
        x = self._half(x, emb, self.conv1, self.bn1, x1, weight=w1)
        return self._half(x, None, self.conv2, self.bn2, None, pool, out_view, head, head_act, up_to)


# Test hooks (module attributes, flipped by the tests with monkeypatch; decided A/B switches of earlier rounds, no longer environment variables)
_OVERLAP_LSTM = True          # the TemporalEncoder's recurrence on a side stream beside the first encoder blocks (training)
_OVERLAP_LSTM_GRAPH = True    # ... also inside a captured step (a second branch of the hipGraph)
_CONV_FIRST = True            # conv0_0.conv1 on the first-layer kernel (reads the fp32 NCHW input and the fp32 weights directly)
_VIRTUAL_CONCAT = True        # decoder concatenations as loader sources / U-Net++ row buffers instead of materialised tensors
_FANOUT = True                # U-Net++: one alias per reader of a row slot, the readers' gradients summed by one kernel

_SIDE_STREAMS = {}
_OVERLAP_OFF = [0]          # > 0: inside train_graph.GraphedTrainStep (its warm-up steps and the capture run on ONE stream)


class lstm_overlap_disabled:
    """Context: the TemporalEncoder stays on the current stream (A/B timing, single-stream debugging)."""

    def __enter__(self):
        _OVERLAP_OFF[0] += 1

    def __exit__(self, *exc):
        _OVERLAP_OFF[0] -= 1
        return False


def _overlap_lstm(t: torch.Tensor, training: bool) -> bool:
    """The TemporalEncoder's recurrence occupies one workgroup per sample (16-32 of the 256 CUs) for ~0.45 ms forward and
    ~0.9 ms backward at the reference's 828 steps: in training it runs on a side stream, beside the first encoder blocks
    (forward) and -- autograd replays a node on its forward stream -- beside the encoder's backward (also under a process
    group: dist.GradSync makes a bucket's launching stream wait for every stream that produced one of its gradients).
    Not in eval (hipGraph sessions, latency)."""
    if not (training and t.is_cuda) or _OVERLAP_OFF[0] > 0 or not _OVERLAP_LSTM:
        return False
    # inside a captured train step the side stream forks from / joins the capturing stream (wait_stream both ways): the
    # overlap becomes two branches of the hipGraph (MAU_OVERLAP_LSTM_GRAPH=0: one branch)
    return not torch.cuda.is_current_stream_capturing() or _OVERLAP_LSTM_GRAPH


class TemporalEncoder(nn.Module):
    """LSTM(1->hidden) last hidden state -> Linear, src/model.py:23-34."""

    def __init__(self, seq_len, hidden_dim, out_dim):
        super().__init__()
        self.lstm = nn.LSTM(input_size=1, hidden_size=hidden_dim, batch_first=True)
        self.fc = nn.Linear(hidden_dim, out_dim)

    def forward(self, x):
        lstm = self.lstm
        if lstm.hidden_size > F_.lib.mau_lstm_max_hidden():
            # (the reference's configurations use 96 and 32, conf/config.yaml / test/evaluate.py:152-160; there is no second
            #  backend behind this path: no nn.LSTM / MIOpen fallback)
            raise NotImplementedError(f"TemporalEncoder: lstm hidden size {lstm.hidden_size} exceeds the persistent HIP kernel's "
                                      f"limit of {F_.lib.mau_lstm_max_hidden()} (one gate row per thread)")
        # the whole (up to 828-step, conf/config.yaml:20) recurrence in one persistent HIP launch per direction
        h = F_.LSTMLast.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
        return F_.Linear.apply(h, self.fc.weight, self.fc.bias)

    def forward_async(self, x):
        """forward() on the side stream of x's device; returns (embedding, join): call join() on the consuming stream
        before the embedding's first use."""
        dev = x.device
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)                       # the series and the parameters (last optimizer step) are ready
        with torch.cuda.stream(side):
            out = self.forward(x)
        x.record_stream(side)

        def join():
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(side)
            out.record_stream(cur)
        return out, join


class MetadataEncoder(nn.Module):
    """Linear(F,32) -> ReLU -> Linear(32,out), src/model.py:38-48."""

    def __init__(self, in_features, out_dim):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(in_features, 32), nn.ReLU(), nn.Linear(32, out_dim))

    def forward(self, x):
        return F_.MetaMLP.apply(x, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias)


class _NetBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._rt = _Runtime()

    def _bind_runtime(self):
        self._pack_group = F_.PackGroup()
        for m in self.modules():
            if isinstance(m, VGGBlock):
                m._rt = self._rt
                m._group = self._pack_group
                self._pack_group.add(m.conv1.weight)
                self._pack_group.add(m.conv2.weight)

    # -- extras -------------------------------------------------------------
    def set_precision(self, precision: str):
        if precision not in _DTYPES:
            raise ValueError(f"precision must be one of {list(_DTYPES)}")
        self._rt.precision = precision
        self.freeze_inference(False)
        return self

    def freeze_inference(self, on: bool = True):
        """Eval-mode, no-grad forwards stop re-deriving what depends only on the parameters: the packed conv weights
        and the folded BatchNorm scale/shift are computed on the first frozen forward and reused (18 + 18 launches
        per forward saved: 17 % of the B=1 512x512 latency).  The parameters must not change while frozen;
        ``train()``, ``load_state_dict`` and ``set_precision`` unfreeze.  ``GraphedInference`` freezes its model."""
        for m in self.modules():
            if isinstance(m, VGGBlock):
                m._frozen = [{}, {}] if on else None
        return self

    def train(self, mode: bool = True):
        if mode:
            self.freeze_inference(False)
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.freeze_inference(False)
        return super().load_state_dict(*args, **kwargs)

    def set_sync_bn(self, group=None, world_size: Optional[int] = None, direct: Optional[bool] = None):
        """All-reduce BatchNorm batch statistics over ``group`` (RCCL); ``None``/world 1 = local BN.
        Collective: every rank of the group must call it (it creates the group's SyncBN communicator, ``dist.rccl_comm``).
        ``direct``: RCCL called directly on the compute stream (None = ``MAU_RCCL_DIRECT``, default on) or ProcessGroupNCCL."""
        import torch.distributed as dist
        if group is None and not (dist.is_available() and dist.is_initialized()):
            self._rt.group, self._rt.world, self._rt.comm = None, 1, None
            return self
        from .dist import rccl_comm
        g = group if group is not None else dist.group.WORLD
        self._rt.group = g
        self._rt.world = world_size if world_size is not None else dist.get_world_size(g)
        on_gpu = next(self.parameters()).is_cuda
        self._rt.comm = rccl_comm(g, "bn", direct) if on_gpu else None
        return self

    def _entry(self, maps) -> Act:
        """maps: (B,C,H,W) fp32 as the reference's collate_fn delivers it, or an ``Act`` already in the network's
        layout and dtype (``mau_amd.data``: one-hot channels generated on the device)."""
        if isinstance(maps, Act):
            if maps.t.dtype != self._rt.dtype:
                raise RuntimeError(f"packed input is {maps.t.dtype}, the network computes in {self._rt.dtype}")
            return maps
        # 16-bit modes, at most 8 input channels (BASELINE configs: 6), no gradient w.r.t. the input: the first convolution reads the
        # tensor as it is (mau_conv3x3_first_fwd) -- no NCHW -> NHWC layout kernel, no packed copy of its weights.  MAU_CONV_FIRST=0: A/B.
        if (self._rt.dtype != torch.float32 and maps.dim() == 4 and maps.shape[1] <= F_.lib.mau_conv3x3_first_max_channels()
                and not (maps.requires_grad and torch.is_grad_enabled()) and _CONV_FIRST):
            F_._require_cuda(maps, "UrbanPredictor.forward(maps)")
            return Act(maps, maps.shape[1], nchw=True)
        return Act(F_.ToNHWC.apply(maps, self._rt.dtype), maps.shape[1])

    def _block_pool(self, block: VGGBlock, x: Act, out_view=None):
        """(pool(block(x)), block(x)): the encoder block's second BatchNorm+ReLU pass also writes the pooled tensor, and
        its backward adds the pool's and the skip connection's gradients in one pass."""
        a, p = block(x, pool=True, out_view=out_view)
        return p, a

    def _fusable(self, skip: Act) -> bool:
        """Virtual concat needs a 16-bit activation dtype and a first tensor that ends on a 16-channel stage boundary."""
        return self._rt.dtype != torch.float32 and skip.C % 16 == 0 and _VIRTUAL_CONCAT

    def _pool(self, a: Act) -> Act:
        return Act(F_.MaxPool2x2.apply(a.t, a.C), a.C)

    def _head(self, a: Act) -> torch.Tensor:
        return F_.Head.apply(a.t, a.C, self.final.weight, self.final.bias)


class UrbanPredictor_unet(_NetBase):
    """src/model.py:195-292."""

    def __init__(self, spatial_channels, seq_len, temporal_dim, meta_features, meta_dim, lstm_dim, out_channels,
                 nb_filter=None, temporal_embeddings=True, metadata_embeddings=True):
        super().__init__()
        # the reference announces the construction on stdout and through loguru (src/model.py:202-203); same line, through
        # print and the standard `logging` module (loguru is not a dependency of this package).  MAU_QUIET=1 silences the print.
        msg = f'UrbanPredictor_unet initialized with temporal_embeddings={temporal_embeddings}, metadata_embeddings={metadata_embeddings}'
        logging.getLogger("mau_amd").info(msg)
        if os.environ.get("MAU_QUIET", "0") != "1":
            print(msg)
        if nb_filter is None:
            nb_filter = [32, 64, 128, 256, 512]
        self.temporal_dim = temporal_dim
        self.meta_dim = meta_dim
        self.temporal_embeddings = temporal_embeddings
        self.metadata_embeddings = metadata_embeddings
        self.temporal_encoder = TemporalEncoder(seq_len, hidden_dim=lstm_dim, out_dim=temporal_dim)
        self.meta_encoder = MetadataEncoder(meta_features, meta_dim)
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv0_0 = VGGBlock(spatial_channels, nb_filter[0], nb_filter[0])
        self.conv1_0 = VGGBlock(nb_filter[0], nb_filter[1], nb_filter[1])
        self.conv2_0 = VGGBlock(nb_filter[1], nb_filter[2], nb_filter[2])
        self.conv3_0 = VGGBlock(nb_filter[2], nb_filter[3], nb_filter[3])
        bottleneck_in = nb_filter[3]
        if self.temporal_embeddings:
            bottleneck_in += temporal_dim
        if self.metadata_embeddings:
            bottleneck_in += meta_dim
        self.conv4_0 = VGGBlock(bottleneck_in, nb_filter[4], nb_filter[4])
        self.conv3_1 = VGGBlock(nb_filter[3] + nb_filter[4], nb_filter[3], nb_filter[3])
        self.conv2_1 = VGGBlock(nb_filter[2] + nb_filter[3], nb_filter[2], nb_filter[2])
        self.conv1_1 = VGGBlock(nb_filter[1] + nb_filter[2], nb_filter[1], nb_filter[1])
        self.conv0_1 = VGGBlock(nb_filter[0] + nb_filter[1], nb_filter[0], nb_filter[0])
        self.final = nn.Conv2d(nb_filter[0], out_channels, kernel_size=1)
        self._bind_runtime()

    def _up_target(self, skip: Act, low_hw):
        """(H, W) when ``up(low)`` for this skip can be produced by the block that computes ``low`` (its only consumer is the
        upsample: bottleneck and decoder blocks, src/model.py:279-282): the virtual concat is in use and the x2 upsample
        lands exactly on the skip's size (no second resize, src/model.py:243-246); else None."""
        if self._fusable(skip) and (2 * low_hw[0], 2 * low_hw[1]) == (skip.H, skip.W):
            return (skip.H, skip.W)
        return None

    def _dec(self, block: VGGBlock, skip: Act, low: Act, low_is_up: bool = False, **kw):
        # block(cat([skip, _upsample_match(up(low), skip)], 1)), src/model.py:243-246,279-282
        if low_is_up:                                    # ``low`` already is up(low): produced by the block below (VGGBlock up_to)
            return block(skip, None, low, **kw)
        if self._fusable(skip):
            up = Act(F_.UpsampleTo.apply(low.t, low.C, True, skip.H, skip.W), low.C)
            return block(skip, None, up, **kw)           # [skip, up] are two sources of the conv loader: no concat buffer
        t = F_.ConcatUp.apply(low.t, low.C, True, (skip.C,), skip.t)
        return block(Act(t, skip.C + low.C), **kw)

    def _fused_block(self, block: VGGBlock, x: Act, embs: List[torch.Tensor], **kw) -> Act:
        """conv4_0(fuse_embeddings(x, ...)) with the broadcast folded into the conv loader (src/model.py:248-259)."""
        if not embs:
            return block(x, **kw)
        emb = embs[0] if len(embs) == 1 else torch.cat(embs, dim=1)     # order [temporal, meta]
        if x.C % 8 == 0 and emb.shape[1] % 8 == 0:
            return block(x, emb.float(), **kw)
        # channel counts that do not fit the 8-channel vector granularity: materialise the concat
        return block(Act(F_.BcastCat.apply(x.t, x.C, emb), x.C + emb.shape[1]), **kw)

    def forward(self, maps, temp_series, metadata):
        join = None
        if self.temporal_embeddings and _overlap_lstm(temp_series, self.training):
            temporal_emb, join = self.temporal_encoder.forward_async(temp_series)
        else:
            temporal_emb = self.temporal_encoder(temp_series) if self.temporal_embeddings else None
        meta_emb = self.meta_encoder(metadata) if self.metadata_embeddings else None
        x = self._entry(maps)
        p, x0_0 = self._block_pool(self.conv0_0, x)
        p, x1_0 = self._block_pool(self.conv1_0, p)
        p, x2_0 = self._block_pool(self.conv2_0, p)
        x4_0, x3_0 = self._block_pool(self.conv3_0, p)
        if join is not None:
            join()
        # A bottleneck / decoder block's output is read ONLY by the x2 upsample of the level above: the block returns
        # up(output) directly (BatchNorm + ReLU applied by the resize kernel; the low-resolution activation is never written)
        embs = [e for e in (temporal_emb, meta_emb) if e is not None]
        skips = (x3_0, x2_0, x1_0, x0_0)
        blocks = (self.conv3_1, self.conv2_1, self.conv1_1, self.conv0_1)
        tgt = self._up_target(skips[0], (x4_0.H, x4_0.W))
        low = self._fused_block(self.conv4_0, x4_0, embs, up_to=tgt)
        for i, (blk, skip) in enumerate(zip(blocks, skips)):
            was_up, h, w = tgt is not None, skip.H, skip.W
            if i == 3:                                   # the last block feeds the 1x1 head: it returns final(output)
                return self._dec(blk, skip, low, was_up, head=self.final)
            tgt = self._up_target(skips[i + 1], (h, w))
            low = self._dec(blk, skip, low, was_up, up_to=tgt)

    @torch.no_grad()
    def forward_metadata_sweep(self, maps, temp_series, metadata):
        """Eval-mode sweep over B metadata vectors for ONE tile (test/metadata_sensitivity.py:294-311, :408-419 repeat
        the tile B times and vary only the metadata).  The encoder (src/model.py:267-273) does not see the
        embeddings, so it runs once at batch 1; only the bottleneck and decoder run at batch B.
        maps (1,C,H,W); temp_series (1,T) or (B,T); metadata (B,F).  Returns (B,out_channels,H,W), identical to
        ``self(maps.expand(B,...), temp_series.expand(B,...), metadata)`` in eval mode.
        """
        if self.training:
            raise RuntimeError("forward_metadata_sweep is an eval-mode path (BatchNorm must use running statistics)")
        if maps.shape[0] != 1:
            raise ValueError("forward_metadata_sweep expects a single tile: maps.shape[0] == 1")
        B = metadata.shape[0]
        if temp_series.shape[0] == 1 and B > 1:
            temp_series = temp_series.expand(B, -1)
        temporal_emb = self.temporal_encoder(temp_series) if self.temporal_embeddings else None
        meta_emb = self.meta_encoder(metadata) if self.metadata_embeddings else None
        x = self._entry(maps)
        x0_0 = self.conv0_0(x)
        x1_0 = self.conv1_0(self._pool(x0_0))
        x2_0 = self.conv2_0(self._pool(x1_0))
        x3_0 = self.conv3_0(self._pool(x2_0))
        x4_0 = self._pool(x3_0)

        def rep(a: Act) -> Act:          # batch broadcast of an encoder activation (data movement only)
            return Act(a.t.expand(B, -1, -1, -1).contiguous(), a.C)

        x0_0, x1_0, x2_0, x3_0, x4_0 = rep(x0_0), rep(x1_0), rep(x2_0), rep(x3_0), rep(x4_0)
        x4_0 = self._fused_block(self.conv4_0, x4_0, [e for e in (temporal_emb, meta_emb) if e is not None])
        x3_1 = self._dec(self.conv3_1, x3_0, x4_0)
        x2_1 = self._dec(self.conv2_1, x2_0, x3_1)
        x1_1 = self._dec(self.conv1_1, x1_0, x2_1)
        x0_1 = self._dec(self.conv0_1, x0_0, x1_1)
        return self._head(x0_1)


class UrbanPredictor_unetpp(_NetBase):
    """src/model.py:51-193 (nested U-Net; embeddings concatenated into every decoder node)."""

    def __init__(self, spatial_channels, seq_len, temporal_dim, meta_features, meta_dim, lstm_dim, out_channels,
                 base_filters=32, deep_supervision=False, **kwargs):
        super().__init__()
        self._rt.dgrad_first = True           # (launch order of a block's backward: _Runtime.dgrad_first)
        nb = [base_filters, base_filters * 2, base_filters * 4, base_filters * 8, base_filters * 16]
        self.deep_supervision = deep_supervision
        self.pool = nn.MaxPool2d(2, 2)
        self.embed_dim = temporal_dim + meta_dim
        e = temporal_dim + meta_dim
        self.conv0_0 = VGGBlock(spatial_channels, nb[0], nb[0])
        self.conv1_0 = VGGBlock(nb[0], nb[1], nb[1])
        self.conv2_0 = VGGBlock(nb[1], nb[2], nb[2])
        self.conv3_0 = VGGBlock(nb[2], nb[3], nb[3])
        self.conv4_0 = VGGBlock(nb[3], nb[4], nb[4])
        self.conv0_1 = VGGBlock(nb[0] + nb[1] + e, nb[0], nb[0])
        self.conv1_1 = VGGBlock(nb[1] + nb[2] + e, nb[1], nb[1])
        self.conv2_1 = VGGBlock(nb[2] + nb[3] + e, nb[2], nb[2])
        self.conv3_1 = VGGBlock(nb[3] + nb[4] + e, nb[3], nb[3])
        self.conv0_2 = VGGBlock(nb[0] * 2 + nb[1] + e, nb[0], nb[0])
        self.conv1_2 = VGGBlock(nb[1] * 2 + nb[2] + e, nb[1], nb[1])
        self.conv2_2 = VGGBlock(nb[2] * 2 + nb[3] + e, nb[2], nb[2])
        self.conv0_3 = VGGBlock(nb[0] * 3 + nb[1] + e, nb[0], nb[0])
        self.conv1_3 = VGGBlock(nb[1] * 3 + nb[2] + e, nb[1], nb[1])
        self.conv0_4 = VGGBlock(nb[0] * 4 + nb[1] + e, nb[0], nb[0])
        self.temporal_encoder = TemporalEncoder(seq_len, hidden_dim=lstm_dim, out_dim=temporal_dim)
        self.meta_encoder = MetadataEncoder(meta_features, meta_dim)
        if self.deep_supervision:
            self.final1 = nn.Conv2d(nb[0], out_channels, kernel_size=1)
            self.final2 = nn.Conv2d(nb[0], out_channels, kernel_size=1)
            self.final3 = nn.Conv2d(nb[0], out_channels, kernel_size=1)
            self.final4 = nn.Conv2d(nb[0], out_channels, kernel_size=1)
        else:
            self.final = nn.Conv2d(nb[0], out_channels, kernel_size=1)
        self._bind_runtime()

    def _node(self, block: VGGBlock, skips: List[Act], below: Act, emb: torch.Tensor, out_view=None, rows: bool = False, head=None):
        # cat([skips..., _upsample_match(below, (H, W)), emb_map], 1), src/model.py:111-121,136-177
        # rows: the skips sit side by side in one row buffer (an argument, not module state: forward stays re-entrant)
        fused_emb = (sum(s.C for s in skips) + below.C) % 8 == 0 and emb.shape[1] % 8 == 0
        rows = rows or out_view is not None
        if fused_emb and self._fusable(skips[0]) and (len(skips) == 1 or rows):
            # [skips | up | broadcast(emb)] are three sources of the conv loader; with row buffers the skips of a row
            # already sit side by side (RowPrefix: a view, no copy)
            first = skips[0] if len(skips) == 1 else Act(F_.RowPrefix.apply(skips[0].C, *[s.t for s in skips]), sum(s.C for s in skips))
            up = Act(F_.UpsampleTo.apply(below.t, below.C, False, skips[0].H, skips[0].W), below.C)
            return block(first, emb, up, out_view=out_view, head=head)
        t = F_.ConcatUp.apply(below.t, below.C, False, tuple(s.C for s in skips), *[s.t for s in skips])
        x = Act(t, sum(s.C for s in skips) + below.C)
        if x.C % 8 == 0 and emb.shape[1] % 8 == 0:
            return block(x, emb, out_view=out_view, head=head)       # broadcast embedding folded into the conv loader
        return block(Act(F_.BcastCat.apply(x.t, x.C, emb), x.C + emb.shape[1]), out_view=out_view, head=head)

    def forward(self, maps, temp_series, metadata):
        join = None
        if _overlap_lstm(temp_series, self.training):
            temporal_emb, join = self.temporal_encoder.forward_async(temp_series)
        else:
            temporal_emb = self.temporal_encoder(temp_series)
        meta_emb = self.meta_encoder(metadata)
        x = self._entry(maps)
        # Row buffers: the nodes x^{i,0..} of one resolution are written side by side into one (N,H,W,slots*C) buffer, so
        # that "cat of the earlier nodes of the row" is a view.  Needs 64-channel-aligned blocks and a 16-bit dtype.
        nb0 = self.conv0_0.conv2.out_channels
        # The buffers are written by raw kernels through ``BNState.out_view`` and read through views: no torch in-place op may
        # ever touch them (it would bump the version counter autograd checks for the saved slots).
        use_rows = self._rt.dtype != torch.float32 and nb0 % 64 == 0 and _VIRTUAL_CONCAT
        N, H, W = x.N, x.H, x.W
        hs, ws = [H], [W]
        for _ in range(3):
            hs.append(hs[-1] // 2)
            ws.append(ws[-1] // 2)

        def row_hw(level, slots):
            if not use_rows:
                return [None] * slots
            C = nb0 << level
            buf = torch.empty((N, hs[level], ws[level], slots * C), dtype=self._rt.dtype, device=x.t.device)
            return [buf[..., j * C:(j + 1) * C] for j in range(slots)]

        r0, r1, r2 = row_hw(0, 4), row_hw(1, 3), row_hw(2, 2)
        # Every node but the last of a row is read several times -- by the later nodes of its row and by the node above it
        # (src/model.py:136-177).  Each reader gets its own alias (functional.Fanout): the readers' gradients are then summed by one
        # kernel in the alias node's backward instead of by autograd's generic strided adds (16 launches, 0.85 ms of a B=16 step).
        fan_on = self.training and torch.is_grad_enabled() and _FANOUT
        ds = 1 if self.deep_supervision else 0            # (the deep-supervision heads read x^{0,1..3} once more)

        class Fan:
            def __init__(self, a: Act, k: int):
                self.items = [Act(t, a.C) for t in F_.Fanout.apply(a.t, k)] if (fan_on and k > 1) else [a] * k
                self.i = 0

            def take(self) -> Act:
                self.i += 1
                return self.items[self.i - 1]

        p, x0_0 = self._block_pool(self.conv0_0, x, r0[0])         # (the skip feeds every node of the row)
        p, x1_0 = self._block_pool(self.conv1_0, p, r1[0])
        x0_0, x1_0 = Fan(x0_0, 4), Fan(x1_0, 4)
        if join is not None:
            join()
        emb = torch.cat([temporal_emb, meta_emb], dim=1).float()           # src/model.py:103
        x0_1 = Fan(self._node(self.conv0_1, [x0_0.take()], x1_0.take(), emb, r0[1]), 3 + ds)
        p, x2_0 = self._block_pool(self.conv2_0, p, r2[0])
        x2_0 = Fan(x2_0, 3)
        x1_1 = Fan(self._node(self.conv1_1, [x1_0.take()], x2_0.take(), emb, r1[1]), 3)
        x0_2 = Fan(self._node(self.conv0_2, [x0_0.take(), x0_1.take()], x1_1.take(), emb, r0[2]), 2 + ds)
        p, x3_0 = self._block_pool(self.conv3_0, p)
        x3_0 = Fan(x3_0, 2)
        x2_1 = Fan(self._node(self.conv2_1, [x2_0.take()], x3_0.take(), emb, r2[1]), 2)
        x1_2 = Fan(self._node(self.conv1_2, [x1_0.take(), x1_1.take()], x2_1.take(), emb, r1[2]), 2)
        x0_3 = Fan(self._node(self.conv0_3, [x0_0.take(), x0_1.take(), x0_2.take()], x1_2.take(), emb, r0[3]), 1 + ds)
        x4_0 = self.conv4_0(p)
        x3_1 = self._node(self.conv3_1, [x3_0.take()], x4_0, emb, rows=use_rows)
        x2_2 = self._node(self.conv2_2, [x2_0.take(), x2_1.take()], x3_1, emb, rows=use_rows)
        x1_3 = self._node(self.conv1_3, [x1_0.take(), x1_1.take(), x1_2.take()], x2_2, emb, rows=use_rows)
        last = [x0_0.take(), x0_1.take(), x0_2.take(), x0_3.take()]
        if self.deep_supervision:
            x0_4 = self._node(self.conv0_4, last, x1_3, emb, rows=use_rows)
            return [F_.Head.apply(a.t, a.C, f.weight, f.bias, False)          # bare 1x1 convs, no tanh (src/model.py:180-185)
                    for a, f in ((x0_1.take(), self.final1), (x0_2.take(), self.final2), (x0_3.take(), self.final3), (x0_4, self.final4))]
        # x^{0,4} is read only by the 1x1 head: the block returns final(x^{0,4}) (src/model.py:187-193)
        return self._node(self.conv0_4, last, x1_3, emb, rows=use_rows, head=self.final)


class UrbanPredictor(nn.Module):
    """Dispatcher, src/model.py:295-329 (same signature; unknown ``model_type`` raises ValueError)."""

    def __init__(self, model_type, spatial_channels, seq_len, temporal_dim, meta_features, meta_dim, lstm_dim,
                 out_channels, base_filters=64, deep_supervision=False, **kwargs):
        super().__init__()
        if model_type == 'unet++':
            self.model = UrbanPredictor_unetpp(
                spatial_channels=spatial_channels, seq_len=seq_len, temporal_dim=temporal_dim,
                meta_features=meta_features, meta_dim=meta_dim, lstm_dim=lstm_dim, out_channels=out_channels,
                base_filters=base_filters, deep_supervision=deep_supervision, **kwargs)
        elif model_type == 'unet':
            self.model = UrbanPredictor_unet(
                spatial_channels=spatial_channels, seq_len=seq_len, temporal_dim=temporal_dim,
                meta_features=meta_features, meta_dim=meta_dim, lstm_dim=lstm_dim, out_channels=out_channels,
                nb_filter=[base_filters, base_filters * 2, base_filters * 4, base_filters * 8, base_filters * 16],
                **kwargs)
        else:
            raise ValueError(f"Unsupported model_type: {model_type}")

    def forward(self, maps, temp_series, metadata):
        return self.model(maps, temp_series, metadata)

    # extras forwarded to the network
    def forward_metadata_sweep(self, maps, temp_series, metadata):
        """One tile x B metadata vectors with the encoder computed once (U-Net only; see UrbanPredictor_unet)."""
        if not hasattr(self.model, "forward_metadata_sweep"):
            raise NotImplementedError("encoder reuse needs a metadata-independent encoder: model_type 'unet' only")
        return self.model.forward_metadata_sweep(maps, temp_series, metadata)

    def set_precision(self, precision: str):
        self.model.set_precision(precision)
        return self

    def freeze_inference(self, on: bool = True):
        self.model.freeze_inference(on)
        return self

    def load_state_dict(self, *args, **kwargs):
        self.model.freeze_inference(False)
        return super().load_state_dict(*args, **kwargs)

    def set_sync_bn(self, group=None, world_size=None, direct=None):
        self.model.set_sync_bn(group, world_size, direct)
        return self
