"""TEST INFRASTRUCTURE ONLY -- functional fp32 CPU restatement of the reference hot path.

Every function restates one item of the reference's ``src/model.py`` /
``src/train.py`` / ``src/utils/losses.py`` (cited as file:line, relative to the
reference repo root) on plain ``torch`` CPU tensors, NCHW fp32, operating on a
flat ``state_dict``-style mapping instead of ``nn.Module`` objects.  It is the
checker for the HIP path and the timed ``cpu_baseline`` of ``bench.py``; it is
never imported by the product package.

Parity: PINNED by ``tests/golden/*.npz`` (made by ``tests/golden/make_golden.py``
from the imported reference) -- see ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, src/model.py:13,15
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default, src/model.py:13,15

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- #
# construction (parameter names, shapes, default init in the reference's order)
# --------------------------------------------------------------------------- #
def _vgg_names(cin: int, cmid: int, cout: int):
    # src/model.py:9-16 -- conv1, bn1, conv2, bn2 are created in this order
    return [("conv1", nn.Conv2d(cin, cmid, 3, padding=1)), ("bn1", nn.BatchNorm2d(cmid)),
            ("conv2", nn.Conv2d(cmid, cout, 3, padding=1)), ("bn2", nn.BatchNorm2d(cout))]


def _add(sd: State, prefix: str, mod: nn.Module):
    for k, v in mod.state_dict().items():
        sd[f"{prefix}.{k}"] = v.detach().clone()


def _add_vgg(sd: State, prefix: str, cin: int, cmid: int, cout: int):
    for name, mod in _vgg_names(cin, cmid, cout):
        _add(sd, f"{prefix}.{name}", mod)


def init_state(model_type: str, spatial_channels: int, seq_len: int, temporal_dim: int,
               meta_features: int, meta_dim: int, lstm_dim: int, out_channels: int,
               base_filters: int = 64, temporal_embeddings: bool = True,
               metadata_embeddings: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Default-initialised parameters/buffers with the reference's key names.

    Consumes the global torch RNG in exactly the order of the reference's
    constructors (src/model.py:52-96 for ``unet++``, :196-241 for ``unet``,
    dispatcher :295-326), so ``torch.manual_seed(s); init_state(...)`` equals the
    reference's ``torch.manual_seed(s); UrbanPredictor(...).state_dict()``
    (key *order* differs; key set and values are identical).
    """
    nb = [base_filters * m for m in (1, 2, 4, 8, 16)]          # src/model.py:54,322
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def encoders():
        # TemporalEncoder src/model.py:23-27, MetadataEncoder :38-45
        _add(sd, "model.temporal_encoder.lstm", nn.LSTM(input_size=1, hidden_size=lstm_dim, batch_first=True))
        _add(sd, "model.temporal_encoder.fc", nn.Linear(lstm_dim, temporal_dim))
        _add(sd, "model.meta_encoder.fc.0", nn.Linear(meta_features, 32))
        _add(sd, "model.meta_encoder.fc.2", nn.Linear(32, meta_dim))

    if model_type == "unet":
        encoders()                                              # :213-216 (before the convs)
        _add_vgg(sd, "model.conv0_0", spatial_channels, nb[0], nb[0])
        _add_vgg(sd, "model.conv1_0", nb[0], nb[1], nb[1])
        _add_vgg(sd, "model.conv2_0", nb[1], nb[2], nb[2])
        _add_vgg(sd, "model.conv3_0", nb[2], nb[3], nb[3])
        bott = nb[3] + (temporal_dim if temporal_embeddings else 0) + (meta_dim if metadata_embeddings else 0)
        _add_vgg(sd, "model.conv4_0", bott, nb[4], nb[4])       # :228-233
        _add_vgg(sd, "model.conv3_1", nb[3] + nb[4], nb[3], nb[3])
        _add_vgg(sd, "model.conv2_1", nb[2] + nb[3], nb[2], nb[2])
        _add_vgg(sd, "model.conv1_1", nb[1] + nb[2], nb[1], nb[1])
        _add_vgg(sd, "model.conv0_1", nb[0] + nb[1], nb[0], nb[0])
        _add(sd, "model.final", nn.Conv2d(nb[0], out_channels, kernel_size=1))
    elif model_type == "unet++":
        e = temporal_dim + meta_dim                             # :60
        for i in range(5):                                      # :64-68
            _add_vgg(sd, f"model.conv{i}_0", spatial_channels if i == 0 else nb[i - 1], nb[i], nb[i])
        for j in range(1, 5):                                   # :71-83 (creation order: by column j, row i)
            for i in range(0, 5 - j):
                _add_vgg(sd, f"model.conv{i}_{j}", nb[i] * j + nb[i + 1] + e, nb[i], nb[i])
        # the reference creates conv0_1,conv1_1,conv2_1,conv3_1, conv0_2,... which is the loop order above
        encoders()                                              # :86-87 (after the convs)
        _add(sd, "model.final", nn.Conv2d(nb[0], out_channels, kernel_size=1))
    else:
        raise ValueError(f"Unsupported model_type: {model_type}")   # src/model.py:326
    return sd


def clone_state(sd: State, requires_grad: bool = False) -> State:
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


def is_param(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


# --------------------------------------------------------------------------- #
# operators
# --------------------------------------------------------------------------- #
def conv_bn_relu(x, sd: State, conv: str, bn: str, training: bool):
    """One ``relu(bn(conv(x)))`` of VGGBlock.forward (src/model.py:18-21)."""
    y = F.conv2d(x, sd[f"{conv}.weight"], sd[f"{conv}.bias"], padding=1)          # :12,14
    if training:
        sd[f"{bn}.num_batches_tracked"] += 1                                       # nn.BatchNorm2d.forward
    y = F.batch_norm(y, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"], sd[f"{bn}.weight"],
                     sd[f"{bn}.bias"], training, BN_MOMENTUM, BN_EPS)              # :13,15
    return F.relu(y)                                                               # :16


def vgg_block(x, sd: State, prefix: str, training: bool):
    """VGGBlock.forward, src/model.py:18-21."""
    x = conv_bn_relu(x, sd, f"{prefix}.conv1", f"{prefix}.bn1", training)
    return conv_bn_relu(x, sd, f"{prefix}.conv2", f"{prefix}.bn2", training)


def maxpool2x2(x):
    """nn.MaxPool2d(2, 2), src/model.py:57,218 (floor mode)."""
    return F.max_pool2d(x, 2, 2)


def upsample2x(x):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=True), src/model.py:219."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def resize_bilinear(x, size):
    """F.interpolate(size=..., bilinear, align_corners=True), src/model.py:121,245."""
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)


def unet_upsample_match(low, target):
    """``_upsample_match(self.up(low), target)`` of the U-Net decoder, src/model.py:243-246,279-282."""
    up = upsample2x(low)
    if up.shape[2:] != target.shape[2:]:
        up = resize_bilinear(up, target.shape[2:])
    return up


def metadata_encoder(md, sd: State, prefix: str = "model.meta_encoder"):
    """MetadataEncoder.forward, src/model.py:47-48 (Linear -> ReLU -> Linear, :41-45)."""
    h = F.relu(F.linear(md, sd[f"{prefix}.fc.0.weight"], sd[f"{prefix}.fc.0.bias"]))
    return F.linear(h, sd[f"{prefix}.fc.2.weight"], sd[f"{prefix}.fc.2.bias"])


def temporal_encoder(ts, sd: State, prefix: str = "model.temporal_encoder"):
    """TemporalEncoder.forward, src/model.py:29-34: single-layer LSTM(1->H), last hidden -> Linear."""
    w_ih, w_hh = sd[f"{prefix}.lstm.weight_ih_l0"], sd[f"{prefix}.lstm.weight_hh_l0"]
    b_ih, b_hh = sd[f"{prefix}.lstm.bias_ih_l0"], sd[f"{prefix}.lstm.bias_hh_l0"]
    B, T = ts.shape
    H = w_hh.shape[1]
    h = ts.new_zeros(B, H)
    c = ts.new_zeros(B, H)
    for t in range(T):                                       # gate order i, f, g, o (torch.nn.LSTM)
        g = F.linear(ts[:, t:t + 1], w_ih, b_ih) + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
    return F.linear(h, sd[f"{prefix}.fc.weight"], sd[f"{prefix}.fc.bias"])


def fuse_embeddings(spatial, temporal_emb, meta_emb):
    """UrbanPredictor_unet.fuse_embeddings, src/model.py:248-259 -- order [spatial, temporal, meta]."""
    B, _, H, W = spatial.shape
    parts = [spatial]
    for emb in (temporal_emb, meta_emb):
        if emb is not None:
            parts.append(emb[:, :, None, None].expand(B, emb.shape[1], H, W))
    return torch.cat(parts, dim=1)


def head(x, sd: State, prefix: str = "model.final"):
    """1x1 conv + per-channel activation, src/model.py:187-193 / :284-292."""
    out = F.conv2d(x, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"])
    if out.shape[1] == 2:
        return torch.cat([torch.tanh(out[:, 0:1]), out[:, 1:2]], dim=1)
    return out


# --------------------------------------------------------------------------- #
# networks
# --------------------------------------------------------------------------- #
def unet_forward(sd: State, maps, temp_series, metadata, training: bool,
                 temporal_embeddings: bool = True, metadata_embeddings: bool = True):
    """UrbanPredictor_unet.forward, src/model.py:261-292."""
    t_emb = temporal_encoder(temp_series, sd) if temporal_embeddings else None     # :263
    m_emb = metadata_encoder(metadata, sd) if metadata_embeddings else None        # :264
    x0_0 = vgg_block(maps, sd, "model.conv0_0", training)                          # :267
    x1_0 = vgg_block(maxpool2x2(x0_0), sd, "model.conv1_0", training)              # :268
    x2_0 = vgg_block(maxpool2x2(x1_0), sd, "model.conv2_0", training)              # :269
    x3_0 = vgg_block(maxpool2x2(x2_0), sd, "model.conv3_0", training)              # :270
    x4_0 = maxpool2x2(x3_0)                                                        # :273
    if temporal_embeddings or metadata_embeddings:
        x4_0 = fuse_embeddings(x4_0, t_emb, m_emb)                                 # :274-275
    x4_0 = vgg_block(x4_0, sd, "model.conv4_0", training)                          # :276
    x3_1 = vgg_block(torch.cat([x3_0, unet_upsample_match(x4_0, x3_0)], 1), sd, "model.conv3_1", training)
    x2_1 = vgg_block(torch.cat([x2_0, unet_upsample_match(x3_1, x2_0)], 1), sd, "model.conv2_1", training)
    x1_1 = vgg_block(torch.cat([x1_0, unet_upsample_match(x2_1, x1_0)], 1), sd, "model.conv1_1", training)
    x0_1 = vgg_block(torch.cat([x0_0, unet_upsample_match(x1_1, x0_0)], 1), sd, "model.conv0_1", training)
    return head(x0_1, sd)                                                          # :284-292


def unetpp_forward(sd: State, maps, temp_series, metadata, training: bool, deep_supervision: bool = False):
    """UrbanPredictor_unetpp.forward, src/model.py:123-193; ``deep_supervision=True`` returns the four bare 1x1
    convolutions ``final1..final4`` of x0_1..x0_4 (:180-185, no tanh).

    Both encoders are always used (:125-126); every decoder node takes
    ``cat([skips..., upsample_to(H,W)(below), emb_map])`` (:136-177) with
    ``emb = cat([temporal_emb, meta_emb])`` (:103).
    """
    emb = torch.cat([temporal_encoder(temp_series, sd), metadata_encoder(metadata, sd)], dim=1)
    X: Dict[tuple, torch.Tensor] = {}

    def node(i, j):
        skips = [X[(i, k)] for k in range(j)]
        H, W = skips[0].shape[2:]
        up = resize_bilinear(X[(i + 1, j - 1)], (H, W))                            # :111-121
        emb_map = emb[:, :, None, None].expand(emb.shape[0], emb.shape[1], H, W)   # :98-108
        X[(i, j)] = vgg_block(torch.cat(skips + [up, emb_map], 1), sd, f"model.conv{i}_{j}", training)

    # execution order of the reference (:129-177)
    X[(0, 0)] = vgg_block(maps, sd, "model.conv0_0", training)
    X[(1, 0)] = vgg_block(maxpool2x2(X[(0, 0)]), sd, "model.conv1_0", training)
    node(0, 1)
    X[(2, 0)] = vgg_block(maxpool2x2(X[(1, 0)]), sd, "model.conv2_0", training)
    node(1, 1)
    node(0, 2)
    X[(3, 0)] = vgg_block(maxpool2x2(X[(2, 0)]), sd, "model.conv3_0", training)
    node(2, 1)
    node(1, 2)
    node(0, 3)
    X[(4, 0)] = vgg_block(maxpool2x2(X[(3, 0)]), sd, "model.conv4_0", training)
    node(3, 1)
    node(2, 2)
    node(1, 3)
    node(0, 4)
    if deep_supervision:                                                           # :180-185
        return [F.conv2d(X[(0, j)], sd[f"model.final{j}.weight"], sd[f"model.final{j}.bias"]) for j in (1, 2, 3, 4)]
    return head(X[(0, 4)], sd)                                                     # :187-193


def forward(model_type: str, sd: State, maps, temp_series, metadata, training: bool,
            temporal_embeddings: bool = True, metadata_embeddings: bool = True):
    """UrbanPredictor.forward dispatch, src/model.py:298-329."""
    if model_type == "unet":
        return unet_forward(sd, maps, temp_series, metadata, training, temporal_embeddings, metadata_embeddings)
    if model_type == "unet++":
        return unetpp_forward(sd, maps, temp_series, metadata, training)
    raise ValueError(f"Unsupported model_type: {model_type}")


# --------------------------------------------------------------------------- #
# loss / train step
# --------------------------------------------------------------------------- #
def loss_mse(outputs, targets):
    """compute_loss_mse, src/utils/losses.py:27-39."""
    mse = F.mse_loss(outputs, targets)
    return {"total": mse, "mse": mse}


def gradient_loss(pred, target):
    """gradient_loss, src/utils/losses.py:5-25: mean | |dy pred| - |dy target| | + the same along x."""
    dy_p = torch.abs(pred[:, :, 1:, :] - pred[:, :, :-1, :])
    dx_p = torch.abs(pred[:, :, :, 1:] - pred[:, :, :, :-1])
    dy_t = torch.abs(target[:, :, 1:, :] - target[:, :, :-1, :])
    dx_t = torch.abs(target[:, :, :, 1:] - target[:, :, :, :-1])
    return {"gradient": torch.mean(torch.abs(dy_p - dy_t)) + torch.mean(torch.abs(dx_p - dx_t))}


def loss_mse_gradient(outputs, targets, lambda_grad: float = 0.1):
    """compute_loss_mse_gradient, src/utils/losses.py:41-57."""
    mse = loss_mse(outputs, targets)["mse"]
    g = gradient_loss(outputs, targets)["gradient"]
    return {"total": mse + lambda_grad * g, "mse": mse, "gradient": g}


def loss_l1_gradient(outputs, targets, lambda_grad: float = 0.1):
    """Differentiable part of compute_loss_l1_grad_ssim, src/utils/losses.py:59-99: the SSIM term is wrapped in
    ``torch.Tensor(...)`` there (:96), i.e. detached -- it shifts the reported value, never the gradient."""
    l1 = F.l1_loss(outputs, targets)
    g = gradient_loss(outputs, targets)["gradient"]
    return {"pixel": l1, "gradient": g, "l1_gradient": l1 + lambda_grad * g}


def train_step(model_type: str, sd: State, optimizer, maps, temp_series, metadata, targets, **flags):
    """Inner step of src/train.py:243-256 (MSE criterion, no clipping): fwd, loss, bwd, step, zero_grad.

    ``sd`` holds leaf tensors with requires_grad (see clone_state); ``optimizer``
    was built over ``[sd[k] for k in sd if is_param(k)]`` in this order.
    Returns (loss, outputs, grads-by-key).
    """
    out = forward(model_type, sd, maps, temp_series, metadata, True, **flags)
    loss = loss_mse(out, targets)["total"]
    loss.backward()
    grads = {k: (v.grad.detach().clone() if v.grad is not None else None)
             for k, v in sd.items() if is_param(k)}
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach(), out.detach(), grads


# --------------------------------------------------------------------------- #
# data-parallel BatchNorm equivalence (new behaviour of the build)
# --------------------------------------------------------------------------- #
def bn_train_pooled(ys: Sequence[torch.Tensor], gamma, beta, eps: float = BN_EPS) -> List[torch.Tensor]:
    """Train-mode BN over the union of per-rank batches ``ys`` (each NCHW).

    Equals ``F.batch_norm(cat(ys), training=True)`` split back per rank; states the
    sum / sum-of-squares exchange the product performs over RCCL.
    """
    n = sum(y.numel() // y.shape[1] for y in ys)
    s = sum(y.double().sum(dim=(0, 2, 3)) for y in ys)
    q = sum((y.double() ** 2).sum(dim=(0, 2, 3)) for y in ys)
    mean = s / n
    var = (q / n - mean * mean).clamp_min(0)
    scale = (gamma.double() / torch.sqrt(var + eps))
    shift = beta.double() - mean * scale
    return [(y.double() * scale[None, :, None, None] + shift[None, :, None, None]).float() for y in ys]


def synthetic_batch(batch: int, spatial_channels: int = 6, size: int = 256, meta_features: int = 4,
                    seq_len: int = 10, out_channels: int = 2, seed: int = 1234):
    """SURVEY 8(d) synthetic inputs: x, ts, md, tgt ~ N(0,1) drawn in this order from one Generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, spatial_channels, size, size, generator=g)
    ts = torch.randn(batch, seq_len, generator=g)
    md = torch.randn(batch, meta_features, generator=g)
    tgt = torch.randn(batch, out_channels, size, size, generator=g)
    return x, ts, md, tgt
