#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (``/root/reference`` is not on the GPU box):

    python tests/golden/make_golden.py

Imports ``/root/reference/src/model.py`` (with a no-op ``loguru`` stub: loguru is
not installed and is used for one log line, src/model.py:202), runs the
reference modules on seeded inputs on the CPU and writes inputs + expected
outputs as ``.npz`` (fp32).  No reference source is copied: fixtures are data.
The fixture ids follow SURVEY.md 8(c): G1..G8 (+ G9 losses, G10 deep supervision, G11 TemporalEncoder at T=828).
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MAU_REFERENCE", "/root/reference")


def import_reference():
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules.setdefault("loguru", lg)
    sys.path.insert(0, REF)
    import src.model as ref_model  # noqa: E402
    return ref_model


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


def sd_arrays(prefix, sd):
    return {f"{prefix}/{k}": v for k, v in sd.items()}


def randomize_bn(mod, gen):
    """Make BN affine/running buffers non-trivial so eval mode is a real test."""
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) * 1.5 - 0.25)   # some negative gammas
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.3)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


def g1_vgg(ref):
    for tag, (B, cin, cmid, cout, H, W) in {"a": (2, 6, 16, 16, 32, 32), "b": (2, 23, 8, 8, 31, 31),
                                              "c": (3, 12, 40, 24, 17, 21)}.items():
        torch.manual_seed(100 + ord(tag))
        blk = ref.VGGBlock(cin, cmid, cout)
        g = torch.Generator().manual_seed(7)
        randomize_bn(blk, g)
        sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
        x = torch.randn(B, cin, H, W, generator=g, requires_grad=True)
        dy = torch.randn(B, cout, H, W, generator=g)
        blk.train()
        y = blk(x)
        y.backward(dy)
        grads = {f"grad/{k}": p.grad for k, p in blk.named_parameters()}
        sd1 = {k: v.clone() for k, v in blk.state_dict().items()}
        blk.eval()
        with torch.no_grad():
            y_eval = blk(x)
        npz(f"g1_vgg_{tag}.npz", x=x, dy=dy, y_train=y, dx=x.grad, y_eval=y_eval,
            **sd_arrays("sd0", sd0), **sd_arrays("sd1", sd1), **grads)


def g2_spatial(ref):
    g = torch.Generator().manual_seed(11)
    out = {}
    pool = torch.nn.MaxPool2d(2, 2)
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    for tag, (B, C, H, W) in {"even": (2, 5, 16, 16), "odd": (2, 7, 31, 31), "rect": (1, 3, 13, 10)}.items():
        x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
        y = pool(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        out.update({f"pool_{tag}/x": x, f"pool_{tag}/y": y, f"pool_{tag}/dy": dy, f"pool_{tag}/dx": x.grad})
    # U-Net decoder up-path: up x2 then (only if needed) resize to the skip's size, src/model.py:243-246,279-282
    for tag, (B, C, h, w, H, W) in {"x2": (2, 6, 16, 16, 32, 32), "odd": (2, 5, 15, 15, 31, 31),
                                    "rect": (1, 4, 6, 5, 13, 10), "one": (1, 3, 1, 1, 3, 3)}.items():
        x = torch.randn(B, C, h, w, generator=g, requires_grad=True)
        u = up(x)
        if u.shape[2:] != (H, W):
            u = F.interpolate(u, size=(H, W), mode="bilinear", align_corners=True)
        du = torch.randn(u.shape, generator=g)
        u.backward(du)
        out.update({f"up_{tag}/x": x, f"up_{tag}/y": u, f"up_{tag}/dy": du, f"up_{tag}/dx": x.grad})
    # U-Net++ up-path: straight resize to the target size, src/model.py:111-121
    for tag, (B, C, h, w, H, W) in {"x2": (1, 4, 8, 8, 16, 16), "odd": (2, 3, 15, 15, 31, 31), "rect": (1, 2, 7, 5, 15, 11)}.items():
        x = torch.randn(B, C, h, w, generator=g, requires_grad=True)
        u = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)
        du = torch.randn(u.shape, generator=g)
        u.backward(du)
        out.update({f"resize_{tag}/x": x, f"resize_{tag}/y": u, f"resize_{tag}/dy": du, f"resize_{tag}/dx": x.grad})
    npz("g2_spatial.npz", **out)


def g3_bottleneck(ref):
    """fuse_embeddings + the first conv of conv4_0 (border-aware broadcast), via a tiny reference U-Net."""
    torch.manual_seed(3)
    net = quiet(ref.UrbanPredictor_unet, 6, 10, 16, 4, 8, 12, 2, nb_filter=[4, 8, 16, 32, 64],
                temporal_embeddings=True, metadata_embeddings=True)
    g = torch.Generator().manual_seed(5)
    spatial = torch.randn(2, 32, 5, 4, generator=g, requires_grad=True)
    t_emb = torch.randn(2, 16, generator=g, requires_grad=True)
    m_emb = torch.randn(2, 8, generator=g, requires_grad=True)
    fused = net.fuse_embeddings(spatial, t_emb, m_emb)
    y = net.conv4_0.conv1(fused)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    npz("g3_bottleneck.npz", spatial=spatial, t_emb=t_emb, m_emb=m_emb, fused=fused, y=y, dy=dy,
        d_spatial=spatial.grad, d_t_emb=t_emb.grad, d_m_emb=m_emb.grad,
        weight=net.conv4_0.conv1.weight, bias=net.conv4_0.conv1.bias,
        d_weight=net.conv4_0.conv1.weight.grad, d_bias=net.conv4_0.conv1.bias.grad)
    # metadata / temporal encoders on their own (src/model.py:23-48)
    md = torch.randn(3, 4, generator=g, requires_grad=True)
    ts = torch.randn(3, 12, generator=g)
    me = net.meta_encoder(md)
    te = net.temporal_encoder(ts)
    dme = torch.randn(me.shape, generator=g)
    me.backward(dme)
    npz("g3_encoders.npz", md=md, ts=ts, meta_emb=me, temporal_emb=te, d_meta_emb=dme, d_md=md.grad,
        **sd_arrays("sd", net.state_dict()),
        **{f"grad/{k}": p.grad for k, p in net.meta_encoder.named_parameters()})


def g4_head(ref):
    torch.manual_seed(4)
    net = quiet(ref.UrbanPredictor_unet, 6, 10, 16, 4, 8, 12, 2, nb_filter=[12, 8, 16, 32, 64])
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 12, 9, 11, generator=g, requires_grad=True)
    out = net.final(x)
    y = torch.cat([torch.tanh(out[:, 0:1]), out[:, 1:2]], dim=1)                   # src/model.py:287-290
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    npz("g4_head.npz", x=x, y=y, dy=dy, dx=x.grad, weight=net.final.weight, bias=net.final.bias,
        d_weight=net.final.weight.grad, d_bias=net.final.bias.grad)


def full_model_case(ref, name, model_type, B, C, F_, H, W, T, base, flags, seed, steps=1):
    torch.manual_seed(seed)
    kw = dict(model_type=model_type, spatial_channels=C, seq_len=T, temporal_dim=8, meta_features=F_,
              meta_dim=8, lstm_dim=12, out_channels=2, base_filters=base, deep_supervision=False, **flags)
    net = quiet(ref.UrbanPredictor, **kw)
    g = torch.Generator().manual_seed(seed + 1000)
    randomize_bn(net, g)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(B, C, H, W, generator=g)
    ts = torch.randn(B, T, generator=g)
    md = torch.randn(B, F_, generator=g)
    tgt = torch.randn(B, 2, H, W, generator=g)
    net.eval()
    with torch.no_grad():
        out_eval = net(x, ts, md)
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-3)           # conf/config.yaml:41,48,52
    out = net(x, ts, md)
    loss = F.mse_loss(out, tgt)                                                    # src/utils/losses.py:33
    loss.backward()
    grads = {f"grad/{k}": p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    nograd = [k for k, p in net.named_parameters() if p.grad is None]
    opt.step()
    opt.zero_grad()
    sd1 = {k: v.clone() for k, v in net.state_dict().items()}
    meta = dict(kw=kw, B=B, H=H, W=W, nograd=nograd, lr=1e-4, weight_decay=1e-3)
    npz(name, x=x, ts=ts, md=md, tgt=tgt, out_train=out, out_eval=out_eval, loss=loss.detach().reshape(1),
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        **sd_arrays("sd0", sd0), **sd_arrays("sd1", sd1), **grads)


def g7_full_summary(ref):
    """Full-size base_filters=64 model: too big to commit; keep summary statistics + sampled elements."""
    torch.manual_seed(0)
    net = quiet(ref.UrbanPredictor, 'unet', 6, 10, 64, 4, 64, 96, 2, temporal_embeddings=False, metadata_embeddings=True)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(2, 6, 256, 256, generator=g)
    ts = torch.randn(2, 10, generator=g)
    md = torch.randn(2, 4, generator=g)
    tgt = torch.randn(2, 2, 256, 256, generator=g)
    net.train()
    out = net(x, ts, md)
    loss = F.mse_loss(out, tgt)
    loss.backward()
    pg = [(k, p.grad) for k, p in net.named_parameters() if p.grad is not None]
    gnorm = torch.sqrt(sum((v.double() ** 2).sum() for _, v in pg)).item()
    idx = torch.randint(0, out.numel(), (512,), generator=torch.Generator().manual_seed(99))
    per_param_norm = {k: float(v.double().norm()) for k, v in pg}
    summary = dict(loss=float(loss), out_mean=float(out.mean()), out_std=float(out.std()),
                   out_absmax=float(out.abs().max()), grad_norm=gnorm, n_params=sum(p.numel() for p in net.parameters()),
                   n_state=len(net.state_dict()), per_param_grad_norm=per_param_norm,
                   nograd=[k for k, p in net.named_parameters() if p.grad is None])
    with open(os.path.join(HERE, "g7_full_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    sample_grads = {}
    for k, v in pg:
        if k.endswith("conv1.weight") or k.endswith("bn2.weight") or "final" in k or "meta_encoder" in k:
            flat = v.flatten()
            sample_grads[f"grad_head/{k}"] = flat[:64].clone()
    npz("g7_full_samples.npz", out_idx=idx.numpy(), out_vals=out.flatten()[idx], **sample_grads)
    print("g7:", {k: summary[k] for k in ("loss", "out_mean", "out_std", "out_absmax", "grad_norm")})


def g8_syncbn(ref):
    """Reference single-device B=4 VGG block, to be matched by 2 x B=2 / 4 x B=1 with pooled statistics."""
    torch.manual_seed(8)
    blk = ref.VGGBlock(5, 8, 8)
    g = torch.Generator().manual_seed(9)
    randomize_bn(blk, g)
    sd0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.randn(4, 5, 12, 12, generator=g, requires_grad=True)
    dy = torch.randn(4, 8, 12, 12, generator=g)
    blk.train()
    y = blk(x)
    y.backward(dy)
    sd1 = {k: v.clone() for k, v in blk.state_dict().items()}
    npz("g8_syncbn.npz", x=x, dy=dy, y=y, dx=x.grad, **sd_arrays("sd0", sd0), **sd_arrays("sd1", sd1),
        **{f"grad/{k}": p.grad for k, p in blk.named_parameters()})


def g9_losses():
    """src/utils/losses.py: piq (SSIM) is not installed; it is stubbed so the module imports, and only the
    functions that never reach it are exercised (gradient_loss :5-25, compute_loss_mse :27-39,
    compute_loss_mse_gradient :41-57; L1 = F.l1_loss :68)."""
    stub = types.ModuleType("piq")
    stub.ssim = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("piq.ssim is not available in this container"))
    sys.modules.setdefault("piq", stub)
    import src.utils.losses as L
    g = torch.Generator().manual_seed(90)
    out = {}
    for tag, shape in {"a": (2, 2, 17, 13), "b": (3, 2, 32, 32), "c": (1, 2, 250, 250)}.items():
        o = torch.randn(shape, generator=g, requires_grad=True)
        t = torch.randn(shape, generator=g)
        d = L.compute_loss_mse_gradient(o, t)
        d["total"].backward()
        out.update({f"{tag}/out": o, f"{tag}/tgt": t, f"{tag}/mse_gradient_total": d["total"].detach().reshape(1),
                    f"{tag}/mse": d["mse"].detach().reshape(1), f"{tag}/gradient": d["gradient"].detach().reshape(1),
                    f"{tag}/d_mse_gradient": o.grad.clone()})
        o.grad = None
        l1 = F.l1_loss(o, t)
        gl = L.gradient_loss(o, t)["gradient"]
        (l1 + 0.1 * gl).backward()                 # the differentiable part of compute_loss_l1_grad_ssim (:68-71,99)
        out.update({f"{tag}/l1": l1.detach().reshape(1), f"{tag}/d_l1_gradient": o.grad.clone()})
    npz("g9_losses.npz", **out)


def g10_deep_supervision(ref):
    """U-Net++ with deep_supervision=True: four bare 1x1 heads on x0_1..x0_4 (src/model.py:180-185), train-mode
    forward/backward of the summed MSE of the four outputs, and the eval-mode outputs."""
    torch.manual_seed(70)
    kw = dict(model_type="unet++", spatial_channels=6, seq_len=8, temporal_dim=8, meta_features=4, meta_dim=8, lstm_dim=12,
              out_channels=2, base_filters=4, deep_supervision=True)
    net = quiet(ref.UrbanPredictor, **kw)
    g = torch.Generator().manual_seed(1070)
    randomize_bn(net, g)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    B, H, W = 2, 32, 40
    x, ts, md = torch.randn(B, 6, H, W, generator=g), torch.randn(B, 8, generator=g), torch.randn(B, 4, generator=g)
    tgt = torch.randn(B, 2, H, W, generator=g)
    net.eval()
    with torch.no_grad():
        outs_eval = net(x, ts, md)
    net.train()
    outs = net(x, ts, md)
    assert isinstance(outs, list) and len(outs) == 4
    loss = sum(F.mse_loss(o, tgt) for o in outs)
    loss.backward()
    grads = {f"grad/{k}": p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    nograd = [k for k, p in net.named_parameters() if p.grad is None]
    meta = dict(kw=kw, B=B, H=H, W=W, nograd=nograd)
    npz("g10_unetpp_deepsup.npz", x=x, ts=ts, md=md, tgt=tgt, loss=loss.detach().reshape(1),
        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        **{f"out_train{j}": o for j, o in enumerate(outs)}, **{f"out_eval{j}": o for j, o in enumerate(outs_eval)},
        **sd_arrays("sd0", sd0), **grads)


def g11_temporal_encoder_828(ref):
    """TemporalEncoder (src/model.py:23-34) at the reference's real sequence length (conf/config.yaml:20 temporal_length 828),
    hidden 96 -> 64 as in conf/config.yaml:46-47: last-hidden embedding and every parameter gradient."""
    torch.manual_seed(111)
    enc = ref.TemporalEncoder(828, hidden_dim=96, out_dim=64)
    g = torch.Generator().manual_seed(112)
    ts = torch.randn(3, 828, generator=g)
    emb = enc(ts)
    demb = torch.randn(emb.shape, generator=g)
    emb.backward(demb)
    npz("g11_temporal_828.npz", ts=ts, emb=emb, demb=demb, **sd_arrays("sd", enc.state_dict()),
        **{f"grad/{k}": p.grad for k, p in enc.named_parameters()})


def main():
    ref = import_reference()
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # regenerate selected fixtures only: make_golden.py g10 ...
        for name in sys.argv[1:]:
            fn = {"g10": lambda: g10_deep_supervision(ref), "g11": lambda: g11_temporal_encoder_828(ref)}[name]
            fn()
        return
    g1_vgg(ref)
    g2_spatial(ref)
    g3_bottleneck(ref)
    g4_head(ref)
    full_model_case(ref, "g5_unet_even.npz", "unet", 2, 6, 4, 64, 64, 10, 4,
                    dict(temporal_embeddings=False, metadata_embeddings=True), seed=50)
    full_model_case(ref, "g5_unet_odd.npz", "unet", 2, 23, 8, 62, 62, 9, 4,
                    dict(temporal_embeddings=True, metadata_embeddings=True), seed=51)
    full_model_case(ref, "g5_unet_noemb.npz", "unet", 1, 6, 4, 32, 48, 6, 4,
                    dict(temporal_embeddings=False, metadata_embeddings=False), seed=52)
    full_model_case(ref, "g6_unetpp.npz", "unet++", 2, 6, 4, 32, 32, 12, 4, {}, seed=60)
    full_model_case(ref, "g6_unetpp_odd.npz", "unet++", 1, 7, 8, 34, 34, 5, 4, {}, seed=61)
    g7_full_summary(ref)
    g8_syncbn(ref)
    g9_losses()
    g10_deep_supervision(ref)
    g11_temporal_encoder_828(ref)


if __name__ == "__main__":
    main()
