"""Pin the CPU oracle (oracle/unet_ref.py) against fixtures generated from the reference itself.

The fixtures were produced by tests/golden/make_golden.py importing /root/reference/src/model.py.
Where the oracle uses the same torch CPU operators as the reference the match is bit-exact;
the hand-rolled LSTM loop is allowed 1e-6.
"""
import json
import os

import pytest
import torch

from oracle import unet_ref as R
from tests.helpers import GOLDEN, load_npz, meta_of, rel_err, sub, t

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_vgg_block(tag):
    d = load_npz(f"g1_vgg_{tag}.npz")
    sd = {f"blk.{k}": v for k, v in sub(d, "sd0").items()}
    for k in list(sd):
        if R.is_param(k):
            sd[k].requires_grad_(True)
    x = t(d["x"]).requires_grad_(True)
    y = R.vgg_block(x, sd, "blk", True)
    assert torch.equal(y, t(d["y_train"]))
    y.backward(t(d["dy"]))
    assert torch.equal(x.grad, t(d["dx"]))
    for k, g in sub(d, "grad").items():
        assert torch.equal(sd[f"blk.{k}"].grad, g), k
    for k, v in sub(d, "sd1").items():          # running stats / num_batches_tracked after one step
        assert torch.equal(sd[f"blk.{k}"].detach(), v), k
    with torch.no_grad():
        ye = R.vgg_block(x, sd, "blk", False)
    assert torch.equal(ye, t(d["y_eval"]))


def test_g2_spatial():
    d = load_npz("g2_spatial.npz")
    for tag in ("even", "odd", "rect"):
        x = t(d[f"pool_{tag}/x"]).requires_grad_(True)
        y = R.maxpool2x2(x)
        assert torch.equal(y, t(d[f"pool_{tag}/y"]))
        y.backward(t(d[f"pool_{tag}/dy"]))
        assert torch.equal(x.grad, t(d[f"pool_{tag}/dx"]))
    for tag in ("x2", "odd", "rect", "one"):
        x = t(d[f"up_{tag}/x"]).requires_grad_(True)
        ref_y = t(d[f"up_{tag}/y"])
        y = R.unet_upsample_match(x, ref_y)
        assert torch.equal(y, ref_y)
        y.backward(t(d[f"up_{tag}/dy"]))
        assert torch.equal(x.grad, t(d[f"up_{tag}/dx"]))
    for tag in ("x2", "odd", "rect"):
        x = t(d[f"resize_{tag}/x"]).requires_grad_(True)
        ref_y = t(d[f"resize_{tag}/y"])
        y = R.resize_bilinear(x, ref_y.shape[2:])
        assert torch.equal(y, ref_y)
        y.backward(t(d[f"resize_{tag}/dy"]))
        assert torch.equal(x.grad, t(d[f"resize_{tag}/dx"]))


def test_g3_bottleneck_and_encoders():
    d = load_npz("g3_bottleneck.npz")
    sp, te, me = (t(d[k]).requires_grad_(True) for k in ("spatial", "t_emb", "m_emb"))
    fused = R.fuse_embeddings(sp, te, me)
    assert torch.equal(fused, t(d["fused"]))
    w, b = t(d["weight"]).requires_grad_(True), t(d["bias"]).requires_grad_(True)
    y = torch.nn.functional.conv2d(fused, w, b, padding=1)
    assert torch.equal(y, t(d["y"]))
    y.backward(t(d["dy"]))
    assert torch.equal(sp.grad, t(d["d_spatial"]))
    assert torch.equal(te.grad, t(d["d_t_emb"]))
    assert torch.equal(me.grad, t(d["d_m_emb"]))
    e = load_npz("g3_encoders.npz")
    sd = {f"model.{k}": v for k, v in sub(e, "sd").items()}
    assert torch.equal(R.metadata_encoder(t(e["md"]), sd), t(e["meta_emb"]))
    assert rel_err(R.temporal_encoder(t(e["ts"]), sd), t(e["temporal_emb"])) < 1e-6


def test_g4_head():
    d = load_npz("g4_head.npz")
    sd = {"model.final.weight": t(d["weight"]), "model.final.bias": t(d["bias"])}
    x = t(d["x"]).requires_grad_(True)
    y = R.head(x, sd)
    assert torch.equal(y, t(d["y"]))
    y.backward(t(d["dy"]))
    assert torch.equal(x.grad, t(d["dx"]))


FULL = ["g5_unet_even.npz", "g5_unet_odd.npz", "g5_unet_noemb.npz", "g6_unetpp.npz", "g6_unetpp_odd.npz"]


@pytest.mark.parametrize("name", FULL)
def test_g5_g6_full_models(name):
    d = load_npz(name)
    m = meta_of(d)
    kw = m["kw"]
    flags = {k: kw[k] for k in ("temporal_embeddings", "metadata_embeddings") if k in kw}
    sd0 = sub(d, "sd0")
    x, ts, md, tgt = (t(d[k]) for k in ("x", "ts", "md", "tgt"))
    # eval forward
    sd = R.clone_state(sd0)
    with torch.no_grad():
        out_eval = R.forward(kw["model_type"], sd, x, ts, md, False, **flags)
    lstm_on = kw["model_type"] == "unet++" or flags.get("temporal_embeddings", True)
    tol = 2e-6 if lstm_on else 0.0
    assert rel_err(out_eval, t(d["out_eval"])) <= tol
    # one train step
    sd = R.clone_state(sd0, requires_grad=True)
    params = [sd[k] for k in sd if R.is_param(k)]
    opt = torch.optim.AdamW(params, lr=m["lr"], weight_decay=m["weight_decay"])
    loss, out, grads = R.train_step(kw["model_type"], sd, opt, x, ts, md, tgt, **flags)
    assert rel_err(out, t(d["out_train"])) <= tol
    assert abs(float(loss) - float(d["loss"][0])) <= 1e-6 * abs(float(d["loss"][0]))
    ref_grads = sub(d, "grad")
    for k in m["nograd"]:
        assert grads[k] is None, k                     # unused temporal encoder -> grad None
    for k, g in ref_grads.items():
        if tol == 0.0:
            assert torch.equal(grads[k], g), k
        else:
            assert rel_err(grads[k], g) <= 5e-5 or float((grads[k] - g).abs().max()) < 1e-7, k
    for k, v in sub(d, "sd1").items():
        a = sd[k].detach()
        if tol == 0.0 or not a.is_floating_point():
            assert torch.equal(a, v), k
        else:
            assert float((a - v).abs().max()) <= 2.5e-4, k   # Adam: sign-like updates of ~lr on noisy tiny grads


def test_init_state_matches_reference_layout():
    """Key set / shapes / counts quoted in SURVEY 8(b): 138 entries (unet), 222 (unet++)."""
    torch.manual_seed(0)
    sd = R.init_state("unet", 6, 10, 64, 4, 64, 96, 2, temporal_embeddings=False, metadata_embeddings=True)
    assert len(sd) == 138
    assert sum(v.numel() for k, v in sd.items() if R.is_param(k)) == 32028834
    sdpp = R.init_state("unet++", 6, 10, 64, 4, 64, 96, 2)
    assert len(sdpp) == 222
    assert sum(v.numel() for k, v in sdpp.items() if R.is_param(k)) == 38594850
    d = load_npz("g5_unet_even.npz")
    kw = meta_of(d)["kw"]
    small = R.init_state(**{k: v for k, v in kw.items() if k != "deep_supervision"})
    ref = sub(d, "sd0")
    assert set(small) == set(ref)
    for k in ref:
        assert small[k].shape == ref[k].shape and small[k].dtype == ref[k].dtype, k
    with pytest.raises(ValueError):
        R.init_state("resnet", 6, 10, 64, 4, 64, 96, 2)


def test_g7_full_size_known_answer():
    """Full base_filters=64, 256x256, B=2 model reproduces the reference's known-answer statistics."""
    with open(os.path.join(GOLDEN, "g7_full_summary.json")) as f:
        s = json.load(f)
    torch.manual_seed(0)
    sd = R.clone_state(R.init_state("unet", 6, 10, 64, 4, 64, 96, 2, temporal_embeddings=False,
                                    metadata_embeddings=True), requires_grad=True)
    x, ts, md, tgt = R.synthetic_batch(2)
    out = R.unet_forward(sd, x, ts, md, True, temporal_embeddings=False, metadata_embeddings=True)
    loss = R.loss_mse(out, tgt)["total"]
    loss.backward()
    assert abs(float(loss) - s["loss"]) < 1e-6
    assert abs(float(out.mean()) - s["out_mean"]) < 1e-6
    assert abs(float(out.std()) - s["out_std"]) < 1e-6
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for k, v in sd.items() if R.is_param(k) and v.grad is not None))
    assert abs(float(gn) - s["grad_norm"]) < 1e-5
    g7 = load_npz("g7_full_samples.npz")
    assert torch.equal(out.detach().flatten()[t(g7["out_idx"])], t(g7["out_vals"]))
    for k in s["nograd"]:
        assert sd[k].grad is None


def test_g8_syncbn_pooled_equals_single_device():
    d = load_npz("g8_syncbn.npz")
    sd0 = sub(d, "sd0")
    x = t(d["x"])
    y1 = torch.nn.functional.conv2d(x, sd0["conv1.weight"], sd0["conv1.bias"], padding=1)
    for split in (2, 4):
        parts = list(y1.chunk(split, dim=0))
        pooled = torch.cat(R.bn_train_pooled(parts, sd0["bn1.weight"], sd0["bn1.bias"]), dim=0)
        ref = torch.nn.functional.batch_norm(y1, None, None, sd0["bn1.weight"], sd0["bn1.bias"], True, 0.1, 1e-5)
        assert rel_err(pooled, ref) < 1e-6


def test_g9_losses_oracle():
    """gradient_loss / compute_loss_mse_gradient / the differentiable part of compute_loss_l1_grad_ssim."""
    d = load_npz("g9_losses.npz")
    for tag in ("a", "b", "c"):
        o = t(d[f"{tag}/out"]).requires_grad_(True)
        tg = t(d[f"{tag}/tgt"])
        r = R.loss_mse_gradient(o, tg)
        r["total"].backward()
        assert torch.equal(r["total"].detach().reshape(1), t(d[f"{tag}/mse_gradient_total"]))
        assert torch.equal(r["gradient"].detach().reshape(1), t(d[f"{tag}/gradient"]))
        assert torch.equal(o.grad, t(d[f"{tag}/d_mse_gradient"]))
        o.grad = None
        r2 = R.loss_l1_gradient(o, tg)
        r2["l1_gradient"].backward()
        assert torch.equal(r2["pixel"].detach().reshape(1), t(d[f"{tag}/l1"]))
        assert torch.equal(o.grad, t(d[f"{tag}/d_l1_gradient"]))


def test_g10_unetpp_deep_supervision():
    """deep_supervision=True (src/model.py:180-185): four bare 1x1 heads, no tanh; forward (eval + train) and the
    gradients of the summed MSE against the fixture generated from the reference."""
    import torch.nn.functional as F
    d = load_npz("g10_unetpp_deepsup.npz")
    m = meta_of(d)
    x, ts, md, tgt = (t(d[k]) for k in ("x", "ts", "md", "tgt"))
    sd = R.clone_state(sub(d, "sd0"))
    with torch.no_grad():
        outs = R.unetpp_forward(sd, x, ts, md, False, deep_supervision=True)
    assert len(outs) == 4
    for j, o in enumerate(outs):
        assert rel_err(o, t(d[f"out_eval{j}"])) <= 2e-6
    sd = R.clone_state(sub(d, "sd0"), requires_grad=True)
    outs = R.unetpp_forward(sd, x, ts, md, True, deep_supervision=True)
    for j, o in enumerate(outs):
        assert rel_err(o, t(d[f"out_train{j}"])) <= 2e-6
    loss = sum(F.mse_loss(o, tgt) for o in outs)
    assert abs(float(loss) - float(d["loss"][0])) <= 1e-6 * abs(float(d["loss"][0]))
    loss.backward()
    for k, g in sub(d, "grad").items():
        assert rel_err(sd[k].grad, g) <= 5e-5 or float((sd[k].grad - g).abs().max()) < 1e-7, k
    for k in m["nograd"]:
        assert sd[k].grad is None, k                       # model.final.* is unused under deep supervision


def test_g11_temporal_encoder_828():
    """TemporalEncoder at the reference's real length (828 monthly temperatures): the oracle's explicit LSTM loop against
    the reference's nn.LSTM on the fixture -- embedding and every parameter gradient (an 828-step fp32 recurrence: 1e-5)."""
    d = load_npz("g11_temporal_828.npz")
    sd = {f"model.temporal_encoder.{k}": t(v).requires_grad_(True) for k, v in sub(d, "sd").items()}
    emb = R.temporal_encoder(t(d["ts"]), sd)
    assert rel_err(emb, t(d["emb"])) < 1e-5
    emb.backward(t(d["demb"]))
    for k, g in sub(d, "grad").items():
        assert rel_err(sd[f"model.temporal_encoder.{k}"].grad, g) < 1e-4, k
