"""GPU parity tests, whole-network level: UrbanPredictor (unet / unet++) forward, loss, backward,
BatchNorm buffer updates and one AdamW step against the fixtures generated from the reference."""
import pytest
import torch

from oracle import unet_ref as R
from tests.helpers import load_npz, meta_of, rel_err, rel_l2, sub, t

pytestmark = pytest.mark.gpu

FULL = ["g5_unet_even.npz", "g5_unet_odd.npz", "g5_unet_noemb.npz", "g6_unetpp.npz", "g6_unetpp_odd.npz"]


@pytest.fixture(scope="module")
def mau():
    import mau_amd
    assert torch.cuda.is_available()
    return mau_amd


def build(mau, d, prec):
    m = meta_of(d)
    net = mau.UrbanPredictor(**m["kw"])
    net.load_state_dict(sub(d, "sd0"), strict=True)          # reference checkpoint layout, strict
    return net.cuda().set_precision(prec), m


@pytest.mark.parametrize("name", FULL)
def test_fp32_parity_full_model(mau, name):
    """fp32 parity mode: <= 1e-3 relative on outputs, loss, every gradient, BN buffers, AdamW step."""
    d = load_npz(name)
    net, m = build(mau, d, "fp32")
    x, ts, md, tgt = (t(d[k]).cuda() for k in ("x", "ts", "md", "tgt"))
    net.eval()
    with torch.no_grad():
        out_eval = net(x, ts, md)
    assert rel_err(out_eval.cpu(), t(d["out_eval"])) < 1e-3
    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=m["lr"], weight_decay=m["weight_decay"])
    out = net(x, ts, md)
    assert out.shape == tgt.shape and out.dtype == torch.float32
    assert rel_err(out.cpu(), t(d["out_train"])) < 1e-3
    loss = mau.compute_loss_mse(out, tgt)["total"]
    assert abs(float(loss) - float(d["loss"][0])) < 1e-4 * abs(float(d["loss"][0]))
    loss.backward()
    params = dict(net.named_parameters())
    for k in m["nograd"]:
        assert params[k].grad is None, k                      # unused temporal encoder
    worst = 0.0
    for k, gref in sub(d, "grad").items():
        got = params[k].grad.cpu()
        if k.endswith(".conv1.bias") or k.endswith(".conv2.bias"):
            assert float(got.abs().max()) == 0.0 and float(gref.abs().max()) < 5e-3, k
            continue
        e = rel_err(got, gref)
        worst = max(worst, e)
        assert e < 1e-3 or float((got - gref).abs().max()) < 1e-7, (k, e)
    opt.step()
    sd1 = sub(d, "sd1")
    sd = net.state_dict()
    for k, v in sd1.items():
        a = sd[k].cpu()
        if not a.is_floating_point():
            assert torch.equal(a, v), k
        elif "running_" in k:
            assert rel_err(a, v) < 1e-3, k
        elif k.endswith((".conv1.bias", ".conv2.bias")):
            # conv biases in front of BN see pure rounding-noise gradients in the reference (its first AdamW step moves them by
            # up to lr in a random direction), exact zeros here (only the weight decay moves them): |difference| <= lr (+ margin)
            assert float((a - v).abs().max()) <= 2.5 * m["lr"], k
        else:
            # every other parameter: the first AdamW step is ~ -lr * sign(g), so an update in the WRONG direction is 2 lr away.
            # Bound: 99 % of a tensor's elements within 0.2 lr of the reference's (measured: >= 99.7 %; the rest are elements whose
            # gradient is rounding noise around zero -- g / (|g| + eps) is discontinuous there and the sign may flip), none further
            # than one flipped sign, and the update vector as a whole within 25 % (relative L2) of the reference's: a wrong-way
            # update (relative L2 = 2, no element close) or a stale / doubled one fails all three
            diff = (a - v).abs()
            close = float((diff <= 0.2 * m["lr"]).float().mean())
            assert close >= 0.99 or diff.numel() - int((diff <= 0.2 * m["lr"]).sum()) <= 1, (k, close)
            assert float(diff.max()) <= 2.05 * m["lr"], k
            upd_ref = v - sub(d, "sd0")[k]
            if float(upd_ref.norm()) > 0:
                assert rel_l2(a - sub(d, "sd0")[k], upd_ref) <= 0.25, (k, rel_l2(a - sub(d, "sd0")[k], upd_ref))


def _autocast_yardstick(d, m):
    """Error of the REFERENCE algorithm itself under torch's CPU bf16 autocast vs its fp32 result:
    the inherent bf16 noise of this network/fixture (ReLU-mask flips, tiny-batch BatchNorm)."""
    kw = m["kw"]
    flags = {k: kw[k] for k in ("temporal_embeddings", "metadata_embeddings") if k in kw}
    x, ts, md, tgt = (t(d[k]) for k in ("x", "ts", "md", "tgt"))
    sd = R.clone_state(sub(d, "sd0"), requires_grad=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = R.forward(kw["model_type"], sd, x, ts, md, True, **flags)
    R.loss_mse(out.float(), tgt)["total"].backward()
    gref = sub(d, "grad")
    num = sum(float(((sd[k].grad.double() - g.double()) ** 2).sum()) for k, g in gref.items())
    den = sum(float((g.double() ** 2).sum()) for g in gref.values())
    return rel_l2(out.detach().float(), t(d["out_train"])), (num / den) ** 0.5


@pytest.mark.parametrize("name", FULL)
def test_bf16_mode_full_model(mau, name):
    """bf16 throughput mode.  Bound: no worse than 1.5x the error the reference's own operators show
    under torch's bf16 autocast on the same fixture (+2e-2), see SURVEY D8."""
    d = load_npz(name)
    net, m = build(mau, d, "bf16")
    out_yard, grad_yard = _autocast_yardstick(d, m)
    x, ts, md, tgt = (t(d[k]).cuda() for k in ("x", "ts", "md", "tgt"))
    net.train()
    out = net(x, ts, md)
    e_out = rel_l2(out.detach().cpu(), t(d["out_train"]))
    loss = mau.compute_loss_mse(out, tgt)["total"]
    loss.backward()
    params = dict(net.named_parameters())
    gref = sub(d, "grad")
    num = sum(float(((params[k].grad.cpu().double() - g.double()) ** 2).sum()) for k, g in gref.items())
    den = sum(float((g.double() ** 2).sum()) for g in gref.values())
    e_grad = (num / den) ** 0.5
    print(f"{name}: bf16 out relL2 {e_out:.4f} (autocast yardstick {out_yard:.4f}); grad relL2 {e_grad:.4f} (yardstick {grad_yard:.4f})")
    assert e_out <= 1.5 * out_yard + 2e-2
    assert abs(float(loss) - float(d["loss"][0])) < 0.1 * abs(float(d["loss"][0]))
    assert e_grad <= 1.5 * grad_yard + 2e-2


def test_full_size_known_answer_fp32(mau):
    """BASELINE config 1 (base_filters=64, 256x256, B=2): the reference's known-answer statistics
    (tests/golden/g7_full_summary.json) from the HIP fp32 path."""
    import json
    import os
    from tests.helpers import GOLDEN
    with open(os.path.join(GOLDEN, "g7_full_summary.json")) as f:
        s = json.load(f)
    torch.manual_seed(0)
    net = mau.UrbanPredictor('unet', 6, 10, 64, 4, 64, 96, 2, temporal_embeddings=False, metadata_embeddings=True)
    net = net.cuda().set_precision("fp32").train()
    x, ts, md, tgt = (v.cuda() for v in R.synthetic_batch(2))
    out = net(x, ts, md)
    loss = mau.compute_loss_mse(out, tgt)["total"]
    loss.backward()
    assert abs(float(loss) - s["loss"]) < 1e-4 * s["loss"]
    assert abs(float(out.mean()) - s["out_mean"]) < 1e-4
    assert abs(float(out.std()) - s["out_std"]) < 1e-4
    g7 = load_npz("g7_full_samples.npz")
    assert rel_err(out.detach().flatten()[t(g7["out_idx"]).cuda()].cpu(), t(g7["out_vals"])) < 1e-3
    params = dict(net.named_parameters())
    gn = sum(float((p.grad.double() ** 2).sum()) for p in params.values() if p.grad is not None) ** 0.5
    assert abs(gn - s["grad_norm"]) < 1e-3 * s["grad_norm"]
    # Per-parameter gradient norms against the committed known answers, bounded by a yardstick MEASURED HERE (as
    # test_production_shape_fp32_vs_oracle does): the oracle's own fp32 gradients against an fp64 evaluation of the same graph
    # on the same inputs.  |norm(a) - norm(b)| <= norm(a - b), so a parameter's norm may differ from the reference's by the
    # length of the reference's own rounding-error vector -- times 2: the HIP path must be as close to the reference as the
    # reference is to exact arithmetic (and never needs more than 1e-3 of the norm itself when that is larger).
    torch.manual_seed(0)
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    sd0 = R.init_state("unet", 6, 10, 64, 4, 64, 96, 2, **flags)
    xc, tsc, mdc, tgtc = R.synthetic_batch(2)
    sd32 = R.clone_state(sd0, requires_grad=True)
    R.loss_mse(R.forward("unet", sd32, xc, tsc, mdc, True, **flags), tgtc)["total"].backward()
    sd64 = {k: (v.detach().double().requires_grad_(True) if R.is_param(k) else v.detach().clone().double() if v.is_floating_point() else v.clone())
            for k, v in sd0.items()}
    R.loss_mse(R.forward("unet", sd64, xc.double(), tsc.double(), mdc.double(), True, **flags), tgtc.double())["total"].backward()
    worst = (0.0, "", 0.0, 0.0)
    for k, ref_norm in s["per_param_grad_norm"].items():
        if k.endswith(".conv1.bias") or k.endswith(".conv2.bias"):
            continue
        got = float(params[k].grad.double().norm())
        yard = float((sd32[k].grad.double() - sd64[k].grad).norm())                                  # reference fp32 vs fp64, this parameter
        # (the oracle run on THIS host against the committed answer of the reference on another: the same bound applies)
        assert abs(float(sd32[k].grad.double().norm()) - ref_norm) <= max(1e-3 * ref_norm, 2.0 * yard), k
        dev = abs(got - ref_norm)
        if dev / max(ref_norm, 1e-30) > worst[0]:
            worst = (dev / max(ref_norm, 1e-30), k, dev, yard)
        assert dev <= max(1e-3 * ref_norm, 2.0 * yard), (k, got, ref_norm, yard)
        # and element-wise: relative L2 against the oracle's gradient within 2x the oracle's own fp32-vs-fp64 relative L2 (+1e-3)
        e2 = rel_l2(params[k].grad.cpu(), sd32[k].grad)
        y2 = rel_l2(sd32[k].grad, sd64[k].grad.float())
        assert e2 <= 2.0 * y2 + 1e-3, (k, e2, y2)
    print(f"known answer: worst per-parameter norm deviation {worst[0]:.2e} of the norm on {worst[1]} (|dev| {worst[2]:.3e}, reference "
          f"fp32-vs-fp64 error vector there {worst[3]:.3e})")


@pytest.mark.parametrize("model_type,B", [("unet", 2), ("unet++", 1)])
def test_production_shape_fp32_vs_oracle(mau, model_type, B):
    """The reference's real training shape (conf/config.yaml:15,18,19): 23 channels, 250x250 tiles
    (250->125->62->31->15: every decoder level takes the odd-size second resize), 8 metadata features,
    temporal + metadata embeddings, base_filters 64.  HIP fp32 path vs the CPU oracle, same seed.

    Tolerances: outputs and loss <= 1e-3 max-norm (north star).  Gradients at this depth/batch are judged against a
    yardstick COMPUTED HERE: the deviation of the reference's own fp32 gradients from an fp64 evaluation of the same
    graph on the same inputs (ReLU masks flip where a pre-activation differs in the last bits and small-batch BatchNorm
    amplifies it).  Bound: 2x that deviation (+1e-3) -- relative L2 per parameter, max-norm against the worst parameter's
    deviation -- i.e. the HIP fp32 path must be as close to the reference as the reference is to exact arithmetic.  The element-wise
    ``allclose(rtol=1e-3, atol=1e-3*max|ref|)`` pass fraction is printed next to the max-norm figure."""
    flags = {} if model_type == "unet++" else dict(temporal_embeddings=True, metadata_embeddings=True)
    torch.manual_seed(11)
    net = mau.UrbanPredictor(model_type, 23, 12, 64, 8, 64, 96, 2, base_filters=64, **flags)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, 23, 250, 250, generator=g)
    ts = torch.randn(B, 12, generator=g)
    md = torch.randn(B, 8, generator=g)
    tgt = torch.randn(B, 2, 250, 250, generator=g)
    sd = R.clone_state(sd0, requires_grad=True)
    ref = R.forward(model_type, sd, x, ts, md, True, **flags)
    ref_loss = R.loss_mse(ref, tgt)["total"]
    ref_loss.backward()
    # yardstick: the same graph in fp64 (same oracle code, double tensors)
    sd64 = {k: (v.detach().double().requires_grad_(True) if R.is_param(k) else v.detach().clone().double() if v.is_floating_point() else v.clone())
            for k, v in sd0.items()}
    ref64 = R.forward(model_type, sd64, x.double(), ts.double(), md.double(), True, **flags)
    R.loss_mse(ref64, tgt.double())["total"].backward()
    net = net.cuda().set_precision("fp32").train()
    out = net(x.cuda(), ts.cuda(), md.cuda())
    loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
    loss.backward()
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-3
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * float(ref_loss)
    params = dict(net.named_parameters())
    worst = (0.0, 0.0, 0.0, 0.0, "")
    close_n = close_d = 0
    keys = [k for k, v in sd.items() if R.is_param(k) and v.grad is not None and not k.endswith((".conv1.bias", ".conv2.bias"))]
    # max-norm is decided by single ReLU-mask flips and jumps from parameter to parameter: its yardstick is the reference's
    # WORST fp32-vs-fp64 max-norm deviation over all parameters; relative L2 is judged per parameter
    y_max = max(rel_err(sd[k].grad, sd64[k].grad) for k in keys)
    for k in keys:
        v = sd[k]
        got = params[k].grad.cpu()
        e, e2 = rel_err(got, v.grad), rel_l2(got, v.grad)
        y_e, y_e2 = rel_err(v.grad, sd64[k].grad), rel_l2(v.grad, sd64[k].grad)      # reference fp32 vs fp64
        small = float((got - v.grad).abs().max()) < 1e-6
        assert (e <= 2 * y_max + 1e-3 and e2 <= 2 * y_e2 + 1e-3) or small, (k, e, e2, y_max, y_e2)
        if e > worst[0]:
            worst = (e, e2, y_e, y_e2, k)
        close_n += int(torch.isclose(got, v.grad, rtol=1e-3, atol=1e-3 * float(v.grad.abs().max())).sum())
        close_d += got.numel()
    print(f"{model_type}: worst gradient max-norm {worst[0]:.2e} / relL2 {worst[1]:.2e} on {worst[4]} "
          f"(reference fp32-vs-fp64 there: {worst[2]:.2e} / {worst[3]:.2e}); element-wise allclose(1e-3) pass fraction "
          f"{close_n / max(close_d, 1):.6f}")
    # bf16 throughput mode on the same shape: finite, and close in relative L2
    net.zero_grad(set_to_none=True)
    net.load_state_dict(sd0)
    net.set_precision("bf16")
    out16 = net(x.cuda(), ts.cuda(), md.cuda())
    assert torch.isfinite(out16).all()
    assert rel_l2(out16.detach().cpu(), ref.detach()) < 0.1


@pytest.mark.parametrize("model_type", ["unet", "unet++"])
def test_graphed_inference_matches_eager(mau, model_type):
    """hipGraph-captured inference (app path) == the eager eval forward, also for new input values."""
    flags = {} if model_type == "unet++" else dict(temporal_embeddings=True, metadata_embeddings=True)
    torch.manual_seed(3)
    net = mau.UrbanPredictor(model_type, 23, 12, 16, 8, 16, 24, 2, base_filters=8, **flags).cuda().eval()
    g = torch.Generator().manual_seed(4)
    mk = lambda: (torch.randn(1, 23, 96, 96, generator=g).cuda(), torch.randn(1, 12, generator=g).cuda(), torch.randn(1, 8, generator=g).cuda())
    a, b = mk(), mk()
    sess = mau.GraphedInference(net, *a)
    with torch.no_grad():
        ref_a, ref_b = net(*a), net(*b)
    assert torch.equal(sess(*a), ref_a)
    assert torch.equal(sess(*b), ref_b)
    with pytest.raises(ValueError):
        sess(torch.zeros(2, 23, 96, 96, device="cuda"), b[1], b[2])
    # host tensors go straight into the session's buffers (one copy, not host -> device -> buffer); a caller that fills ``sess.inputs``
    # itself pays no copy at all; ``clone_output=False`` hands out the session's own output buffer
    assert torch.equal(sess(*(t_.cpu() for t_ in a)), ref_a)
    for dst, src in zip(sess.inputs, b):
        dst.copy_(src)
    assert torch.equal(sess(*sess.inputs), ref_b)
    sess2 = mau.GraphedInference(net, *a, clone_output=False)
    o1 = sess2(*a)
    assert torch.equal(o1, ref_a) and sess2(*b).data_ptr() == o1.data_ptr() and torch.equal(o1, ref_b)


@pytest.mark.parametrize("opt_kw", [dict(fused=True), dict(foreach=True)])
def test_multi_step_training_tracks_oracle(mau, opt_kw):
    """Several optimizer steps: the weights used by step k must be the weights written by step k-1.
    (Regression: fused AdamW updates parameters in place WITHOUT bumping Tensor._version, which once made a
    version-keyed packed-weight cache serve stale weights from the second step on.)"""
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    torch.manual_seed(21)
    net = mau.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=8, **flags)
    sd = R.clone_state({k: v.clone() for k, v in net.state_dict().items()}, requires_grad=True)
    ref_opt = torch.optim.AdamW([sd[k] for k in sd if R.is_param(k)], lr=3e-3, weight_decay=1e-3)
    net = net.cuda().set_precision("fp32").train()
    opt = torch.optim.AdamW(net.parameters(), lr=3e-3, weight_decay=1e-3, **opt_kw)
    g = torch.Generator().manual_seed(22)
    ref_losses, losses = [], []
    for step in range(5):
        x, ts, md, tgt = torch.randn(2, 6, 32, 32, generator=g), torch.randn(2, 10, generator=g), torch.randn(2, 4, generator=g), torch.randn(2, 2, 32, 32, generator=g)
        loss_ref, _, _ = R.train_step("unet", sd, ref_opt, x, ts, md, tgt, **flags)
        out = net(x.cuda(), ts.cuda(), md.cuda())
        loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
        loss.backward()
        opt.step()
        opt.zero_grad()
        ref_losses.append(float(loss_ref))
        losses.append(float(loss))
    # Step 0 runs on identical weights: forward parity (1e-5).  From then on two fp32 implementations drift apart through
    # AdamW (m / sqrt(v) turns rounding noise in near-zero gradient entries into lr-sized steps): 3.7e-3 at step 3 was measured
    # after nothing but the rounding of the upsample's source coordinate changed.  Stale packed weights -- what this test is
    # for -- leave the loss at the initial model's level instead: 8 % off by step 3 (1.145 -> 1.056 in the oracle).
    for k, (a, b) in enumerate(zip(losses, ref_losses)):
        assert abs(a - b) < (1e-5 if k == 0 else 1e-2) * abs(b), (k, losses, ref_losses)
    assert ref_losses[3] < 0.95 * ref_losses[0]          # (the margin the 1e-2 band relies on)
    # with stale weights the eval output after training would equal the initial model's: check it moved with the oracle
    net.eval()
    with torch.no_grad():
        out_eval = net(x.cuda(), ts.cuda(), md.cuda()).cpu()
        ref_eval = R.forward("unet", {k: v.detach() for k, v in sd.items()}, x, ts, md, False, **flags)
    assert rel_err(out_eval, ref_eval) < 2e-2


@pytest.mark.parametrize("flags", [dict(temporal_embeddings=False, metadata_embeddings=True), dict(temporal_embeddings=True, metadata_embeddings=True)])
def test_metadata_sweep_reuses_encoder(mau, flags):
    """forward_metadata_sweep (encoder once at B=1) == the reference's way (tile repeated B times), eval mode."""
    torch.manual_seed(5)
    net = mau.UrbanPredictor("unet", 23, 12, 16, 8, 16, 24, 2, base_filters=8, **flags).cuda().eval()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 23, 62, 62, generator=g).cuda()
    ts = torch.randn(1, 12, generator=g).cuda()
    md = torch.randn(7, 8, generator=g).cuda()
    with torch.no_grad():
        ref = net(x.expand(7, -1, -1, -1).contiguous(), ts.expand(7, -1).contiguous(), md)
    got = net.forward_metadata_sweep(x, ts, md)
    assert torch.equal(got, ref)
    net.train()
    with pytest.raises(RuntimeError):
        net.forward_metadata_sweep(x, ts, md)
    with pytest.raises(NotImplementedError):
        mau.UrbanPredictor("unet++", 23, 12, 16, 8, 16, 24, 2, base_filters=8).cuda().eval().forward_metadata_sweep(x, ts, md)


def test_frozen_inference_matches_and_unfreezes(mau):
    """freeze_inference: same outputs with the packed weights / folded BN coefficients computed once; train(),
    load_state_dict and set_precision drop the frozen copies so that later forwards see the new parameters."""
    torch.manual_seed(11)
    net = mau.UrbanPredictor("unet", 23, 12, 16, 8, 16, 24, 2, base_filters=8, temporal_embeddings=False).cuda().eval()
    x, ts, md = torch.randn(2, 23, 48, 40).cuda(), torch.randn(2, 12).cuda(), torch.randn(2, 8).cuda()
    with torch.no_grad():
        ref = net(x, ts, md)
        net.freeze_inference()
        assert torch.equal(net(x, ts, md), ref) and torch.equal(net(x, ts, md), ref)
        blk = net.model.conv2_0
        assert blk._frozen is not None and "wf" in blk._frozen[0] and "wf" in blk._frozen[1]
        # parameters change under a frozen session only through the documented doors
        sd = {k: (v + 0.05 * torch.randn_like(v) if v.is_floating_point() else v) for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        assert blk._frozen is None
        new = net(x, ts, md)
        assert not torch.equal(new, ref)
        net.freeze_inference()
        assert torch.equal(net(x, ts, md), new)
    net.train()
    assert blk._frozen is None
    net.eval().freeze_inference().set_precision("fp32")
    assert blk._frozen is None


@pytest.mark.parametrize("model_type,base", [("unet", 16), ("unet++", 16), ("unet++", 64)])
def test_virtual_concat_and_fused_pool_are_bitwise_equal_to_materialised(mau, model_type, base, monkeypatch):
    """The decoder's torch.cat([skip, up], 1) (src/model.py:279-282) read by the conv loader from two tensors must give,
    bit for bit, the results of the materialised concat buffer (same K order, same arithmetic): outputs, loss and every
    gradient of one bf16 training step at base_filters=16 (all channel counts on 16-channel stage boundaries).  U-Net++ at
    base_filters=64 additionally runs on ROW BUFFERS (the nodes of a row written side by side, "cat of the earlier nodes"
    a view: functional.RowPrefix) -- same bitwise requirement, plus an eval forward."""
    flags = {} if model_type == "unet++" else dict(temporal_embeddings=False, metadata_embeddings=True)
    g = torch.Generator().manual_seed(31)
    x, ts, md = torch.randn(2, 6, 48, 40, generator=g).cuda(), torch.randn(2, 10, generator=g).cuda(), torch.randn(2, 4, generator=g).cuda()
    tgt = torch.randn(2, 2, 48, 40, generator=g).cuda()
    res = []
    for virt in ("1", "0"):
        monkeypatch.setattr(__import__("mau_amd.model", fromlist=["x"]), "_VIRTUAL_CONCAT", virt == "1")
        torch.manual_seed(30)
        net = mau.UrbanPredictor(model_type, 6, 10, 16, 4, 16, 24, 2, base_filters=base, **flags).cuda().set_precision("bf16").train()
        out = net(x, ts, md)
        loss = mau.compute_loss_mse(out, tgt)["total"]
        loss.backward()
        net.eval()
        with torch.no_grad():
            ev = net(x, ts, md)
        res.append((out.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, ev))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][3], res[1][3])
    assert res[0][2].keys() == res[1][2].keys()
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
def test_training_step_is_bitwise_reproducible(mau, prec):
    """Two identical steps from identical state give bit-identical outputs, loss, gradients and BatchNorm buffers: every
    reduction of the path (BatchNorm statistics, split-K weight gradient, loss, head) is slab + fixed-order, no atomics --
    in the fp32 parity mode too (its weight gradient wrote with float atomics until round 3)."""
    g = torch.Generator().manual_seed(41)
    x, ts, md = torch.randn(4, 6, 64, 64, generator=g).cuda(), torch.randn(4, 10, generator=g).cuda(), torch.randn(4, 4, generator=g).cuda()
    tgt = torch.randn(4, 2, 64, 64, generator=g).cuda()
    res = []
    for _ in range(2):
        torch.manual_seed(40)
        net = mau.UrbanPredictor("unet", 6, 10, 16, 4, 16, 24, 2, base_filters=32, temporal_embeddings=False).cuda().set_precision(prec).train()
        out = net(x, ts, md)
        loss = mau.compute_loss_mse(out, tgt)["total"]
        loss.backward()
        res.append((out.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in net.state_dict().items() if "running_" in k}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k


@pytest.mark.parametrize("model_type", ["unet", "unet++"])
def test_lstm_side_stream_overlap_is_bitwise_neutral(mau, model_type, monkeypatch):
    """In training the TemporalEncoder runs on a side stream (forward beside the first encoder blocks, backward -- replayed by
    autograd on its forward stream -- beside the encoder's backward): outputs, loss and EVERY gradient must equal the
    single-stream run bit for bit, at the reference's sequence length, three steps in a row (stream joins, allocator reuse)."""
    g = torch.Generator().manual_seed(51)
    x, ts, md = torch.randn(3, 6, 64, 64, generator=g).cuda(), torch.randn(3, 828, generator=g).cuda(), torch.randn(3, 4, generator=g).cuda()
    tgt = torch.randn(3, 2, 64, 64, generator=g).cuda()
    res = []
    for flag in ("0", "1"):
        monkeypatch.setattr(__import__("mau_amd.model", fromlist=["x"]), "_OVERLAP_LSTM", flag == "1")
        torch.manual_seed(50)
        net = mau.UrbanPredictor(model_type, 6, 828, 16, 4, 16, 96, 2, base_filters=32, temporal_embeddings=True).cuda().set_precision("bf16").train()
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
        steps = []
        for _ in range(3):
            out = net(x, ts, md)
            out = out[-1] if isinstance(out, (list, tuple)) else out
            loss = mau.compute_loss_mse(out, tgt)["total"]
            loss.backward()
            steps.append((out.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
            opt.step()
            opt.zero_grad()
        res.append(steps)
    for (o0, l0, g0), (o1, l1, g1) in zip(*res):
        assert torch.equal(o0, o1) and torch.equal(l0, l1)
        assert g0.keys() == g1.keys() and any("temporal_encoder.lstm" in k for k in g0)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("name", ["g5_unet_even.npz", "g5_unet_odd.npz", "g6_unetpp.npz"])
def test_fp16_mode_full_model(mau, name):
    """fp16 mode (v_mfma_f32_32x32x16_f16, fp32 accumulation; BASELINE configs[4]): eval and train outputs against the
    reference fixtures.  fp16 carries 3 more mantissa bits than bf16, so it is held to the bf16 yardstick without slack
    factor; the training gradients (no loss scaling) are checked in relative L2 against the same yardstick."""
    d = load_npz(name)
    net, m = build(mau, d, "fp16")
    out_yard, grad_yard = _autocast_yardstick(d, m)
    x, ts, md, tgt = (t(d[k]).cuda() for k in ("x", "ts", "md", "tgt"))
    net.eval()
    with torch.no_grad():
        e_eval = rel_l2(net(x, ts, md).cpu(), t(d["out_eval"]))
    net.train()
    out = net(x, ts, md)
    e_out = rel_l2(out.detach().cpu(), t(d["out_train"]))
    loss = mau.compute_loss_mse(out, tgt)["total"]
    loss.backward()
    params = dict(net.named_parameters())
    gref = sub(d, "grad")
    num = sum(float(((params[k].grad.cpu().double() - g.double()) ** 2).sum()) for k, g in gref.items())
    den = sum(float((g.double() ** 2).sum()) for g in gref.values())
    e_grad = (num / den) ** 0.5
    print(f"{name}: fp16 eval relL2 {e_eval:.4f}, train relL2 {e_out:.4f} (bf16 autocast yardstick {out_yard:.4f}); grad relL2 {e_grad:.4f} ({grad_yard:.4f})")
    assert e_eval <= 5e-3
    assert e_out <= out_yard + 5e-3
    assert e_grad <= grad_yard + 2e-2


CONFIG5 = [(6, 4, dict(temporal_embeddings=False, metadata_embeddings=True)), (23, 8, dict(temporal_embeddings=True, metadata_embeddings=True))]


@pytest.mark.parametrize("cin,nmeta,flags", CONFIG5, ids=["6ch-4meta", "23ch-8meta-app"])
def test_config5_512x512_inference_vs_oracle(mau, cin, nmeta, flags):
    """BASELINE configs[4] / the Streamlit app's call (app/model_utils.py:102-109, app/processing_utils.py:112-177): base-64
    U-Net, eval mode, ONE 512x512 tile, against the CPU oracle on the same seeded weights and inputs.  BatchNorm running
    statistics are first moved off their initial (0, 1) by two training-mode forwards of the oracle, so that the folded
    BN+ReLU epilogue is exercised with non-trivial coefficients.
      fp32 parity mode <= 1e-3 max-norm;  bf16 / fp16 by relative L2 (<= 3e-2 / <= 5e-3);
    through the plain eval forward, ``freeze_inference()`` and ``GraphedInference`` (which must agree bit for bit)."""
    torch.manual_seed(50 + cin)
    net = mau.UrbanPredictor("unet", cin, 12, 64, nmeta, 64, 96, 2, base_filters=64, **flags)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(51)
    with torch.no_grad():
        for _ in range(2):      # running statistics: two train-mode passes of the oracle at 64x64 (updates sd's buffers in place)
            R.forward("unet", sd, torch.randn(4, cin, 64, 64, generator=g), torch.randn(4, 12, generator=g), torch.randn(4, nmeta, generator=g), True, **flags)
    x, ts, md = torch.randn(1, cin, 512, 512, generator=g), torch.randn(1, 12, generator=g), torch.randn(1, nmeta, generator=g)
    with torch.no_grad():
        ref = R.forward("unet", sd, x, ts, md, False, **flags)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    xc, tc, mc = x.cuda(), ts.cuda(), md.cuda()
    for prec, metric, tol in (("fp32", rel_err, 1e-3), ("bf16", rel_l2, 3e-2), ("fp16", rel_l2, 5e-3)):
        net.set_precision(prec)
        with torch.no_grad():
            plain = net(xc, tc, mc)
            net.freeze_inference()
            frozen = net(xc, tc, mc)
            sess = mau.GraphedInference(net, xc, tc, mc)
            graphed = sess(xc, tc, mc)
            net.freeze_inference(False)
        e = metric(plain.cpu(), ref)
        print(f"config5 {cin}ch {prec}: {metric.__name__} {e:.2e}")
        assert plain.shape == (1, 2, 512, 512) and e <= tol, (prec, e)
        assert torch.equal(plain, frozen) and torch.equal(plain, graphed), prec


def test_metadata_sweep_vs_oracle_250(mau):
    """forward_metadata_sweep at the reference's production tile (250x250x23, 8 metadata features; the sensitivity sweeps
    of test/metadata_sensitivity.py:294-311 repeat one tile B times) against the ORACLE's repeated-tile eval forward."""
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    torch.manual_seed(60)
    net = mau.UrbanPredictor("unet", 23, 12, 32, 8, 32, 24, 2, base_filters=16, **flags)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(61)
    with torch.no_grad():
        R.forward("unet", sd, torch.randn(4, 23, 62, 62, generator=g), torch.randn(4, 12, generator=g), torch.randn(4, 8, generator=g), True, **flags)
    B = 5
    x, ts, md = torch.randn(1, 23, 250, 250, generator=g), torch.randn(1, 12, generator=g), torch.randn(B, 8, generator=g)
    with torch.no_grad():
        ref = R.forward("unet", sd, x.expand(B, -1, -1, -1), ts.expand(B, -1), md, False, **flags)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    for prec, metric, tol in (("fp32", rel_err, 1e-3), ("bf16", rel_l2, 3e-2)):
        net.set_precision(prec)
        got = net.forward_metadata_sweep(x.cuda(), ts.cuda(), md.cuda())
        assert got.shape == (B, 2, 250, 250)
        assert metric(got.cpu(), ref) <= tol, prec


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_g10_unetpp_deep_supervision(mau, prec):
    """deep_supervision=True: list of four bare 1x1-conv outputs (no tanh, src/model.py:180-185), gradients of their
    summed MSE; fp32 <= 1e-3 against the reference fixture, bf16 by relative L2."""
    d = load_npz("g10_unetpp_deepsup.npz")
    net, m = build(mau, d, prec)
    x, ts, md, tgt = (t(d[k]).cuda() for k in ("x", "ts", "md", "tgt"))
    tol_out, tol_l2 = (1e-3, None) if prec == "fp32" else (None, 6e-2)
    net.eval()
    with torch.no_grad():
        outs = net(x, ts, md)
    assert isinstance(outs, list) and len(outs) == 4
    for j, o in enumerate(outs):
        ref = t(d[f"out_eval{j}"])
        assert o.shape == ref.shape
        assert (rel_err(o.cpu(), ref) < tol_out) if tol_out else (rel_l2(o.cpu(), ref) < tol_l2)
    net.train()
    outs = net(x, ts, md)
    for j, o in enumerate(outs):
        ref = t(d[f"out_train{j}"])
        assert (rel_err(o.cpu(), ref) < tol_out) if tol_out else (rel_l2(o.cpu(), ref) < tol_l2)
    loss = sum(mau.compute_loss_mse(o, tgt)["total"] for o in outs)
    assert abs(float(loss) - float(d["loss"][0])) < (1e-4 if prec == "fp32" else 5e-2) * abs(float(d["loss"][0]))
    loss.backward()
    params = dict(net.named_parameters())
    for k in m["nograd"]:
        assert params[k].grad is None, k
    if prec == "fp32":
        for k, gref in sub(d, "grad").items():
            got = params[k].grad.cpu()
            if k.endswith(".conv1.bias") or k.endswith(".conv2.bias"):
                continue                                         # identically zero here, rounding noise in the reference
            assert rel_err(got, gref) < 1e-3 or float((got - gref).abs().max()) < 1e-7, k


MATRIX = [
    # model_type, C, T, F, out_channels, base_filters, flags, (B, H, W)
    ("unet", 5, 7, 3, 2, 6, dict(temporal_embeddings=True, metadata_embeddings=False), (2, 37, 41)),   # temporal only, no channel count a multiple of 8
    ("unet", 6, 6, 4, 1, 8, dict(temporal_embeddings=True, metadata_embeddings=True), (2, 32, 32)),    # single output channel: no tanh
    ("unet", 9, 5, 2, 3, 8, dict(temporal_embeddings=False, metadata_embeddings=True), (1, 48, 32)),   # three output channels, batch of one
    ("unet++", 4, 6, 5, 1, 6, {}, (2, 35, 33)),
    ("unet++", 6, 4, 4, 2, 8, {}, (3, 32, 48)),
    ("unet", 6, 4, 4, 2, 8, dict(temporal_embeddings=False, metadata_embeddings=True), (2, 16, 16)),    # bottleneck of 1x1 pixels
    ("unet", 6, 4, 4, 2, 8, dict(temporal_embeddings=True, metadata_embeddings=True), (3, 17, 19)),     # 17 -> 8 -> 4 -> 2 -> 1, every up-step re-sized
    ("unet++", 6, 4, 4, 2, 8, {}, (2, 18, 17)),
]


@pytest.mark.parametrize("cfg", MATRIX, ids=[f"{c[0]}-C{c[1]}-o{c[4]}-b{c[5]}-{c[7][1]}x{c[7][2]}" for c in MATRIX])
def test_fp32_matches_oracle_on_constructor_matrix(mau, cfg):
    """Corners of the constructor surface the reference fixtures do not visit (temporal-only embeddings, out_channels
    1 and 3, channel counts that are not multiples of 8, batch of one): the module in fp32 mode against the pinned
    oracle on the module's own randomly initialised parameters -- outputs, loss and every gradient <= 1e-3."""
    import torch.nn.functional as F
    model_type, C, T, Fm, oc, base, flags, (B, H, W) = cfg
    torch.manual_seed(sum(map(ord, model_type)) + C + oc + base)
    net = mau.UrbanPredictor(model_type, C, T, 8, Fm, 8, 12, oc, base_filters=base, **flags).cuda().set_precision("fp32")
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():                                   # non-trivial BatchNorm affine parameters / running statistics
        for k, v in net.state_dict().items():
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith("running_mean") or ".bn" in k and k.endswith("bias"):
                v.copy_(0.3 * torch.randn(v.shape, generator=g))
            elif ".bn" in k and k.endswith("weight"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    x, ts, md = torch.randn(B, C, H, W, generator=g), torch.randn(B, T, generator=g), torch.randn(B, Fm, generator=g)
    tgt = torch.randn(B, oc, H, W, generator=g)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for training in (False, True):
        net.train(training)
        sd = R.clone_state(sd_cpu, requires_grad=training)
        ref = R.forward(model_type, sd, x, ts, md, training, **flags)
        ctx = torch.enable_grad() if training else torch.no_grad()
        with ctx:
            out = net(x.cuda(), ts.cuda(), md.cuda())
        assert out.shape == ref.shape
        assert rel_err(out.detach().cpu(), ref.detach()) < 1e-3
    loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
    ref_loss = F.mse_loss(ref, tgt)
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    loss.backward()
    ref_loss.backward()
    # Per tensor <= 1e-3, except that ONE ReLU mask may legitimately flip where a pre-activation lies within fp32
    # rounding of zero (conv0_1.bn2 channel 1 of the unet++/base 6 case: one pixel of 13,860; the reference's own
    # operators are on the other side by ~1e-7): that moves the small per-channel sums by up to 6e-3 of their size.
    # Such tensors must still agree to 1e-2, and the whole gradient, taken as one vector, to 1e-3 in relative L2.
    num = den = 0.0
    for k, p in net.named_parameters():
        gref = sd[k].grad
        if gref is None:
            assert p.grad is None, k
            continue
        if k.endswith(".conv1.bias") or k.endswith(".conv2.bias"):
            continue                                        # exactly zero here, rounding noise in the reference operators
        got = p.grad.cpu()
        e = rel_err(got, gref)
        assert e < 1e-2 or float((got - gref).abs().max()) < 1e-6, (k, e)
        num += float(((got - gref).double() ** 2).sum())
        den += float((gref.double() ** 2).sum())
    assert (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5


@pytest.mark.parametrize("cfg", MATRIX, ids=[f"{c[0]}-C{c[1]}-o{c[4]}-b{c[5]}-{c[7][1]}x{c[7][2]}" for c in MATRIX])
def test_bf16_sane_on_constructor_matrix(mau, cfg):
    """The same corners through the bf16 kernels (64-wide variants with tiny channel counts, 1-3 K chunks, ragged
    tiles): outputs within bf16 distance of the fp32 oracle, finite gradients of the right overall size."""
    import torch.nn.functional as F
    model_type, C, T, Fm, oc, base, flags, (B, H, W) = cfg
    torch.manual_seed(3 + C + oc + base)
    net = mau.UrbanPredictor(model_type, C, T, 8, Fm, 8, 12, oc, base_filters=base, **flags).cuda().set_precision("bf16").train()
    g = torch.Generator().manual_seed(17)
    x, ts, md = torch.randn(B, C, H, W, generator=g), torch.randn(B, T, generator=g), torch.randn(B, Fm, generator=g)
    tgt = torch.randn(B, oc, H, W, generator=g)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    # eval mode (running statistics): well conditioned at every size
    net.eval()
    with torch.no_grad():
        out_e = net(x.cuda(), ts.cuda(), md.cuda())
        ref_e = R.forward(model_type, R.clone_state(sd_cpu), x, ts, md, False, **flags)
    assert out_e.dtype == torch.float32 and rel_l2(out_e.cpu(), ref_e) < 0.1
    net.train()
    sd = R.clone_state(sd_cpu, requires_grad=True)
    ref = R.forward(model_type, sd, x, ts, md, True, **flags)
    out = net(x.cuda(), ts.cuda(), md.cuda())
    # train-mode BatchNorm over a handful of values (1x1-pixel bottleneck, batch 2-3) amplifies bf16 rounding without
    # bound (two samples normalise to +-1 whatever their distance): compare the outputs only when the deepest level
    # still averages over >= 16 values per channel
    well_conditioned = B * (H // 16) * (W // 16) >= 16
    if well_conditioned:
        assert rel_l2(out.detach().cpu(), ref.detach()) < 0.1
    loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
    ref_loss = F.mse_loss(ref, tgt)
    assert torch.isfinite(loss) and (not well_conditioned or abs(float(loss) - float(ref_loss)) < 0.05 * abs(float(ref_loss)))
    loss.backward()
    ref_loss.backward()
    n_ours = n_ref = 0.0
    for k, p in net.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None, k
            continue
        assert torch.isfinite(p.grad).all(), k
        if not (k.endswith(".conv1.bias") or k.endswith(".conv2.bias")):
            n_ours += float((p.grad.double() ** 2).sum())
            n_ref += float((sd[k].grad.double() ** 2).sum())
    assert not well_conditioned or abs((n_ours / n_ref) ** 0.5 - 1.0) < 0.3


def test_train_cli_writes_reference_checkpoint_and_reloads(mau, tmp_path):
    """two steps of ``mau_amd.train`` at the reference's production configuration (conf/config.yaml: 23
    channels, 250x250 tiles, 8 metadata features, B=16, l1-gradient-ssim loss, AdamW) write the ``.pth`` of
    src/train.py:303-319; ``checkpoint.load_model`` (the rules of app/model_utils.py:16-100) must rebuild a model whose
    eval output equals the trained model's bit for bit, and the dict must carry the reference's keys and value types."""
    from mau_amd import checkpoint, train
    from mau_amd.config import CONFIG
    old = CONFIG.MODELS_DIR
    CONFIG.MODELS_DIR = str(tmp_path)
    try:
        res = train.run(device="gpu", temporal_embeddings=False, metadata_embeddings=True, model_type="unet", jobid="t",
                        epochs=1, steps_per_epoch=2, precision="bf16")
    finally:
        CONFIG.MODELS_DIR = old
    path = res["checkpoint_path"]
    assert path is not None and path.endswith("urban-predictor-metaemb_trial_0_best_jobt.pth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "step", "model_state_dict", "optimizer_state_dict", "loss", "hyperparameters", "model_type",
                       "study_name", "trial_id", "metadata_input_length"}
    hp = ck["hyperparameters"]
    assert hp["target_channels"] == "after_ndvi,after_temp" and isinstance(hp["input_channels"], str) and "before_dw" in hp["input_channels"]
    assert ck["step"] == 2 and ck["model_type"] == "unet" and ck["metadata_input_length"] == 8 and len(ck["model_state_dict"]) == 138
    assert all(torch.isfinite(v).all() for v in ck["model_state_dict"].values() if v.is_floating_point())
    g = torch.Generator().manual_seed(70)
    x, ts, md = torch.randn(1, 23, 250, 250, generator=g).cuda(), torch.randn(1, 24, generator=g).cuda(), torch.randn(1, 8, generator=g).cuda()
    trained = res["model"].eval()
    with torch.no_grad():
        ref = trained(x, ts, md)
    loaded = checkpoint.load_model(path, device="cuda", spatial_channels=23, seq_len=CONFIG.dataset.temporal_length)
    assert isinstance(loaded, mau.UrbanPredictor) and not loaded.training
    with torch.no_grad():
        assert torch.equal(loaded(x, ts, md), ref)
    assert checkpoint.run_inference(loaded, x.cpu(), md.cpu(), ts.cpu()).shape == (1, 2, 250, 250)


def _train_n_steps(mau, model_type, prec, steps, graphed, seed=60, T=24, fused_opt=False, base_filters=16):
    """``steps`` training steps on a sequence of different batches; returns (losses, parameters, BN buffers)."""
    flags = {} if model_type == "unet++" else dict(temporal_embeddings=True, metadata_embeddings=True)
    torch.manual_seed(seed)
    net = mau.UrbanPredictor(model_type, 6, T, 16, 4, 16, 24, 2, base_filters=base_filters, **flags).cuda().set_precision(prec).train()
    opt = (mau.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-3) if fused_opt
           else torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-3, fused=True, capturable=True))
    crit = mau.compute_loss_mse_gradient
    g = torch.Generator().manual_seed(seed + 1)
    step = mau.GraphedTrainStep(net, opt, crit, warmup=2) if graphed else None
    losses = []
    for _ in range(steps):
        x, ts, md = torch.randn(3, 6, 64, 48, generator=g).cuda(), torch.randn(3, T, generator=g).cuda(), torch.randn(3, 4, generator=g).cuda()
        tgt = torch.randn(3, 2, 64, 48, generator=g).cuda()
        if graphed:
            losses.append(step(x, ts, md, tgt).clone())
        else:
            loss = crit(net(x, ts, md), tgt)["total"]
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(loss.detach().clone())
    if graphed:
        assert step.graph is not None and step.calls == steps
    # an EAGER eval forward after the last step must see the last step's weights (the replays change them behind Python's back)
    net.eval()
    with torch.no_grad():
        ev = net(x, ts, md)
    return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, ev


@pytest.mark.parametrize("fused_opt", [False, True])
@pytest.mark.parametrize("model_type,prec", [("unet", "bf16"), ("unet++", "bf16"), ("unet", "fp32")])
def test_graphed_train_step_matches_eager(mau, model_type, prec, fused_opt):
    """train_graph.GraphedTrainStep: forward + criterion + backward + fused AdamW captured ONCE into a hipGraph (call 3) and replayed
    (calls 4..6) must give, bit for bit, the losses, parameters, BatchNorm buffers and the following eval output of the same six steps
    launched kernel by kernel -- with a different batch every step (static input buffers refreshed by copies)."""
    a = _train_n_steps(mau, model_type, prec, 6, graphed=False, fused_opt=fused_opt)
    b = _train_n_steps(mau, model_type, prec, 6, graphed=True, fused_opt=fused_opt)
    for la, lb in zip(a[0], b[0]):
        assert torch.equal(la, lb), (a[0], b[0])
    assert float(a[0][0]) != float(a[0][5])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
    assert torch.equal(a[2], b[2])


def test_graphed_train_step_unetpp_row_buffers(mau, monkeypatch):
    """U-Net++ at the production width (base_filters=64: the nodes of a row live in row buffers, functional.RowPrefix) through
    GraphedTrainStep, against the eager steps.  Regression: the autograd context of a block used to hold the row-buffer view it returns
    (a reference cycle), the previous step's graph and its AccumulateGrad nodes survived into the capture, and hipStreamEndCapture
    crashed the process -- the narrow models of the test above never take the row-buffer path."""
    from mau_amd import functional as F_
    monkeypatch.setattr(F_, "_EMB_FOLD_MIN_WORK", 0)          # (and with the broadcast embedding folded, as at production size)
    a = _train_n_steps(mau, "unet++", "bf16", 5, graphed=False, fused_opt=True, base_filters=64)
    b = _train_n_steps(mau, "unet++", "bf16", 5, graphed=True, fused_opt=True, base_filters=64)
    for la, lb in zip(a[0], b[0]):
        assert torch.equal(la, lb), (a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
    assert torch.equal(a[2], b[2])


def test_capture_refuses_a_live_autograd_graph(mau):
    """A loss / output of an earlier step that still carries its grad_fn keeps that step's AccumulateGrad nodes alive, bound to the
    stream they were created on; a capture that meets them ends in hipStreamEndCapture taking the process down (round 3's records).
    The step driver now finds them BEFORE capturing and raises a Python error that names the cause; once the tensor is dropped
    the same object captures and replays."""
    torch.manual_seed(5)
    net = mau.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=8, temporal_embeddings=False, metadata_embeddings=True).cuda().train()
    opt = mau.AdamW(net.parameters(), lr=1e-3)
    step = mau.GraphedTrainStep(net, opt, mau.compute_loss_mse, warmup=1)
    g = torch.Generator().manual_seed(6)
    x, ts, md, tgt = torch.randn(2, 6, 32, 32, generator=g).cuda(), torch.randn(2, 10, generator=g).cuda(), torch.randn(2, 4, generator=g).cuda(), torch.randn(2, 2, 32, 32, generator=g).cuda()
    step(x, ts, md, tgt)                                         # warm-up (eager)
    kept = mau.compute_loss_mse(net(x, ts, md), tgt)["total"]    # user code holding on to a graph
    with pytest.raises(RuntimeError, match="autograd graph of an earlier step is still alive"):
        step(x, ts, md, tgt)
    assert step.graph is None
    del kept
    step.calls -= 1                                              # (the refused call did not happen)
    l1 = float(step(x, ts, md, tgt))
    l2 = float(step(x, ts, md, tgt))
    assert step.graph is not None and l2 < l1


def test_inplace_op_on_a_row_buffer_raises_in_backward(mau, monkeypatch):
    """U-Net++ row buffers are written by raw kernels through ``BNState.out_view`` and read through views; a torch in-place op on
    the buffer would make autograd rebuild the views' history from a base that has none.  It must not pass silently: the slots are
    saved tensors of their consumers, whose version check turns the bump into an error in backward."""
    from mau_amd import model as M
    seen = []
    orig = M.UrbanPredictor_unetpp._node

    def spy(self, block, skips, below, emb, out_view=None, **kw):
        if out_view is not None:
            seen.append(out_view)
        return orig(self, block, skips, below, emb, out_view, **kw)

    monkeypatch.setattr(M.UrbanPredictor_unetpp, "_node", spy)
    torch.manual_seed(5)
    net = mau.UrbanPredictor("unet++", 6, 10, 16, 4, 16, 24, 2, base_filters=64).cuda().set_precision("bf16").train()
    x, ts, md = torch.randn(2, 6, 32, 32).cuda(), torch.randn(2, 10).cuda(), torch.randn(2, 4).cuda()
    out = net(x, ts, md)
    assert seen and seen[0]._base is not None            # the slots are views of one buffer per row
    seen[0]._base.add_(0)                                # any in-place torch op on the buffer
    # (autograd's own words: either the saved-tensor version check, or "is a view and its base ... has been modified inplace.
    #  This view was created inside a custom Function ... This behavior is forbidden.")
    with pytest.raises(RuntimeError, match="modified by an inplace operation|has been modified inplace"):
        out.float().sum().backward()


def test_weight_gradient_stream_is_bitwise_neutral(mau, monkeypatch):
    """functional._OVERLAP_WGRAD = 2: the weight gradients of a backward pass run on their own stream and join ONCE, from an autograd
    engine callback when the pass has ended.  Same kernels: three optimizer steps -- the second with gradient ACCUMULATION (two
    backward passes before the step: the accumulating pass must fall back to the per-layer join, it adds on the main stream) --
    must leave bit-identical gradients, parameters and BatchNorm buffers to the one-stream run; torch's AdamW (no gradient arena)
    and mau_amd.AdamW (arena slots) both."""
    from mau_amd import functional as F_
    g = torch.Generator().manual_seed(81)
    data = [(torch.randn(2, 6, 48, 40, generator=g).cuda(), torch.randn(2, 10, generator=g).cuda(), torch.randn(2, 4, generator=g).cuda(),
             torch.randn(2, 2, 48, 40, generator=g).cuda()) for _ in range(4)]
    for arena in (False, True):
        res = []
        for mode in (0, 2):
            monkeypatch.setattr(F_, "_OVERLAP_WGRAD", mode)
            torch.manual_seed(80)
            net = mau.UrbanPredictor("unet", 6, 10, 16, 4, 16, 24, 2, base_filters=16, temporal_embeddings=True).cuda().set_precision("bf16").train()
            opt = mau.AdamW(net.parameters(), lr=1e-3) if arena else torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
            grads = []
            for step, passes in enumerate((1, 2, 1)):
                for k in range(passes):
                    x, ts, md, tgt = data[step + k]
                    mau.compute_loss_mse(net(x, ts, md), tgt)["total"].backward()
                grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            res.append((grads, {k: v.clone() for k, v in net.state_dict().items()}))
        for ga, gb in zip(res[0][0], res[1][0]):
            assert ga.keys() == gb.keys()
            for k in ga:
                assert torch.equal(ga[k], gb[k]), (arena, k)
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), (arena, k)


def test_single_launch_reductions_and_multi_pack_are_bit_identical(mau, monkeypatch):
    """Launch-tail fusions of round 3 against the forms they replace, same arithmetic in the same order: the single-launch slab
    reductions (ticket: the last workgroup runs the second level; functional._FUSED_REDUCE) vs two launches, and the one-launch
    multi-tensor weight pack (functional.PackGroup; functional._PACK_MULTI) vs one launch per layer: two training steps, everything equal."""
    from mau_amd import functional as F_
    g = torch.Generator().manual_seed(71)
    x, ts, md = torch.randn(4, 6, 96, 80, generator=g).cuda(), torch.randn(4, 10, generator=g).cuda(), torch.randn(4, 4, generator=g).cuda()
    tgt = torch.randn(4, 2, 96, 80, generator=g).cuda()
    res = []
    for fused, multi in ((True, "1"), (False, "0")):
        monkeypatch.setattr(F_, "_FUSED_REDUCE", fused)
        monkeypatch.setattr(F_, "_PACK_MULTI", multi == "1")
        torch.manual_seed(70)
        net = mau.UrbanPredictor("unet", 6, 10, 16, 4, 16, 24, 2, base_filters=32, temporal_embeddings=False).cuda().set_precision("bf16").train()
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, fused=True)
        outs = []
        for _ in range(2):
            out = net(x, ts, md)
            loss = mau.compute_loss_mse(out, tgt)["total"]
            loss.backward()
            grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            opt.step()
            opt.zero_grad()
            outs.append((out.detach().clone(), loss.detach().clone(), grads))
        res.append((outs, {k: v.clone() for k, v in net.state_dict().items()}))
    for (oa, la, ga), (ob, lb, gb) in zip(res[0][0], res[1][0]):
        assert torch.equal(oa, ob) and torch.equal(la, lb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    # the tickets of the single-launch reductions are left zeroed by every launch
    for tk in F_._TICKETS.values():
        assert int(tk.abs().sum()) == 0


def test_graphed_inference_survives_an_optimizer_step_elsewhere(mau):
    """the frozen-session cache key contains the process-global optimizer-step generation, so ANY optimizer step (of
    another model) makes the next eager frozen forward re-derive the packed weights and folded BatchNorm coefficients.  They are
    refreshed IN PLACE (persistent buffers): a live GraphedInference graph, which holds their addresses, must keep replaying
    correct results afterwards -- and after this model's own parameters change through ``mark_params_updated``."""
    torch.manual_seed(80)
    net = mau.UrbanPredictor("unet", 6, 10, 16, 4, 16, 24, 2, base_filters=16, temporal_embeddings=False).cuda().eval()
    g = torch.Generator().manual_seed(81)
    a = (torch.randn(1, 6, 64, 64, generator=g).cuda(), torch.randn(1, 10, generator=g).cuda(), torch.randn(1, 4, generator=g).cuda())
    sess = mau.GraphedInference(net, *a)
    with torch.no_grad():
        ref = net(*a)
    assert torch.equal(sess(*a), ref)
    other = torch.nn.Linear(4, 4).cuda()
    opt = torch.optim.SGD(other.parameters(), lr=0.1)
    other(torch.randn(2, 4).cuda()).sum().backward()
    opt.step()                                               # bumps the global generation
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]   # churn the allocator: freed blocks would be reused by now
    with torch.no_grad():
        eager = net(*a)                                      # frozen eager forward: refreshes the copies (in place)
    del junk
    assert torch.equal(eager, ref)
    assert torch.equal(sess(*a), ref)                        # the graph still reads valid, current buffers
    # this model's parameters change behind a frozen session only through the documented door: the graph sees the new weights
    with torch.no_grad():
        for p in net.parameters():
            p.data.mul_(1.01)
    mau.mark_params_updated()
    with torch.no_grad():
        new = net(*a)
    assert not torch.equal(new, ref)
    assert torch.equal(sess(*a), new)


def test_packs_follow_the_optimizer_and_sgd_tracks_the_oracle(mau):
    """The stale-pack regression guard without AdamW's m / sqrt(v) amplification: five SGD-with-momentum steps track the
    oracle to 2e-3 at EVERY step, and the packed weights used by step 1 differ from those of step 0."""
    from mau_amd import functional as F_
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    torch.manual_seed(21)
    net = mau.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=8, **flags)
    sd = R.clone_state({k: v.clone() for k, v in net.state_dict().items()}, requires_grad=True)
    ref_opt = torch.optim.SGD([sd[k] for k in sd if R.is_param(k)], lr=0.05, momentum=0.9)
    net = net.cuda().set_precision("fp32").train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(22)
    w = net.model.conv1_0.conv1.weight
    packs = []
    for step in range(5):
        x, ts, md, tgt = torch.randn(2, 6, 32, 32, generator=g), torch.randn(2, 10, generator=g), torch.randn(2, 4, generator=g), torch.randn(2, 2, 32, 32, generator=g)
        loss_ref, _, _ = R.train_step("unet", sd, ref_opt, x, ts, md, tgt, **flags)
        out = net(x.cuda(), ts.cuda(), md.cuda())
        packs.append(w._mau_pack[1].clone())
        loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
        loss.backward()
        opt.step()
        opt.zero_grad()
        assert abs(float(loss) - float(loss_ref)) < 2e-3 * abs(float(loss_ref)), (step, float(loss), float(loss_ref))
        if step == 0:
            # one plain SGD update from identical weights: every parameter within 1e-3 (gradient parity times lr; no amplification)
            for k, p in net.named_parameters():
                e = rel_l2(p.detach().cpu(), sd[k].detach())
                assert e < 1e-3, (k, e)
    assert not torch.equal(packs[0], packs[1]) and not torch.equal(packs[1], packs[2])
    # (After five momentum steps at this learning rate the weights in front of a BatchNorm -- scale-invariant directions -- have
    #  drifted apart by rounding noise x ReLU flips: 5e-2 measured on conv0_0.conv1, while the losses above stay within 2e-3.)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_fused_adamw_matches_torch_and_keeps_the_packs_fresh(mau, prec):
    """mau_amd.AdamW (csrc/optim.hip: AdamW on every convolution weight + both weight packs in one launch; torch's fused kernel for
    the small parameters) against torch.optim.AdamW on identical models, gradients through the same kernels: four steps, parameters
    and moments equal to 1e-5 of their size; after each step the packs the optimizer wrote must be what a fresh re-pack of the updated
    weights gives (eval forward bitwise equal to the forward after mau_amd.mark_params_updated()); the two optimizers read each
    other's state_dict."""
    from mau_amd import functional as F_
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    g = torch.Generator().manual_seed(101)
    batches = [(torch.randn(2, 23, 62, 50, generator=g).cuda(), torch.randn(2, 10, generator=g).cuda(), torch.randn(2, 8, generator=g).cuda(),
                torch.randn(2, 2, 62, 50, generator=g).cuda()) for _ in range(4)]
    nets, opts = [], []
    for kind in ("torch", "mau"):
        torch.manual_seed(100)
        net = mau.UrbanPredictor("unet", 23, 10, 16, 8, 16, 24, 2, base_filters=16, **flags).cuda().set_precision(prec).train()
        nets.append(net)
        opts.append(torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=1e-2, fused=True) if kind == "torch"
                    else mau.AdamW(net.parameters(), lr=2e-3, weight_decay=1e-2))
    for step, (x, ts, md, tgt) in enumerate(batches):
        losses = []
        for net, opt in zip(nets, opts):
            loss = mau.compute_loss_mse(net(x, ts, md), tgt)["total"]
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        # the packs written by the fused optimizer == a re-pack from the updated master weights
        net = nets[1].eval()
        with torch.no_grad():
            a = net(x, ts, md)
            mau.mark_params_updated()
            b = net(x, ts, md)
        net.train()
        assert torch.equal(a, b), step
        # the two trainings stay together: same loss at every step (the first one bitwise: identical weights and kernels)
        assert losses[0] == losses[1] if step == 0 else abs(losses[0] - losses[1]) <= (2e-3 if prec == "fp32" else 1e-2) * abs(losses[0]), (step, losses)
        if step == 0:                                        # one step from identical state: tight
            for (k, p), (_, q) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
                assert float((p - q).abs().max()) <= 1e-6 + 1e-5 * float(p.abs().max()), k
    # Later steps are judged by the losses above, not per parameter: Adam's update is lr * m / sqrt(v) -- of size lr whatever the
    # gradient's size -- so a last-bit difference in a ~0 gradient (BatchNorm biases start at 0; bf16 activations downstream of a
    # weight that rounded the other way) moves a parameter by up to 2 lr per step: 7e-2 relative measured on conv0_0.bn1.bias after
    # four bf16 steps with BOTH optimizers correct.  In fp32 the weights that start away from zero still agree closely:
    if prec == "fp32":
        for (k, p), (_, q) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
            if k.endswith("weight") and p.dim() > 1:
                assert rel_l2(q.detach().cpu(), p.detach().cpu()) < 1e-2, k
    w = nets[1].model.conv2_0.conv1.weight
    assert w.grad is None and w._mau_grad_slot is not None
    # state_dict interchange
    sd_t, sd_m = opts[0].state_dict(), opts[1].state_dict()
    k0 = next(iter(sd_m["state"]))                           # (parameters that never saw a gradient -- the unused encoders -- have no state)
    assert sd_t["state"].keys() == sd_m["state"].keys() and set(sd_m["state"][k0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd_m["state"][k0]["step"]) == 4.0
    fresh = mau.AdamW(nets[0].parameters(), lr=2e-3, weight_decay=1e-2)
    fresh.load_state_dict(sd_t)
    torch.optim.AdamW(nets[1].parameters(), lr=2e-3, weight_decay=1e-2, fused=True).load_state_dict(sd_m)
    x, ts, md, tgt = batches[0]
    mau.compute_loss_mse(nets[0](x, ts, md), tgt)["total"].backward()
    fresh.step()
    assert all(torch.isfinite(p).all() for p in nets[0].parameters())
