"""Parity at the sizes the bench runs (BASELINE configs[1] and configs[2]): ONE 16-bit training step of the full-width network
through the product's step driver -- ``GraphedTrainStep`` (hipGraph replay) + ``mau_amd.AdamW`` -- against the CPU oracle on the
same seed: output, loss, every parameter's gradient, the post-step weights, the BatchNorm buffers.

Why a test of its own: the kernels these shapes select (the big-tile ``<128,4,8>`` / ``<64,4,8>`` 16x16x32 stage-pair loop,
``wgrad16_kernel``, the gradient arena, ``adamw_pack_kernel``) are never selected by the small fixtures.  The test ASSERTS that
selection (pixel-tile counts of the level-0 and level-1 layers), so a smaller batch cannot silently test other code.

Yardstick (as in test_gpu_model.py): the error of the REFERENCE's own operators under torch's CPU bf16 autocast against their
fp32 result on the same inputs -- the inherent bf16 noise of this network.  The HIP path (bf16 operands, fp32 accumulation, fp32
BatchNorm statistics, fp32 master weights) must stay within 1.5x that (+ a floor), per parameter and globally.
CPU cost on the GPU box: fp32 oracle step + autocast pass, ~1-2 min per test (``MAU_TEST_FULL_B`` lowers the U-Net batch).
"""
import os

import pytest
import torch

from oracle import unet_ref as R
from tests.helpers import rel_err, rel_l2

pytestmark = pytest.mark.gpu
LR, WD = 1e-4, 1e-3                     # conf/config.yaml:41,48,52 (bench.py's optimizer)


_CONFIG2 = {}


def _config2_case():
    """The network, inputs and fp32 oracle step of BASELINE configs[1], built once per test process (the oracle step costs ~1 min of
    host time at B = 32; the bf16 and the fp32 test below compare against the same one)."""
    if not _CONFIG2:
        import mau_amd as mau
        B = int(os.environ.get("MAU_TEST_FULL_B", "32"))
        flags = dict(temporal_embeddings=False, metadata_embeddings=True)
        torch.manual_seed(0)
        net = mau.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, base_filters=64, **flags)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        batch = R.synthetic_batch(B)
        _CONFIG2.update(B=B, flags=flags, net=net, sd0=sd0, batch=batch, ref=_oracle_step("unet", sd0, batch, flags, autocast=False))
    return _CONFIG2


def _oracle_step(model_type, sd0, batch, flags, autocast):
    """(out, loss, grads, sd after one AdamW step) of the oracle in fp32, or with the forward under CPU bf16 autocast."""
    x, ts, md, tgt = batch
    sd = R.clone_state(sd0, requires_grad=True)
    params = [sd[k] for k in sd if R.is_param(k)]
    opt = torch.optim.AdamW(params, lr=LR, weight_decay=WD)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = R.forward(model_type, sd, x, ts, md, True, **flags)
        out = out.float()
    else:
        out = R.forward(model_type, sd, x, ts, md, True, **flags)
    loss = R.loss_mse(out, tgt)["total"]
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in sd.items() if R.is_param(k) and v.grad is not None}
    opt.step()
    return out.detach(), float(loss), grads, {k: v.detach().clone() for k, v in sd.items()}


def _hip_step(mau, net, sd0, batch, warm_batch):
    """One step from ``sd0`` with a fresh optimizer THROUGH THE GRAPH: the step driver captures on its third call, so two eager
    warm-up steps and the capturing step run first (on another batch); then weights, BatchNorm buffers and optimizer state are
    put back IN PLACE (the graph holds their addresses) and one replay is the step under test."""
    opt = mau.AdamW(net.parameters(), lr=LR, weight_decay=WD)
    step = mau.GraphedTrainStep(net, opt, mau.compute_loss_mse, warmup=2, copy_inputs=True)
    for _ in range(3):
        step(*warm_batch)
    assert step.graph is not None
    net.load_state_dict(sd0)                                    # in-place copies; bumps the versions -> the packs are rebuilt before the replay
    for st in opt.state.values():
        st["step"].zero_()
        st["exp_avg"].zero_()
        st["exp_avg_sq"].zero_()
    loss = step(*batch)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters() if p.grad is not None}
    return step.outputs.float().cpu(), float(loss), grads, {k: v.detach().cpu() for k, v in net.state_dict().items()}


def _compare(tag, hip, ref, yard, sd0, nograd_ok=()):
    out, loss, grads, sd1 = hip
    rout, rloss, rgrads, rsd1 = ref
    yout, yloss, ygrads, ysd1 = yard
    e_out, y_out = rel_l2(out, rout), rel_l2(yout, rout)
    print(f"{tag}: output relL2 {e_out:.4f} (autocast yardstick {y_out:.4f}); loss {loss:.6f} vs {rloss:.6f} (autocast {yloss:.6f})")
    assert e_out <= 1.5 * y_out + 1e-2
    assert abs(loss - rloss) <= 1.5 * abs(yloss - rloss) + 5e-3 * abs(rloss)
    assert set(grads) == set(rgrads) - set(nograd_ok), set(grads) ^ set(rgrads)
    num = den = ynum = 0.0
    worst = (0.0, "", 0.0)
    for k, g in rgrads.items():
        if k.endswith((".conv1.bias", ".conv2.bias")):
            # a conv bias in front of train-mode BatchNorm has an identically zero gradient: exact zeros here, rounding noise there
            assert float(grads[k].abs().max()) == 0.0 and float(g.abs().max()) < 1e-2 * max(float(v.abs().max()) for v in rgrads.values()), k
            continue
        e, y = rel_l2(grads[k], g), rel_l2(ygrads[k], g)
        num += float(((grads[k].double() - g.double()) ** 2).sum())
        ynum += float(((ygrads[k].double() - g.double()) ** 2).sum())
        den += float((g.double() ** 2).sum())
        if e - 1.5 * y > worst[0] - 1.5 * worst[2] or not worst[1]:
            worst = (e, k, y)
        assert e <= 1.5 * y + 3e-2, (k, e, y)                   # every parameter, against the reference's own bf16 noise on it
    e_g, y_g = (num / den) ** 0.5, (ynum / den) ** 0.5
    print(f"{tag}: gradients relL2 over all parameters {e_g:.4f} (yardstick {y_g:.4f}); tightest margin on {worst[1]}: {worst[0]:.4f} vs {worst[2]:.4f}")
    assert e_g <= 1.5 * y_g + 1e-2
    # the AdamW step: first-step updates are ~ -lr * sign(g) (+ weight decay): compared as update vectors; the bound is again the
    # reference's own -- sign flips of near-zero gradient elements under bf16 are what both see
    unum = uden = uynum = 0.0
    for k, v in rsd1.items():
        a = sd1[k]
        if not v.is_floating_point():
            assert torch.equal(a, v), k                          # num_batches_tracked
            continue
        if "running_" in k:
            assert rel_l2(a, v) <= 1.5 * rel_l2(ysd1[k], v) + 5e-3, k
            continue
        if k.endswith((".conv1.bias", ".conv2.bias")) or k in nograd_ok:
            continue
        du, dr, dy = (a - sd0[k]).double(), (v - sd0[k]).double(), (ysd1[k] - sd0[k]).double()
        assert float((a - v).abs().max()) <= 2.05 * LR, k          # no element can move further than a flipped sign
        unum += float(((du - dr) ** 2).sum())
        uynum += float(((dy - dr) ** 2).sum())
        uden += float((dr ** 2).sum())
    e_u, y_u = (unum / uden) ** 0.5, (uynum / uden) ** 0.5
    print(f"{tag}: AdamW update relL2 {e_u:.4f} (yardstick {y_u:.4f})")
    assert e_u <= 1.5 * y_u + 2e-2


def test_config2_unet_b32_bf16_graph_step_vs_oracle():
    """BASELINE configs[1]: U-Net base 64, B=32 x 6 x 256 x 256 + 4-dim metadata, bf16 -- the workload of ``python bench.py``."""
    import mau_amd as mau
    from mau_amd import functional as F_
    case = _config2_case()
    B, flags, net, sd0, batch, ref = (case[k] for k in ("B", "flags", "net", "sd0", "batch", "ref"))
    warm = R.synthetic_batch(B, seed=77)
    # the variants under test, by NAME (the slab-row counts of <64,4,8> and <64,4,4> coincide: they cannot tell the two apart):
    # level 0 runs <64,4,4> -- 32 x 16-pixel tiles, 4 waves, two workgroups per CU --, level 1 <128,4,8> (conv3x3_bf16.hip pick_variant)
    from mau_amd import _lib
    code = F_.MAU_BF16
    if B >= 8:
        assert _lib.conv3x3_variant(code, B, 256, 256, 64)[:3] == (32, 4, 64), _lib.conv3x3_variant(code, B, 256, 256, 64)
        assert _lib.conv3x3_variant(code, B, 128, 128, 128)[:3] == (32, 8, 128), _lib.conv3x3_variant(code, B, 128, 128, 128)
    yard = _oracle_step("unet", sd0, batch, flags, autocast=True)
    net = net.cuda().set_precision("bf16").train()
    hip = _hip_step(mau, net, sd0, tuple(v.cuda() for v in batch), tuple(v.cuda() for v in warm))
    nograd = [k for k in sd0 if k.startswith("model.temporal_encoder.")]
    _compare(f"config 2 (B={B})", hip, ref, yard, sd0, nograd_ok=nograd)
    # the BatchNorm statistics the B = 32 step formed -- slab sums of the convolution epilogues through the fp64 finalize -- against the
    # ORACLE's batch moments, read back from the buffers (first step from (0, 1): running_mean = 0.1 mean, running_var = 0.9 + 0.1 var);
    # one level-0 layer of each kernel (first-layer kernel, <64,4,4>) and a level-1 layer (<128,4,8>); bound: the autocast yardstick's own
    _, _, _, sd1 = hip
    for name in ("model.conv0_0.bn1", "model.conv0_0.bn2", "model.conv1_0.bn1", "model.conv0_1.bn1"):
        for buf, back in (("running_mean", lambda v: v / 0.1), ("running_var", lambda v: (v - 0.9) / 0.1)):
            k = f"{name}.{buf}"
            got, want, yd = back(sd1[k].double()), back(ref[3][k].double()), back(yard[3][k].double())
            e, y = rel_l2(got, want), rel_l2(yd, want)
            print(f"config 2 (B={B}) batch moment {k}: relL2 {e:.2e} (autocast yardstick {y:.2e})")
            assert e <= 1.5 * y + 5e-3, (k, e, y)


def test_config2_unet_b32_fp32_step_vs_oracle():
    """The north star's own tolerance at the bench's batch: the HIP **fp32** path on BASELINE configs[1] (B = 32 x 6 x 256 x 256) against
    the fp32 oracle step -- output and loss <= 1e-3 max-norm relative; every gradient by the rule of
    ``test_production_shape_fp32_vs_oracle``: within twice the reference's OWN fp32-vs-fp64 deviation, computed here on the same
    inputs (+1e-3); BatchNorm batch moments (read back from the post-step buffers) of level-0 and level-1 layers <= 1e-4."""
    import mau_amd as mau
    case = _config2_case()
    B, flags, net, sd0, batch, ref = (case[k] for k in ("B", "flags", "net", "sd0", "batch", "ref"))
    rout, rloss, rgrads, rsd1 = ref
    x, ts, md, tgt = batch
    # yardstick: the same graph in fp64 (same oracle code, double tensors)
    sd64 = {k: (v.detach().double().requires_grad_(True) if R.is_param(k) else v.detach().clone().double() if v.is_floating_point() else v.clone())
            for k, v in sd0.items()}
    out64 = R.forward("unet", sd64, x.double(), ts.double(), md.double(), True, **flags)
    R.loss_mse(out64, tgt.double())["total"].backward()
    net = mau.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, base_filters=64, **flags)      # (a network of its own: the bf16 test's holds a captured graph)
    net.load_state_dict(sd0)
    net = net.cuda().set_precision("fp32").train()
    out = net(x.cuda(), ts.cuda(), md.cuda())
    loss = mau.compute_loss_mse(out, tgt.cuda())["total"]
    loss.backward()
    torch.cuda.synchronize()
    e_out = rel_err(out.detach().cpu(), rout)
    print(f"config 2 fp32 (B={B}): output max-norm rel {e_out:.2e} (reference fp32 vs fp64: {rel_err(rout, out64.detach()):.2e}); "
          f"loss {float(loss):.8f} vs {rloss:.8f}")
    assert e_out < 1e-3
    assert abs(float(loss) - rloss) < 1e-4 * abs(rloss)
    params = dict(net.named_parameters())
    keys = [k for k in rgrads if not k.endswith((".conv1.bias", ".conv2.bias"))]
    y_max = max(rel_err(rgrads[k], sd64[k].grad) for k in keys)
    worst = (0.0, 0.0, 0.0, 0.0, "")
    close_n = close_d = 0
    for k in keys:
        g = rgrads[k]
        got = params[k].grad.cpu()
        e, e2 = rel_err(got, g), rel_l2(got, g)
        y_e, y_e2 = rel_err(g, sd64[k].grad), rel_l2(g, sd64[k].grad)
        small = float((got - g).abs().max()) < 1e-6
        assert (e <= 2 * y_max + 1e-3 and e2 <= 2 * y_e2 + 1e-3) or small, (k, e, e2, y_max, y_e2)
        if e > worst[0]:
            worst = (e, e2, y_e, y_e2, k)
        close_n += int(torch.isclose(got, g, rtol=1e-3, atol=1e-3 * float(g.abs().max())).sum())
        close_d += got.numel()
    print(f"config 2 fp32 (B={B}): worst gradient max-norm {worst[0]:.2e} / relL2 {worst[1]:.2e} on {worst[4]} (reference fp32-vs-fp64 "
          f"there: {worst[2]:.2e} / {worst[3]:.2e}); element-wise allclose(1e-3) pass fraction {close_n / max(close_d, 1):.6f}")
    sd1 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for name in ("model.conv0_0.bn1", "model.conv0_0.bn2", "model.conv1_0.bn1", "model.conv0_1.bn1", "model.conv4_0.bn2"):
        for buf, back in (("running_mean", lambda v: v / 0.1), ("running_var", lambda v: (v - 0.9) / 0.1)):
            k = f"{name}.{buf}"
            e = rel_err(back(sd1[k].double()), back(rsd1[k].double()))
            assert e <= 1e-4, (k, e)


def test_config3_unetpp_b16_bf16_graph_step_vs_oracle():
    """BASELINE configs[2]: U-Net++ (decoder-wide embedding concat, row buffers, folded embedding), B=16, 256x256, bf16."""
    import mau_amd as mau
    B = int(os.environ.get("MAU_TEST_FULL_B_UPP", "16"))
    torch.manual_seed(0)
    net = mau.UrbanPredictor("unet++", 6, 10, 64, 4, 64, 96, 2, base_filters=64)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    batch = R.synthetic_batch(B)
    warm = R.synthetic_batch(B, seed=77)
    ref = _oracle_step("unet++", sd0, batch, {}, autocast=False)
    yard = _oracle_step("unet++", sd0, batch, {}, autocast=True)
    net = net.cuda().set_precision("bf16").train()
    hip = _hip_step(mau, net, sd0, tuple(v.cuda() for v in batch), tuple(v.cuda() for v in warm))
    _compare(f"config 3 (B={B})", hip, ref, yard, sd0)


def test_six_bf16_steps_at_production_width_track_the_oracle():
    """Multi-step drift at production WIDTH (base_filters 64: the 64 / 128-wide big-tile variants, ``wgrad16_kernel``, the arena,
    ``adamw_pack_kernel`` writing the packs the next step reads): six AdamW steps of the bf16 U-Net at 4 x 6 x 128 x 128 through the
    product's step driver (eager warm-ups, capture, replays -- every step on a NEW batch) against the fp32 oracle, with the oracle
    under CPU bf16 autocast on the same batches as the yardstick.  The lr is raised to 5e-4 so that six steps MOVE the loss (1.17 ->
    1.03): a path that trained on stale packed weights, or dropped a step's update, would stay at the initial loss -- 0.15 off, against
    a band of ~0.05.  (At 3e-3 the trajectory itself is unstable: fp32 and bf16-autocast oracle part by 0.24 at the third step and
    rejoin -- nothing to hold an implementation to.)"""
    import mau_amd as mau
    B, S, lr = 4, 128, 5e-4
    flags = dict(temporal_embeddings=False, metadata_embeddings=True)
    torch.manual_seed(3)
    net = mau.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, base_filters=64, **flags)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    batches = [R.synthetic_batch(B, size=S, seed=100 + i) for i in range(6)]

    def oracle(autocast):
        sd = R.clone_state(sd0, requires_grad=True)
        opt = torch.optim.AdamW([sd[k] for k in sd if R.is_param(k)], lr=lr, weight_decay=WD)
        losses = []
        for x, ts, md, tgt in batches:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = R.forward("unet", sd, x, ts, md, True, **flags)
                out = out.float()
            else:
                out = R.forward("unet", sd, x, ts, md, True, **flags)
            loss = R.loss_mse(out, tgt)["total"]
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
        x, ts, md, _ = batches[-1]
        with torch.no_grad():
            ev = R.forward("unet", {k: v.detach() for k, v in sd.items()}, x, ts, md, False, **flags)
        return losses, ev

    ref_losses, ref_eval = oracle(False)
    yard_losses, yard_eval = oracle(True)
    assert ref_losses[5] < 0.9 * ref_losses[0], ref_losses            # the six steps move the loss: the margin everything below relies on
    net = net.cuda().set_precision("bf16").train()
    opt = mau.AdamW(net.parameters(), lr=lr, weight_decay=WD)
    step = mau.GraphedTrainStep(net, opt, mau.compute_loss_mse, warmup=2, copy_inputs=True)
    losses = [float(step(*(v.cuda() for v in b))) for b in batches]       # steps 0, 1 eager, step 2 captures, steps 3, 4, 5 replay
    assert step.graph is not None
    print("losses hip", [round(v, 5) for v in losses], "oracle fp32", [round(v, 5) for v in ref_losses], "oracle bf16 autocast", [round(v, 5) for v in yard_losses])
    for k, (a, r, y) in enumerate(zip(losses, ref_losses, yard_losses)):
        assert abs(a - r) <= 2.0 * abs(y - r) + (2e-3 if k == 0 else 3e-2) * abs(r), (k, a, r, y)
    net.eval()
    x, ts, md, _ = batches[-1]
    with torch.no_grad():
        ev = net(x.cuda(), ts.cuda(), md.cuda()).float().cpu()
    e, y = rel_l2(ev, ref_eval), rel_l2(yard_eval, ref_eval)
    print(f"eval output after 6 steps: relL2 {e:.4f} (yardstick {y:.4f})")
    assert e <= 1.5 * y + 3e-2
