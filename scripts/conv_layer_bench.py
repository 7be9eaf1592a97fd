#!/usr/bin/env python3
"""Per-layer timing of the three convolution kernels over the U-Net's conv shapes (B=32, 256x256),
through the C ABI, with events on the launch stream.  Prints TFLOP/s and the HBM floor per layer."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16

B = int(os.environ.get("B", 32)); S = int(os.environ.get("S", 256))
layers = []  # name, Cin, Cout, H
def vgg(name, cin, cmid, cout, h): layers.extend([(name + ".conv1", cin, cmid, h), (name + ".conv2", cmid, cout, h)])
vgg("conv0_0", 6, 64, 64, S); vgg("conv1_0", 64, 128, 128, S // 2); vgg("conv2_0", 128, 256, 256, S // 4)
vgg("conv3_0", 256, 512, 512, S // 8); vgg("conv4_0", 576, 1024, 1024, S // 16)
vgg("conv3_1", 1536, 512, 512, S // 8); vgg("conv2_1", 768, 256, 256, S // 4); vgg("conv1_1", 384, 128, 128, S // 2)
vgg("conv0_1", 192, 64, 64, S)
# EXTRA="name:cin:cout:h,...": further layers (e.g. the U-Net++'s full-resolution nodes x0_1..x0_4: 208 / 272 / 336 / 400 -> 64)
for t in [t for t in os.environ.get("EXTRA", "").split(",") if t]:
    nm, ci, co, hh = t.split(":")
    layers.append((nm, int(ci), int(co), int(hh)))
st = torch.cuda.current_stream().cuda_stream
code = MAU_BF16; dt = torch.bfloat16
def timeit(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}; totf = 0.0
records = []
print(f"{'layer':16s} {'Cin':>5s} {'Cout':>5s} {'H':>4s} {'GFLOP':>8s} | {'fwd us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s} | {'HBM floor us':>11s}")
_only = [t for t in os.environ.get("LAYERS", "").split(",") if t]
for name, cin, cout, h in layers:
    if _only and not any(t in name for t in _only): continue
    N, H, W = B, h, h
    x = torch.randn(N, H, W, F_.pad8(cin), device="cuda").to(dt); x[..., cin:] = 0
    dy = torch.randn(N, H, W, F_.pad8(cout), device="cuda").to(dt)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bias = torch.zeros(cout, device="cuda")
    wf, wd = F_.pack_conv_weights(w, code, forward=True, dgrad=True)
    y = torch.empty(N, H, W, F_.pad8(cout), device="cuda", dtype=dt)
    dx = torch.empty(N, H, W, F_.pad8(cin), device="cuda", dtype=dt)
    tiles = lib.mau_conv3x3_num_pixel_tiles(code, N, H, W, cout); cpad = (cout + 63) // 64 * 64
    slab = torch.empty(tiles, 2 * cpad, device="cuda")
    acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, cout, cin), device="cuda")
    ns = lib.mau_conv3x3_wgrad_splits(code, N, H, W, cout, cin)
    dw = torch.empty_like(w)
    # EPI=post: the forward column times the INFERENCE epilogue (eval-mode BatchNorm + ReLU folded in, no statistics slab)
    post = os.environ.get("EPI") == "post"
    psc, psh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    f_fwd = lambda: call("mau_conv3x3_fwd", x.data_ptr(), x.shape[-1], cin, None, None, 0, wf.data_ptr(), bias.data_ptr(), psc.data_ptr() if post else None,
                         psh.data_ptr() if post else None, y.data_ptr(), y.shape[-1], cout, None if post else slab.data_ptr(), code, N, H, W, st)
    f_dg = lambda: call("mau_conv3x3_fwd", dy.data_ptr(), dy.shape[-1], cout, None, None, 0, wd.data_ptr(), None, None, None, dx.data_ptr(), dx.shape[-1], cin, None, code, N, H, W, st)
    f_wg = lambda: call("mau_conv3x3_wgrad", x.data_ptr(), x.shape[-1], cin, None, None, 0, dy.data_ptr(), dy.shape[-1], cout, acc.data_ptr(), code, N, H, W, st)
    fl = 2.0 * 9 * cin * cout * N * H * W
    tf, td, tw = timeit(f_fwd), timeit(f_dg), timeit(f_wg)
    floor = (F_.pad8(cin) + F_.pad8(cout)) * N * H * W * 2 / 5.5e12
    print(f"{name:16s} {cin:5d} {cout:5d} {h:4d} {fl/1e9:8.1f} | {tf*1e6:8.1f} {fl/tf/1e12:6.0f} | {td*1e6:8.1f} {fl/td/1e12:6.0f} | {tw*1e6:8.1f} {fl/tw/1e12:6.0f} (s={ns:3d}) | {floor*1e6:11.1f}")
    tot["fwd"] += tf; tot["dgrad"] += td if name != "conv0_0.conv1" else 0; tot["wgrad"] += tw; totf += fl
    minb = (F_.pad8(cin) + F_.pad8(cout)) * N * H * W * 2 + 9 * cin * cout * 2
    records.append({"layer": name, "Cin": cin, "Cout": cout, "H": h, "B": B, "gflop": fl / 1e9, "min_hbm_bytes": minb,
                    "variant": ("BN128" if ((cout + 63) // 64 * 64) % 128 == 0 else "BN64") + f"/tiles{tiles}",
                    **{k: {"us": t * 1e6, "tflops": fl / t / 1e12, "mfma_frac": fl / t / 2.5e15, "tbps_min_bytes": minb / t / 1e12, "hbm_frac": minb / t / 8e12}
                       for k, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw))},
                    "wgrad_splits": ns})
print(f"TOTAL fwd {tot['fwd']*1e3:.2f} ms ({totf/tot['fwd']/1e12:.0f} TF/s)  dgrad {tot['dgrad']*1e3:.2f} ms  wgrad {tot['wgrad']*1e3:.2f} ms ({totf/tot['wgrad']/1e12:.0f} TF/s)")

if os.environ.get("OUT"):
    import json
    d00 = [r for r in records if r["layer"].startswith("conv0_0")]
    json.dump({"what": "per-layer timing of the three convolution kernels through the C ABI, events on the launch stream, 8 launches each after 2 warm-ups "
                       "(scripts/conv_layer_bench.py); bf16, B=%d, %dx%d input" % (B, S, S),
               "peaks": {"bf16_mfma_tflops": 2500.0, "hbm_tbps": 8.0, "hbm_achievable_tbps": 6.3},
               "north_star_layer_conv0_0_DoubleConv": {"fwd_us": sum(r["fwd"]["us"] for r in d00), "gflop": sum(r["gflop"] for r in d00),
                                                      "min_hbm_bytes": sum(r["min_hbm_bytes"] for r in d00),
                                                      "fwd_tflops": sum(r["gflop"] for r in d00) / sum(r["fwd"]["us"] for r in d00) * 1e3,
                                                      "fwd_tbps_min_bytes": sum(r["min_hbm_bytes"] for r in d00) / sum(r["fwd"]["us"] for r in d00) * 1e-6,
                                                      "fwd_hbm_frac": sum(r["min_hbm_bytes"] for r in d00) / sum(r["fwd"]["us"] for r in d00) * 1e-6 / 8.0,
                                                      "fwd_hbm_frac_of_achievable": sum(r["min_hbm_bytes"] for r in d00) / sum(r["fwd"]["us"] for r in d00) * 1e-6 / 6.3},
               "totals_ms": {k: v * 1e3 for k, v in tot.items()}, "layers": records}, open(os.environ["OUT"], "w"), indent=1)
