#!/usr/bin/env python3
"""Per-layer timing of the single-launch slab reductions of a U-Net train step (B=32, 256x256): the BatchNorm statistics
finalize behind every convolution (mau_bn_stats_finalize_train) and the backward sums (mau_reduce_rows_f64_f32), through the
C ABI, events on the launch stream.  MAU_LIB selects an A/B build.  Prints a digest of the results (bit-identity across builds)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16

B = int(os.environ.get("B", 32)); S = int(os.environ.get("S", 256))
shapes = []
for lvl, c in enumerate((64, 128, 256, 512, 1024)):
    shapes.append((c, S >> lvl))
st = torch.cuda.current_stream().cuda_stream
dev = torch.device("cuda")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
dig = hashlib.sha256()
tot_f = tot_b = 0.0
print(f"{'C':>5s} {'H':>4s} | {'fwd rows':>8s} {'finalize us':>11s} | {'bwd rows':>8s} {'sums us':>8s}   (us include the launch boundary: back-to-back launches)")
for C, H in shapes:
    N = B; npix = N * H * H
    tiles = lib.mau_conv3x3_num_pixel_tiles(MAU_BF16, N, H, H, C); cpad = (C + 63) // 64 * 64
    torch.manual_seed(C)
    slab = torch.randn(tiles, 2 * cpad, device=dev); slab[:, cpad:] = slab[:, cpad:].abs() * 3 + 1
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
    rmean = torch.zeros(C, device=dev); rvar = torch.ones(C, device=dev); nbt = torch.zeros((), dtype=torch.int64, device=dev)
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    ws = torch.empty(lib.mau_bn_stats_ws_elems(tiles, C), dtype=torch.float64, device=dev)
    tk = F_._tickets(dev)
    f = lambda: call("mau_bn_stats_finalize_train", slab.data_ptr(), tiles, float(tiles), gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
                     rvar.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                     ws.data_ptr(), tk.data_ptr(), C, st)
    tf = timeit(f)
    rows = lib.mau_bn_bwd_rows(npix)
    bslab = torch.randn(rows, 2 * C, device=dev)
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev); g32 = torch.empty(2 * C, device=dev)
    bws = torch.empty(lib.mau_reduce_rows_ws_elems(rows, 2 * C), dtype=torch.float64, device=dev)
    g = lambda: call("mau_reduce_rows_f64_f32", bslab.data_ptr(), rows, 2 * C, 2 * C, sums.data_ptr(), g32.data_ptr(), bws.data_ptr(), tk.data_ptr(), 0.0, st)
    tb = timeit(g)
    torch.cuda.synchronize()
    for t in (scale, shift, mean, invstd, sums[:2 * C], g32): dig.update(t.cpu().numpy().tobytes())
    mult = 2 if C == 1024 else 4          # launches of this shape per U-Net step (encoder + decoder blocks, two BatchNorms each)
    tot_f += mult * tf; tot_b += mult * tb
    print(f"{C:5d} {H:4d} | {tiles:8d} {tf:11.2f} | {rows:8d} {tb:8.2f}")
print(f"per U-Net step (18 + 18 launches): finalize {tot_f:.0f} us, backward sums {tot_b:.0f} us   results sha256 {dig.hexdigest()[:16]}")
