#!/usr/bin/env python3
"""VERDICT r4 #3, priced before building: "BatchNorm-backward sums in the producer" for the nine conv1 -> conv2 edges of the U-Net
(the data gradient of conv2 produces da of conv1's BatchNorm; its epilogue would also write sum(da * mask), sum(da * mask * xhat)).
No kernel is built for the question.  Two launches that exist bracket it, through the C ABI, same call, alternating:
  * the data gradient as it is (plain epilogue) against THE SAME launch with the forward's statistics epilogue (sum, sum^2 from the
    accumulators into a slab): a LOWER bound on what the fused epilogue adds -- it would need the same packed sums PLUS a tile of y
    (16-byte loads in the store layout), the mask and xhat per element;
  * mau_bn_relu_bwd_reduce on the same tensors: what the fusion removes AT MOST (its read of da; the read of y moves into the conv).
Prints per edge: dgrad plain / with sums (us), the reduce pass (us), and  saving_upper = reduce - y_read_at_5.5TB/s - (sums - plain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16

B = int(os.environ.get("B", 32)); S = int(os.environ.get("S", 256))
edges = [("conv0_0", 64, S), ("conv1_0", 128, S // 2), ("conv2_0", 256, S // 4), ("conv3_0", 512, S // 8), ("conv4_0", 1024, S // 16),
         ("conv3_1", 512, S // 8), ("conv2_1", 256, S // 4), ("conv1_1", 128, S // 2), ("conv0_1", 64, S)]     # conv2 is C -> C in every block
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_BF16, torch.bfloat16


def timeit(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
print(f"{'edge':10s} {'C':>5s} {'H':>4s} | {'dgrad':>8s} {'+sums':>8s} {'delta':>7s} | {'reduce':>8s} {'y read':>7s} | saving upper bound (us)")
for name, C, h in edges:
    N, H, W = B, h, h
    dy = torch.randn(N, H, W, C, device="cuda").to(dt)
    yv = torch.randn(N, H, W, C, device="cuda").to(dt)
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    wd = F_.pack_conv_weights(w, code, forward=False, dgrad=True)[1]
    da = torch.empty(N, H, W, C, device="cuda", dtype=dt)
    tiles = lib.mau_conv3x3_num_pixel_tiles(code, N, H, W, C)
    slab = torch.empty(tiles, 2 * C, device="cuda")
    coef = [torch.rand(C, device="cuda") + 0.5 for _ in range(4)]
    rows = lib.mau_bn_bwd_rows(N * H * W)
    rslab = torch.empty(rows, 2 * C, device="cuda")
    plain = lambda: call("mau_conv3x3_fwd", dy.data_ptr(), C, C, None, None, 0, wd.data_ptr(), None, None, None, da.data_ptr(), C, C, None, code, N, H, W, st)
    sums = lambda: call("mau_conv3x3_fwd", dy.data_ptr(), C, C, None, None, 0, wd.data_ptr(), None, None, None, da.data_ptr(), C, C, slab.data_ptr(), code, N, H, W, st)
    red = lambda: call("mau_bn_relu_bwd_reduce", da.data_ptr(), C, yv.data_ptr(), C, *[c.data_ptr() for c in coef], rslab.data_ptr(), C, code, N * H * W, C, st)
    tp = ts = tr = 0.0
    for _ in range(2):                                  # alternating
        tp += timeit(plain) / 2; ts += timeit(sums) / 2; tr += timeit(red) / 2
    yread = N * H * W * C * 2 / 5.5e12 * 1e6
    save = tr - yread - (ts - tp)
    for i, v in enumerate((tp, ts, tr, save)): tot[i] += v
    print(f"{name:10s} {C:5d} {h:4d} | {tp:8.1f} {ts:8.1f} {ts - tp:7.1f} | {tr:8.1f} {yread:7.1f} | {save:8.1f}")
print(f"TOTAL dgrad {tot[0]:.0f} us, with sums {tot[1]:.0f} us (+{tot[1] - tot[0]:.0f}), reduce passes {tot[2]:.0f} us, saving upper bound {tot[3]:.0f} us per step")
