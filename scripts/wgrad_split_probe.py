#!/usr/bin/env python3
"""is the weight gradient's split-K count chosen for grid fill alone, and would fewer splits on the K-heavy layers
cost less than the slab bytes they save?  Times wgrad + unpack (the pair the step pays) per layer for forced split counts
(MAU_WGRAD_SPLITS_FORCE, read per call) around the library's choice, through the C ABI, same call.  B = 32, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16

B = int(os.environ.get("B", 32))
layers = [("conv3_1.conv1", 1536, 512, 32), ("conv2_1.conv1", 768, 256, 64), ("conv4_0.conv1", 576, 1024, 16), ("conv4_0.conv2", 1024, 1024, 16),
          ("conv3_0.conv2", 512, 512, 32), ("conv2_0.conv2", 256, 256, 64), ("conv1_1.conv1", 384, 128, 128), ("conv1_0.conv2", 128, 128, 128),
          ("conv0_1.conv1", 192, 64, 256), ("conv0_0.conv2", 64, 64, 256)]
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_BF16, torch.bfloat16


def timeit(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, cin, cout, h in layers:
    N, H, W = B, h, h
    x = torch.randn(N, H, W, cin, device="cuda").to(dt)
    dy = torch.randn(N, H, W, cout, device="cuda").to(dt)
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    os.environ.pop("MAU_WGRAD_SPLITS_FORCE", None)
    s0 = lib.mau_conv3x3_wgrad_splits(code, N, H, W, cout, cin)
    tiles = (cout + 127) // 128 * ((cin + 63) // 64) if cout % 128 == 0 else (cout + 63) // 64 * ((cin + 63) // 64)
    cands = sorted({s for s in (1, 2, 3, 4, 5, 6, 8, 16, 32, 64, 128, 256, s0, max(1, s0 // 2), max(1, s0 // 4), s0 * 2) if s >= 1})
    out = []
    for s in cands:
        os.environ["MAU_WGRAD_SPLITS_FORCE"] = str(s)
        ns = lib.mau_conv3x3_wgrad_splits(code, N, H, W, cout, cin)
        if ns != s or ns * tiles < 64:
            continue
        acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, cout, cin), device="cuda")

        def pair():
            call("mau_conv3x3_wgrad", x.data_ptr(), cin, cin, None, None, 0, dy.data_ptr(), cout, cout, acc.data_ptr(), code, N, H, W, st)
            call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), ns, dw.data_ptr(), cout, cin, st)
        t = timeit(pair)
        out.append((ns, ns * tiles, t, acc.numel() * 4 / 1e6))
    os.environ.pop("MAU_WGRAD_SPLITS_FORCE", None)
    print(f"{name:14s} {cin:5d}->{cout:4d} @{h:3d}  (co,ci) tiles {tiles:3d}  library s={s0}: " +
          "  ".join(f"[s={ns}{'*' if ns == s0 else ''} wg={b} {t:6.1f}us slab {mb:5.0f}MB]" for ns, b, t, mb in out))
