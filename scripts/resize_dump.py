#!/usr/bin/env python3
"""Dump the upsample forward of a fixed input set to a file (argv[1]); run once with MAU_RESIZE_NO_CELL=1 and once without, then
scripts/resize_cmp.py compares the two dumps bit for bit (the source-cell kernel must equal the row kernel exactly)."""
import sys, os, torch
sys.path.insert(0, "/root/repo")
import mau_amd
from mau_amd._lib import call, MAU_BF16, MAU_F32
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(1)
outs = {}
for (N, C, h, w, H, W) in [(2, 128, 128, 128, 256, 256), (2, 256, 64, 64, 128, 128), (2, 512, 32, 32, 64, 64), (2, 1024, 16, 16, 32, 32), (2, 64, 31, 31, 62, 62), (2, 64, 31, 31, 63, 63), (2, 64, 125, 125, 250, 250), (2, 16, 7, 9, 14, 18), (2, 16, 5, 5, 5, 5)]:
    for code, dt in ((MAU_F32, torch.float32), (MAU_BF16, torch.bfloat16)):
        src = torch.randn(N, h, w, C, device="cuda").to(dt)
        dst = torch.full((N, H, W, C), float("nan"), device="cuda", dtype=dt)
        call("mau_resize_bilinear_fwd", src.data_ptr(), C, h, w, dst.data_ptr(), C, 0, code, N, H, W, C, st)
        torch.cuda.synchronize()
        outs[(N, C, h, w, H, W, str(dt))] = dst.float().cpu()
torch.save(outs, sys.argv[1])
