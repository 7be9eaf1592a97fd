#!/usr/bin/env python3
"""us and TB/s of the 1x1 head forward / backward at the training shape (B=32, 64 channels, 256x256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, lib, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
N, H, W, C, Co = 32, 256, 256, 64, 2
a = torch.randn(N, H, W, C, device="cuda").bfloat16()
w = torch.randn(Co, C, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
out = torch.empty(N, Co, H, W, device="cuda"); dout = torch.randn(N, Co, H, W, device="cuda")
da = torch.empty_like(a)
rows, rowlen = lib.mau_head_bwd_rows(N, H * W), lib.mau_head_bwd_rowlen(C, Co)
slab = torch.zeros(rows, rowlen, device="cuda")
def timeit(f):
    f(); f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e-3
tf = timeit(lambda: call("mau_head_fwd", a.data_ptr(), C, w.data_ptr(), b.data_ptr(), out.data_ptr(), 1, MAU_BF16, N, H * W, C, Co, st))
tb = timeit(lambda: call("mau_head_bwd", a.data_ptr(), C, w.data_ptr(), out.data_ptr(), dout.data_ptr(), da.data_ptr(), C, slab.data_ptr(), 1, MAU_BF16, N, H * W, C, Co, st))
print(f"head_fwd {tf*1e6:.1f} us {(a.numel()*2 + out.numel()*4)/tf/1e12:.2f} TB/s | head_bwd {tb*1e6:.1f} us {(a.numel()*4 + out.numel()*8)/tb/1e12:.2f} TB/s")
