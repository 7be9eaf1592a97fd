#!/usr/bin/env python3
"""GB/s of the streaming BatchNorm kernels at the U-Net's activation shapes (B=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, lib, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for (C, H) in [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]:
    N = 32; npix = N * H * H
    y = torch.randn(npix, C, device="cuda").bfloat16(); da = torch.randn_like(y); a = torch.empty_like(y); dy = torch.empty_like(y)
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda"); mu = torch.randn(C, device="cuda"); isd = torch.rand(C, device="cuda") + 0.5
    rows = lib.mau_bn_bwd_rows(npix); slab = torch.empty(rows, 2 * C, device="cuda"); sums = torch.randn(2 * C, device="cuda", dtype=torch.float64)
    t1 = timeit(lambda: call("mau_bn_relu_apply", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, MAU_BF16, npix, C, st))
    t2 = timeit(lambda: call("mau_bn_relu_bwd_reduce", da.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), slab.data_ptr(), 2 * C // 2, MAU_BF16, npix, C, st))
    t3 = timeit(lambda: call("mau_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), sums.data_ptr(), float(npix), dy.data_ptr(), C, MAU_BF16, npix, C, st))
    b = npix * C * 2
    print(f"C={C:5d} H={H:4d} {b/1e6:7.1f} MB | apply {t1*1e6:7.1f} us {2*b/t1/1e12:5.2f} TB/s | bwd_reduce {t2*1e6:7.1f} us {2*b/t2/1e12:5.2f} TB/s | bwd_apply {t3*1e6:7.1f} us {3*b/t3/1e12:5.2f} TB/s")
