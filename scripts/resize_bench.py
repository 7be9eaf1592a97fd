#!/usr/bin/env python3
"""us and GB/s of the bilinear resize kernels (forward into a concat slice, backward from it) at the U-Net decoder shapes (B=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
N = 32
for (C, h, Cskip) in [(128, 128, 64), (256, 64, 128), (512, 32, 256), (1024, 16, 512)]:
    H = 2 * h
    src = torch.randn(N, h, h, C, device="cuda").bfloat16()
    ld = Cskip + C
    cat = torch.zeros(N, H, H, ld, device="cuda", dtype=torch.bfloat16)
    dsrc = torch.empty_like(src)
    tf = timeit(lambda: call("mau_resize_bilinear_fwd", src.data_ptr(), C, h, h, cat.data_ptr(), ld, Cskip, MAU_BF16, N, H, H, C, st))
    tb = timeit(lambda: call("mau_resize_bilinear_bwd", cat.data_ptr(), ld, Cskip, H, H, dsrc.data_ptr(), C, MAU_BF16, N, h, h, C, st))
    bytes_ = (N * h * h * C + N * H * H * C) * 2
    # the layout the models use since round 2: the upsampled tensor is a tensor of its own (virtual concat), ld = C, offset 0
    own = torch.randn(N, H, H, C, device="cuda").bfloat16()
    tf2 = timeit(lambda: call("mau_resize_bilinear_fwd", src.data_ptr(), C, h, h, own.data_ptr(), C, 0, MAU_BF16, N, H, H, C, st))
    tb2 = timeit(lambda: call("mau_resize_bilinear_bwd", own.data_ptr(), C, 0, H, H, dsrc.data_ptr(), C, MAU_BF16, N, h, h, C, st))
    print(f"C={C:5d} {h:4d}->{H:4d} {bytes_/1e6:7.1f} MB | into a concat slice: fwd {tf*1e6:7.1f} us {bytes_/tf/1e12:5.2f} TB/s  bwd {tb*1e6:7.1f} us {bytes_/tb/1e12:5.2f} TB/s"
          f" | own tensor: fwd {tf2*1e6:7.1f} us {bytes_/tf2/1e12:5.2f} TB/s  bwd {tb2*1e6:7.1f} us {bytes_/tb2/1e12:5.2f} TB/s")
