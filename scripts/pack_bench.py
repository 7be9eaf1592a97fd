#!/usr/bin/env python3
"""us and GB/s of mau_conv3x3_pack_weights (fp32 OIHW -> forward + data-gradient packs) per layer shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, lib, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tot = 0.0
for (cin, cout) in [(6, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512), (576, 1024), (1024, 1024), (1536, 512), (768, 256), (384, 128), (192, 64)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    wf = torch.empty(lib.mau_conv3x3_packed_elems(MAU_BF16, cout, cin), dtype=torch.bfloat16, device="cuda")
    wd = torch.empty(lib.mau_conv3x3_packed_elems(MAU_BF16, cin, cout), dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: call("mau_conv3x3_pack_weights", w.data_ptr(), wf.data_ptr(), wd.data_ptr(), MAU_BF16, cout, cin, st))
    b = w.numel() * 4 * 2 + (wf.numel() + wd.numel()) * 2
    tot += t
    print(f"{cin:5d}->{cout:5d}  {t*1e6:7.1f} us  {b/t/1e12:5.2f} TB/s")
print(f"total {tot*1e3:.3f} ms")
