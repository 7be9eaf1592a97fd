#!/usr/bin/env python3
"""The first convolution (6 -> 64 @ 256x256, B = 32): mau_conv3x3_first_fwd against the generic path (layout kernel + 16-channel stage
of the implicit-GEMM kernel), through the C ABI, events on the launch stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16
B, S, Cin, Cout = int(os.environ.get("B", 32)), int(os.environ.get("S", 256)), 6, 64
st = torch.cuda.current_stream().cuda_stream
dt, code = torch.bfloat16, MAU_BF16
x = torch.randn(B, Cin, S, S, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.1
b = torch.zeros(Cout, device="cuda")
y = torch.empty(B, S, S, Cout, device="cuda", dtype=dt)
x8 = torch.empty(B, S, S, 8, device="cuda", dtype=dt)
rows = lib.mau_conv3x3_first_rows(B, S, S)
slab = torch.empty(rows, 128, device="cuda")
tiles = lib.mau_conv3x3_num_pixel_tiles(code, B, S, S, Cout)
slab2 = torch.empty(tiles, 128, device="cuda")
wf, _ = F_.pack_conv_weights(w, code, forward=True, dgrad=False)
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
f_new = lambda: call("mau_conv3x3_first_fwd", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), Cout, Cout, slab.data_ptr(), x8.data_ptr(), code, B, S, S, st)
f_new_nox8 = lambda: call("mau_conv3x3_first_fwd", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), Cout, Cout, slab.data_ptr(), None, code, B, S, S, st)
f_lay = lambda: call("mau_nchw_to_nhwc", x.data_ptr(), x8.data_ptr(), code, B, Cin, S, S, 8, st)
f_old = lambda: call("mau_conv3x3_fwd", x8.data_ptr(), 8, Cin, None, None, 0, wf.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), Cout, Cout, slab2.data_ptr(), code, B, S, S, st)
mb = (B * Cin * S * S * 4 + B * S * S * Cout * 2) / 1e6
for name, f, bytes_mb in (("first_fwd (+x8 by-product)", f_new, mb + B * S * S * 16 / 1e6), ("first_fwd (no by-product)", f_new_nox8, mb), ("generic: nchw_to_nhwc", f_lay, None), ("generic: conv3x3_fwd 16-channel stage", f_old, None)):
    us = timeit(f)
    print(f"{name:40s} {us:8.1f} us" + (f"   {bytes_mb / us:6.2f} TB/s of {bytes_mb:.0f} MB" if bytes_mb else ""))

# ---- the layer's weight gradient: its own kernel against the generic split-K kernel + unpack ----
dz = torch.randn(B, S, S, Cout, device="cuda").to(dt)
dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
ws = torch.empty(lib.mau_conv3x3_first_wgrad_ws_elems(B, S, S, Cout), device="cuda")
acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, B, S, S, Cout, Cin), device="cuda")
ns = lib.mau_conv3x3_wgrad_splits(code, B, S, S, Cout, Cin)
f_fw = lambda: call("mau_conv3x3_first_wgrad", x8.data_ptr(), dz.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), Cin, Cout, code, B, S, S, st)
def f_gw():
    call("mau_conv3x3_wgrad", x8.data_ptr(), 8, Cin, None, None, 0, dz.data_ptr(), Cout, Cout, acc.data_ptr(), code, B, S, S, st)
    call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), ns, dw.data_ptr(), Cout, Cin, st)
mbw = (B * S * S * Cout * 2 + B * S * S * 16) / 1e6
for name, f in (("first_wgrad (+ slab sum)", f_fw), ("generic: wgrad16 split-K + unpack", f_gw)):
    us = timeit(f)
    print(f"{name:40s} {us:8.1f} us   {mbw / us:6.2f} TB/s of {mbw:.0f} MB")
