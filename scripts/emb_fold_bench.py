#!/usr/bin/env python3
"""Forward / data-gradient / weight-gradient time of a U-Net++ decoder node's first convolution with the 128-channel broadcast
embedding (E = 128) and with the folded form (E = roundup(N, 16) indicator channels): B=16, level-0..2 shapes, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_BF16, torch.bfloat16
def timeit(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 16
for (C0, C1, Cout, H) in [(64, 128, 64, 256), (256, 128, 64, 256), (128, 256, 128, 128), (256, 512, 256, 64)]:
    x = torch.randn(N, H, H, C0, device="cuda").to(dt); x1 = torch.randn(N, H, H, C1, device="cuda").to(dt)
    dy = torch.randn(N, H, H, Cout, device="cuda").to(dt)
    y = torch.empty(N, H, H, Cout, device="cuda", dtype=dt)
    for E in (128, 32, 16):
        Cin = C0 + C1 + E
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        emb = torch.randn(N, E, device="cuda"); ews = torch.empty(N, E, device="cuda", dtype=dt)
        wf, wd = F_.pack_conv_weights(w, code, forward=True, dgrad=True)
        bias = torch.zeros(Cout, device="cuda")
        tiles = lib.mau_conv3x3_num_pixel_tiles(code, N, H, H, Cout); cpad = (Cout + 63) // 64 * 64
        slab = torch.empty(tiles, 2 * cpad, device="cuda")
        dx = torch.empty(N, H, H, F_.pad8(Cin), device="cuda", dtype=dt)
        acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, H, Cout, Cin), device="cuda")
        tf = timeit(lambda: call("mau_conv3x3_fwd2", x.data_ptr(), C0, C0, x1.data_ptr(), C1, C1, emb.data_ptr(), ews.data_ptr(), E, wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), Cout, Cout, slab.data_ptr(), code, N, H, H, st))
        td = timeit(lambda: call("mau_conv3x3_fwd", dy.data_ptr(), Cout, Cout, None, None, 0, wd.data_ptr(), None, None, None, dx.data_ptr(), F_.pad8(Cin), Cin, None, code, N, H, H, st))
        tw = timeit(lambda: call("mau_conv3x3_wgrad2", x.data_ptr(), C0, C0, x1.data_ptr(), C1, C1, emb.data_ptr(), ews.data_ptr(), E, dy.data_ptr(), Cout, Cout, acc.data_ptr(), code, N, H, H, st))
        fl = 2.0 * 9 * Cin * Cout * N * H * H
        print(f"C0={C0:4d} C1={C1:4d} E={E:4d} Cout={Cout:4d} H={H:4d} | fwd {tf:7.1f} us {fl/tf/1e6:6.0f} TF | dgrad {td:7.1f} us {fl/td/1e6:6.0f} TF | wgrad {tw:7.1f} us {fl/tw/1e6:6.0f} TF", flush=True)
