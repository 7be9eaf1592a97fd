#!/usr/bin/env python3
"""us and TB/s (of the algorithmic bytes) of the fused BatchNorm kernels (csrc/bn_fused.hip, resize_fwd_cell_kernel<BN>) against the
separate kernels they replace, at the U-Net's shapes (bf16, B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd._lib import call, lib, MAU_BF16
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_BF16, torch.bfloat16
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 32
def coefs(C):
    return [torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5]
which = os.environ.get("WHICH", "pool,head,up").split(",")
if "pool" in which:
    for C, H in ((64, 256), (128, 128), (256, 64), (512, 32)):
        W = H
        y = torch.randn(N, H, W, C, device="cuda").to(dt); dsk = torch.randn_like(y); dpl = torch.randn(N, H // 2, W // 2, C, device="cuda").to(dt)
        a = torch.empty_like(y); da = torch.empty_like(y); dy = torch.empty_like(y)
        sc, sh, mu, isd = coefs(C)
        npix = N * H * W
        rows = lib.mau_bn_bwd_rows(npix)
        slab = torch.empty(rows, 2 * C, device="cuda"); sums = torch.randn(2 * C + 1, device="cuda", dtype=torch.float64)
        pl = torch.empty(N, H // 2, W // 2, C, device="cuda", dtype=dt); idx = torch.empty(N, H // 2, W // 2, C // 8, device="cuda", dtype=torch.int16)
        t_ap = timeit(lambda: call("mau_bn_relu_apply_pool", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, pl.data_ptr(), C, None, code, N, H, W, C, st))
        t_api = timeit(lambda: call("mau_bn_relu_apply_pool", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, pl.data_ptr(), C, idx.data_ptr(), code, N, H, W, C, st))
        S = y.numel() * 2
        cp = (sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr())
        t_pb = timeit(lambda: call("mau_maxpool2x2_bwd_add", a.data_ptr(), C, dpl.data_ptr(), C, dsk.data_ptr(), C, da.data_ptr(), C, code, N, H, W, C, st))
        t_r = timeit(lambda: call("mau_bn_relu_bwd_reduce", da.data_ptr(), C, y.data_ptr(), C, *cp, slab.data_ptr(), C, code, npix, C, st))
        t_a = timeit(lambda: call("mau_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), C, *cp, sums.data_ptr(), float(npix), dy.data_ptr(), C, code, npix, C, st))
        pa = (y.data_ptr(), C, dpl.data_ptr(), C, idx.data_ptr(), dsk.data_ptr(), C)
        t_fr = timeit(lambda: call("mau_pool_bn_bwd_reduce", *pa, *cp, slab.data_ptr(), C, code, N, H, W, C, st))
        t_fa = timeit(lambda: call("mau_pool_bn_bwd_apply", *pa, *cp, sums.data_ptr(), float(npix), dy.data_ptr(), C, code, N, H, W, C, st))
        print(f"pool C={C:4d} H={H:3d} S={S/1e6:6.1f} MB | unfused poolbwd {t_pb:6.1f} reduce {t_r:6.1f} apply {t_a:6.1f} = {t_pb+t_r+t_a:6.1f} us | "
              f"fused reduce {t_fr:6.1f} ({2.25*S/t_fr/1e6:4.2f} TB/s) apply {t_fa:6.1f} ({3.25*S/t_fa/1e6:4.2f} TB/s) = {t_fr+t_fa:6.1f} us | fwd apply_pool {t_ap:6.1f} -> with argidx {t_api:6.1f}", flush=True)
if "head" in which:
    C, H, Co = 64, 256, 2
    W = H; HW = H * W; npix = N * HW
    y = torch.randn(N, H, W, C, device="cuda").to(dt); a = torch.empty_like(y); da = torch.empty_like(y); dy = torch.empty_like(y)
    sc, sh, mu, isd = coefs(C)
    w = torch.randn(Co, C, device="cuda") * 0.1; b = torch.zeros(Co, device="cuda")
    out = torch.empty(N, Co, H, W, device="cuda"); dout = torch.randn(N, Co, H, W, device="cuda")
    rows = lib.mau_bn_bwd_rows(npix); slab = torch.empty(rows, 2 * C, device="cuda"); sums = torch.randn(2 * C + 1, device="cuda", dtype=torch.float64)
    hrows, rowlen = lib.mau_head_bwd_rows(N, HW), lib.mau_head_bwd_rowlen(C, Co); hslab = torch.empty(hrows, rowlen, device="cuda")
    cp = (sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr())
    S = y.numel() * 2
    t1 = timeit(lambda: call("mau_bn_relu_apply", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, code, npix, C, st))
    t2 = timeit(lambda: call("mau_head_fwd", a.data_ptr(), C, w.data_ptr(), b.data_ptr(), out.data_ptr(), 1, code, N, HW, C, Co, st))
    t3 = timeit(lambda: call("mau_head_bwd", a.data_ptr(), C, w.data_ptr(), out.data_ptr(), dout.data_ptr(), da.data_ptr(), C, hslab.data_ptr(), 1, code, N, HW, C, Co, st))
    t4 = timeit(lambda: call("mau_bn_relu_bwd_reduce", da.data_ptr(), C, y.data_ptr(), C, *cp, slab.data_ptr(), C, code, npix, C, st))
    t5 = timeit(lambda: call("mau_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), C, *cp, sums.data_ptr(), float(npix), dy.data_ptr(), C, code, npix, C, st))
    f1 = timeit(lambda: call("mau_head_bn_fwd", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 1, code, N, HW, C, Co, st))
    f2 = timeit(lambda: call("mau_head_bn_bwd_reduce", y.data_ptr(), C, *cp, w.data_ptr(), out.data_ptr(), dout.data_ptr(), slab.data_ptr(), C, hslab.data_ptr(), 1, code, N, HW, C, Co, st))
    f3 = timeit(lambda: call("mau_head_bn_bwd_apply", y.data_ptr(), C, *cp, sums.data_ptr(), float(npix), w.data_ptr(), out.data_ptr(), dout.data_ptr(), dy.data_ptr(), C, 1, code, N, HW, C, Co, st))
    print(f"head S={S/1e6:6.1f} MB | unfused apply {t1:6.1f} head_fwd {t2:6.1f} head_bwd {t3:6.1f} reduce {t4:6.1f} apply {t5:6.1f} = {t1+t2+t3+t4+t5:6.1f} us | "
          f"fused fwd {f1:6.1f} ({S/f1/1e6:4.2f} TB/s) reduce {f2:6.1f} ({S/f2/1e6:4.2f}) apply {f3:6.1f} ({2*S/f3/1e6:4.2f}) = {f1+f2+f3:6.1f} us", flush=True)
if "up" in which:
    for C, h in ((1024, 16), (512, 32), (256, 64), (128, 128)):
        H = 2 * h
        y = torch.randn(N, h, h, C, device="cuda").to(dt); a = torch.empty_like(y); up = torch.empty(N, H, H, C, device="cuda", dtype=dt)
        sc, sh, mu, isd = coefs(C)
        t1 = timeit(lambda: call("mau_bn_relu_apply", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, code, N * h * h, C, st))
        t2 = timeit(lambda: call("mau_resize_bilinear_fwd", a.data_ptr(), C, h, h, up.data_ptr(), C, 0, code, N, H, H, C, st))
        f = timeit(lambda: call("mau_resize_bilinear_bn_fwd", y.data_ptr(), C, h, h, sc.data_ptr(), sh.data_ptr(), up.data_ptr(), C, 0, code, N, H, H, C, st))
        S = y.numel() * 2
        print(f"up C={C:4d} {h:3d}->{H:3d} S={S/1e6:6.1f} MB | unfused apply {t1:6.1f} resize {t2:6.1f} = {t1+t2:6.1f} us | fused {f:6.1f} us ({5*S/f/1e6:4.2f} TB/s)", flush=True)
