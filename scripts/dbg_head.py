import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import mau_amd
from mau_amd._lib import call, lib, MAU_BF16, MAU_F16
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_F16, torch.float16
torch.manual_seed(0)
N, C, H, Co = 2, 16, 64, 2
W = 48; HW = H * W; npix = N * HW
y = torch.randn(N, H, W, C, device="cuda").to(dt); a = torch.empty_like(y); da = torch.empty_like(y); dy1 = torch.empty_like(y); dy2 = torch.empty_like(y)
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1; mu = torch.randn(C, device="cuda") * 0.1; isd = torch.rand(C, device="cuda") + 0.5
w = torch.randn(Co, C, device="cuda") * 0.1; b = torch.zeros(Co, device="cuda")
out = torch.empty(N, Co, H, W, device="cuda"); dout = torch.randn(N, Co, H, W, device="cuda") * float(os.environ.get("DSCALE", "1e-5"))
rows = lib.mau_bn_bwd_rows(npix); slab = torch.empty(rows, 2 * C, device="cuda"); sums = (torch.randn(2 * C + 1, device="cuda", dtype=torch.float64))
hrows, rowlen = lib.mau_head_bwd_rows(N, HW), lib.mau_head_bwd_rowlen(C, Co); hslab = torch.empty(hrows, rowlen, device="cuda")
cp = (sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr())
call("mau_bn_relu_apply", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), C, code, npix, C, st)
call("mau_head_fwd", a.data_ptr(), C, w.data_ptr(), b.data_ptr(), out.data_ptr(), 1, code, N, HW, C, Co, st)
call("mau_head_bwd", a.data_ptr(), C, w.data_ptr(), out.data_ptr(), dout.data_ptr(), da.data_ptr(), C, hslab.data_ptr(), 1, code, N, HW, C, Co, st)
call("mau_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), C, *cp, sums.data_ptr(), float(npix), dy1.data_ptr(), C, code, npix, C, st)
call("mau_head_bn_bwd_apply", y.data_ptr(), C, *cp, sums.data_ptr(), float(npix), w.data_ptr(), out.data_ptr(), dout.data_ptr(), dy2.data_ptr(), C, 1, code, N, HW, C, Co, st)
torch.cuda.synchronize()
bad = (dy1 != dy2).nonzero()
print("mismatches", len(bad), "of", dy1.numel())
for i in bad[:6]:
    n, yy, xx, c = [int(v) for v in i]
    print((n, yy, xx, c), float(dy1[n, yy, xx, c]), float(dy2[n, yy, xx, c]), "y", float(y[n, yy, xx, c]), "da", float(da[n, yy, xx, c]),
          "dout", [float(dout[n, o, yy, xx]) for o in range(Co)], "out0", float(out[n, 0, yy, xx]), "w", [float(w[o, c]) for o in range(Co)])
import math
def f32(v): return np.float32(v)
def fma(a, b, c): return np.float32(np.float64(a) * np.float64(b) + np.float64(c))   # exact product + one rounding (double holds it)
for i in bad[:6]:
    n, yy, xx, c = [int(v) for v in i]
    scc, shc, muc, isc = (f32(t[c].item()) for t in (sc, sh, mu, isd))
    m1 = f32(sums[c].item() * (1.0 / npix)); m2 = f32(sums[C + c].item() * (1.0 / npix))
    k1 = f32(f32(scc * m2) * isc); k0 = fma(scc, m1, -f32(k1 * muc))
    yv = f32(y[n, yy, xx, c].item())
    t = f32(out[n, 0, yy, xx].item()); f0 = fma(-t, t, f32(1))
    dz0 = f32(f32(dout[n, 0, yy, xx].item()) * f0); dz1 = f32(dout[n, 1, yy, xx].item())
    s = fma(dz1, f32(w[1, c].item()), fma(dz0, f32(w[0, c].item()), f32(0)))
    units = float(s) / 2.0 ** -24
    for name, dav in (("rounded RNE", f32(round(units) * 2.0 ** -24)), ("unrounded", s), ("trunc", f32(math.floor(units) * 2.0 ** -24)), ("ceil", f32(math.ceil(units) * 2.0 ** -24)), ("zero", f32(0))):
        act = fma(yv, scc, shc)
        dzb = dav if act > 0 else f32(0)
        dyv = fma(scc, dzb, -fma(k1, yv, k0))
        print("   ", name, "s units", units, "da", float(dav), "dy f32", float(dyv), "-> f16", float(np.float16(dyv)))
