#!/usr/bin/env python3
"""Exact-integer check of the big-tile convolution variants (the shapes the unit tests are too small to select), through the
C ABI: forward with BatchNorm partial sums, data gradient (no bias), inference epilogue, weight gradient; one and two tensor sources; ragged
image sizes.  MAU_CONV_M16=0 selects the 32x32x16 loop for the big tiles (default: 16x16x32 where the stage count is even)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16, MAU_F16
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
bad = 0
for code, dt in ((MAU_BF16, torch.bfloat16), (MAU_F16, torch.float16)):
    for (N, H, W, C0, C1, Cout, E) in [(32, 64, 64, 64, 0, 128, 0), (32, 64, 64, 64, 128, 128, 0), (8, 256, 256, 64, 0, 64, 0), (32, 60, 70, 64, 0, 128, 0),
                                       (8, 250, 250, 64, 128, 64, 0), (32, 32, 32, 512, 0, 512, 0), (32, 64, 64, 96, 0, 128, 0),
                                       (32, 64, 64, 64, 0, 128, 64), (32, 60, 70, 64, 128, 128, 64), (8, 256, 256, 64, 128, 64, 32), (32, 64, 64, 64, 0, 128, 16)]:
        cin = C0 + C1
        x = torch.randint(-3, 4, (N, H, W, C0), device="cuda").float()
        x1 = torch.randint(-3, 4, (N, H, W, max(C1, 8)), device="cuda").float()
        w = torch.randint(-2, 3, (Cout, cin + E, 3, 3), device="cuda").float()
        bias = torch.randint(-2, 3, (Cout,), device="cuda").float()
        emb = torch.randint(-3, 4, (N, max(E, 8)), device="cuda").float()[:, :E].contiguous() if E else None   # broadcast source (fuse_embeddings)
        emb_ws = torch.empty(N, E, device="cuda", dtype=dt) if E else None
        xin = torch.cat([x, x1[..., :C1]], -1) if C1 else x
        if E: xin = torch.cat([xin, emb.view(N, 1, 1, E).expand(N, H, W, E)], -1)
        ref = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), w, bias, padding=1).permute(0, 2, 3, 1).contiguous()
        xl, x1l = x.to(dt).contiguous(), x1.to(dt).contiguous()
        wf, wd = F_.pack_conv_weights(w, code, forward=True, dgrad=True)
        y = torch.zeros(N, H, W, F_.pad8(Cout), device="cuda", dtype=dt)
        tiles = lib.mau_conv3x3_num_pixel_tiles(code, N, H, W, Cout); cpad = (Cout + 63) // 64 * 64
        slab = torch.zeros(tiles, 2 * cpad, device="cuda")
        eargs = (emb.data_ptr(), emb_ws.data_ptr(), E) if E else (None, None, 0)
        call("mau_conv3x3_fwd2", xl.data_ptr(), xl.shape[-1], C0, x1l.data_ptr() if C1 else None, x1l.shape[-1] if C1 else 0, C1, *eargs,
             wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), y.shape[-1], Cout, slab.data_ptr(), code, N, H, W, st)
        torch.cuda.synchronize()
        ok_y = torch.equal(y[..., :Cout].float(), ref.to(dt).float())
        sums = slab.double().sum(0)
        ok_s = torch.allclose(sums[:Cout], ref.double().sum((0, 1, 2)), rtol=0, atol=0) and torch.allclose(sums[cpad:cpad + Cout], (ref.double() ** 2).sum((0, 1, 2)), rtol=1e-5)
        # data gradient: dy (N,H,W,Cout) -> dx (N,H,W,cin): conv with the rotated, transposed weights, no bias
        dy = torch.randint(-3, 4, (N, H, W, Cout), device="cuda").float()
        refd = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1).contiguous()
        dx = torch.zeros(N, H, W, F_.pad8(cin + E), device="cuda", dtype=dt)
        dyl = dy.to(dt).contiguous()
        call("mau_conv3x3_fwd2", dyl.data_ptr(), dyl.shape[-1], Cout, None, 0, 0, None, None, 0, wd.data_ptr(), None, None, None,
             dx.data_ptr(), dx.shape[-1], cin + E, None, code, N, H, W, st)
        torch.cuda.synchronize()
        ok_d = torch.equal(dx[..., :cin + E].float(), refd.to(dt).float())
        # inference epilogue: relu(scale * (conv + bias) + shift)
        sc = torch.randint(1, 3, (Cout,), device="cuda").float(); sh = torch.randint(-4, 5, (Cout,), device="cuda").float()
        y2 = torch.zeros_like(y)
        call("mau_conv3x3_fwd2", xl.data_ptr(), xl.shape[-1], C0, x1l.data_ptr() if C1 else None, x1l.shape[-1] if C1 else 0, C1, *eargs,
             wf.data_ptr(), bias.data_ptr(), sc.data_ptr(), sh.data_ptr(), y2.data_ptr(), y2.shape[-1], Cout, None, code, N, H, W, st)
        torch.cuda.synchronize()
        ok_p = torch.equal(y2[..., :Cout].float(), torch.relu(ref * sc + sh).to(dt).float())
        # weight gradient (split-K slabs + fixed-order sum): exact on integer data while the sums stay below 2^24
        xs = torch.randint(-1, 2, (N, H, W, C0), device="cuda").float(); x1s = torch.randint(-1, 2, (N, H, W, max(C1, 8)), device="cuda").float()
        dys = torch.randint(-1, 2, (N, H, W, Cout), device="cuda").float()
        xins = torch.cat([xs, x1s[..., :C1]], -1) if C1 else xs
        if E: xins = torch.cat([xins, emb.view(N, 1, 1, E).expand(N, H, W, E)], -1)
        refw = torch.nn.grad.conv2d_weight(xins.permute(0, 3, 1, 2), (Cout, cin + E, 3, 3), dys.permute(0, 3, 1, 2), padding=1)
        acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, Cout, cin + E), device="cuda")
        dw = torch.empty(Cout, cin + E, 3, 3, device="cuda")
        xsl, x1sl, dysl = xs.to(dt).contiguous(), x1s.to(dt).contiguous(), dys.to(dt).contiguous()
        call("mau_conv3x3_wgrad2", xsl.data_ptr(), xsl.shape[-1], C0, x1sl.data_ptr() if C1 else None, x1sl.shape[-1] if C1 else 0, C1, *eargs,
             dysl.data_ptr(), dysl.shape[-1], Cout, acc.data_ptr(), code, N, H, W, st)
        call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), lib.mau_conv3x3_wgrad_splits(code, N, H, W, Cout, cin + E), dw.data_ptr(), Cout, cin + E, st)
        torch.cuda.synchronize()
        ok_w = torch.equal(dw, refw) if float(refw.abs().max()) < 2 ** 24 else torch.allclose(dw, refw, rtol=1e-6)
        print(f"{str(dt)[6:]:9s} N={N:3d} {H}x{W} C0={C0} C1={C1} E={E} Cout={Cout}: fwd {ok_y} stats {ok_s} dgrad {ok_d} post {ok_p} wgrad {ok_w}", flush=True)
        bad += not (ok_y and ok_s and ok_d and ok_p and ok_w)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
