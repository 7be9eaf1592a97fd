#!/usr/bin/env python3
"""Same-process A/B of the weight-gradient kernel's work-item order (MAU_WGRAD_XCD=0: id = split + nsplit * tile, any split count;
1: XCD-contiguous splits, split counts a multiple of the XCD count) over the U-Net's conv shapes (bf16, B=32), interleaved rounds.
Times the kernel alone and kernel + unpack (the split-K sum's cost depends on the split count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16

B = int(os.environ.get("B", 32)); S = int(os.environ.get("S", 256))
layers = []
def vgg(name, cin, cmid, cout, h): layers.extend([(name + ".conv1", cin, cmid, h), (name + ".conv2", cmid, cout, h)])
vgg("conv0_0", 6, 64, 64, S); vgg("conv1_0", 64, 128, 128, S // 2); vgg("conv2_0", 128, 256, 256, S // 4)
vgg("conv3_0", 256, 512, 512, S // 8); vgg("conv4_0", 576, 1024, 1024, S // 16)
vgg("conv3_1", 1536, 512, 512, S // 8); vgg("conv2_1", 768, 256, 256, S // 4); vgg("conv1_1", 384, 128, 128, S // 2)
vgg("conv0_1", 192, 64, 64, S)
st = torch.cuda.current_stream().cuda_stream
code, dt = MAU_BF16, torch.bfloat16
def timeit(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
only = [t for t in os.environ.get("LAYERS", "").split(",") if t]
tot = {"0": [0.0, 0.0], "1": [0.0, 0.0]}
for name, cin, cout, h in layers:
    if only and not any(t in name for t in only): continue
    N, H, W = B, h, h
    x = torch.randn(N, H, W, F_.pad8(cin), device="cuda").to(dt); x[..., cin:] = 0
    dy = torch.randn(N, H, W, F_.pad8(cout), device="cuda").to(dt)
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    res = {}
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ[os.environ.get("AB_VAR", "MAU_WGRAD_XCD")] = mode
            ns = lib.mau_conv3x3_wgrad_splits(code, N, H, W, cout, cin)
            acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, cout, cin), device="cuda")
            f_wg = lambda: call("mau_conv3x3_wgrad", x.data_ptr(), x.shape[-1], cin, None, None, 0, dy.data_ptr(), dy.shape[-1], cout, acc.data_ptr(), code, N, H, W, st)
            def f_both():
                f_wg()
                call("mau_conv3x3_unpack_wgrad", acc.data_ptr(), ns, dw.data_ptr(), cout, cin, st)
            a, b = timeit(f_wg), timeit(f_both)
            r = res.setdefault(mode, [ns, 1e9, 1e9, None])
            r[1], r[2] = min(r[1], a), min(r[2], b)
            if r[3] is None: r[3] = dw.clone()
    same = torch.equal(res["0"][3], res["1"][3]) if res["0"][0] == res["1"][0] else float((res["0"][3] - res["1"][3]).abs().max() / res["0"][3].abs().max())
    fl = 2.0 * 9 * cin * cout * N * H * W
    print(f"{name:16s} old s={res['0'][0]:3d} {res['0'][1]:7.1f} us (+unpack {res['0'][2]:7.1f}) {fl/res['0'][1]/1e6:6.0f} TF | xcd s={res['1'][0]:3d} {res['1'][1]:7.1f} us (+unpack {res['1'][2]:7.1f}) {fl/res['1'][1]/1e6:6.0f} TF | same={same}", flush=True)
    for m in ("0", "1"):
        tot[m][0] += res[m][1]; tot[m][1] += res[m][2]
print(f"TOTAL old {tot['0'][0]/1e3:.3f} ms (+unpack {tot['0'][1]/1e3:.3f}) | xcd {tot['1'][0]/1e3:.3f} ms (+unpack {tot['1'][1]/1e3:.3f})")
