#!/usr/bin/env python3
"""A/B timing of library variants in ONE process family on ONE device (cdna guide rule 24): each variant is a build of
libmau_hip.so with different -D flags (scripts/build_variants.sh); a child process per variant and round times the
selected kernels of scripts/conv_layer_bench.py's layer list, rounds are interleaved.
    python3 scripts/variant_bench.py wgrad|fwd|dgrad var_a.so var_b.so ...   [ROUNDS=3]
"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch
import mau_amd
from mau_amd import functional as F_
from mau_amd._lib import call, lib, MAU_BF16
which = sys.argv[1]
B = int(os.environ.get("B", 32)); S = int(os.environ.get("S", 256))
layers = []
def vgg(name, cin, cmid, cout, h): layers.extend([(name + ".conv1", cin, cmid, h), (name + ".conv2", cmid, cout, h)])
vgg("conv0_0", 6, 64, 64, S); vgg("conv1_0", 64, 128, 128, S // 2); vgg("conv2_0", 128, 256, 256, S // 4)
vgg("conv3_0", 256, 512, 512, S // 8); vgg("conv4_0", 576, 1024, 1024, S // 16)
vgg("conv3_1", 1536, 512, 512, S // 8); vgg("conv2_1", 768, 256, 256, S // 4); vgg("conv1_1", 384, 128, 128, S // 2)
vgg("conv0_1", 192, 64, 64, S)
st = torch.cuda.current_stream().cuda_stream
code = MAU_BF16; dt = torch.bfloat16
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
out = {}
for name, cin, cout, h in layers:
    N, H, W = B, h, h
    x = torch.randn(N, H, W, F_.pad8(cin), device="cuda").to(dt); x[..., cin:] = 0
    dy = torch.randn(N, H, W, F_.pad8(cout), device="cuda").to(dt)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    bias = torch.zeros(cout, device="cuda")
    wf, wd = F_.pack_conv_weights(w, code, forward=True, dgrad=True)
    y = torch.empty(N, H, W, F_.pad8(cout), device="cuda", dtype=dt)
    dx = torch.empty(N, H, W, F_.pad8(cin), device="cuda", dtype=dt)
    tiles = lib.mau_conv3x3_num_pixel_tiles(code, N, H, W, cout); cpad = (cout + 63) // 64 * 64
    slab = torch.empty(tiles, 2 * cpad, device="cuda")
    acc = torch.empty(lib.mau_conv3x3_wgrad_acc_elems(code, N, H, W, cout, cin), device="cuda")
    if which == "fwd":
        f = lambda: call("mau_conv3x3_fwd", x.data_ptr(), x.shape[-1], cin, None, None, 0, wf.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), y.shape[-1], cout, slab.data_ptr(), code, N, H, W, st)
    elif which == "dgrad":
        f = lambda: call("mau_conv3x3_fwd", dy.data_ptr(), dy.shape[-1], cout, None, None, 0, wd.data_ptr(), None, None, None, dx.data_ptr(), dx.shape[-1], cin, None, code, N, H, W, st)
    else:
        f = lambda: call("mau_conv3x3_wgrad", x.data_ptr(), x.shape[-1], cin, None, None, 0, dy.data_ptr(), dy.shape[-1], cout, acc.data_ptr(), code, N, H, W, st)
    out[name] = timeit(f) * 1e6
print("RESULT " + json.dumps(out))
''' % ROOT

def main():
    which = sys.argv[1]
    libs = sys.argv[2:]
    rounds = int(os.environ.get("ROUNDS", 3))
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ, MAU_LIB=os.path.abspath(l))
            p = subprocess.run([sys.executable, "-c", CHILD, which], env=env, capture_output=True, text=True, timeout=600)
            line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
            if not line:
                print(l, "FAILED", p.stderr[-2000:]); return 1
            res[l].append(json.loads(line[0][7:]))
    names = list(res[libs[0]][0].keys())
    print(f"{'layer':16s} " + " ".join(f"{os.path.basename(l)[:14]:>14s}" for l in libs))
    tot = {l: 0.0 for l in libs}
    for n in names:
        row = []
        for l in libs:
            v = min(r[n] for r in res[l]); tot[l] += v; row.append(v)
        print(f"{n:16s} " + " ".join(f"{v:14.1f}" for v in row))
    print(f"{'TOTAL(min) us':16s} " + " ".join(f"{tot[l]:14.1f}" for l in libs))
    med = {l: sorted(sum(r.values()) for r in res[l])[len(res[l]) // 2] for l in libs}
    print(f"{'TOTAL(median)':16s} " + " ".join(f"{med[l]:14.1f}" for l in libs))
    return 0

if __name__ == "__main__":
    sys.exit(main())
