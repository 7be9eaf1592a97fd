#!/usr/bin/env python3
"""Per-layer HBM traffic of the three convolution kernels from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of
scripts/conv_layer_bench.py:   python scripts/layer_traffic.py <dir with pmc_fetch/ and pmc_write/> > table
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 correction, MI355X_MICROARCH.md HBM section); algorithmic bytes =
(Cin + Cout) * N*H*W * 2 + 9*Cin*Cout*2 (forward / data gradient) or ... + 9*Cin*Cout*4 (weight gradient: fp32 result)."""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
B, S = 32, 256
layers = []
def vgg(name, cin, cmid, cout, h): layers.extend([(name + ".conv1", cin, cmid, h), (name + ".conv2", cmid, cout, h)])
vgg("conv0_0", 6, 64, 64, S); vgg("conv1_0", 64, 128, 128, S // 2); vgg("conv2_0", 128, 256, 256, S // 4)
vgg("conv3_0", 256, 512, 512, S // 8); vgg("conv4_0", 576, 1024, 1024, S // 16)
vgg("conv3_1", 1536, 512, 512, S // 8); vgg("conv2_1", 768, 256, 256, S // 4); vgg("conv1_1", 384, 128, 128, S // 2)
vgg("conv0_1", 192, 64, 64, S)


def runs(counter):
    f = sorted(glob.glob(os.path.join(src, "pmc_" + counter.lower().split("_")[0], "*", "*counter_collection.csv")))[-1]
    byd = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and ("conv3x3_bf16_kernel" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"]) and "unpack" not in r["Kernel_Name"]:
            byd[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    out, cur = [], None
    for d in sorted(byd):
        n, v = byd[d]
        if cur is None or cur[0] != n or len(cur[1]) == 10:
            cur = [n, []]
            out.append(cur)
        cur[1].append(v)
    return [(n, sum(v) / len(v)) for n, v in out if len(v) == 10]


fe, wr = runs("FETCH_SIZE"), runs("WRITE_SIZE")
assert len(fe) == len(wr) == 3 * len(layers), (len(fe), len(wr), len(layers))
rec = []
print(f"{'layer':16s} {'pass':6s} {'alg MB':>8s} {'HBM MB':>8s} {'ratio':>6s}   kernel")
tot = collections.defaultdict(lambda: [0.0, 0.0])
for i, (name, cin, cout, h) in enumerate(layers):
    for j, ps in enumerate(("fwd", "dgrad", "wgrad")):
        k = 3 * i + j
        hbm = (2 * fe[k][1] + wr[k][1]) * 1024
        pad8 = lambda c: (c + 7) // 8 * 8
        alg = (pad8(cin) + pad8(cout)) * B * h * h * 2 + 9 * cin * cout * (4 if ps == "wgrad" else 2)
        kern = fe[k][0].split("(")[0][-60:]
        print(f"{name:16s} {ps:6s} {alg / 1e6:8.1f} {hbm / 1e6:8.1f} {hbm / alg:6.2f}   {kern}")
        rec.append({"layer": name, "pass": ps, "algorithmic_bytes": alg, "hbm_bytes_pmc": hbm, "ratio": hbm / alg})
        if not (ps == "dgrad" and name == "conv0_0.conv1"):
            tot[ps][0] += alg
            tot[ps][1] += hbm
for ps, (a, h) in tot.items():
    print(f"TOTAL {ps:6s}: algorithmic {a / 1e9:.2f} GB, measured {h / 1e9:.2f} GB, ratio {h / a:.2f}")
if len(sys.argv) > 2:
    json.dump({"correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024", "layers": rec, "totals": {k: {"algorithmic": a, "hbm": h, "ratio": h / a} for k, (a, h) in tot.items()}}, open(sys.argv[2], "w"), indent=1)
