#!/bin/bash
# round 6, call 6: what the round's kernel changes are worth on the captured step -- the round-5 kernels under ABI 5
# (variants/libmau_base.so: built from the tree right after the ABI change, before any kernel was touched) against the final library,
# FIVE alternations in one call (one box, its clock drifting as it will), U-Net B=32 and U-Net++ B=16
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c6; mkdir -p $O
BASE=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_base.so
for i in 1 2 3 4 5; do
  for L in base final; do
    if [ $L = base ]; then export MAU_LIB=$BASE; else unset MAU_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 12 > $O/b_${L}_$i.json 2>/dev/null; echo "unet $L $i rc=$?"
    timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 12 --model-type unet++ --batch 16 > $O/u_${L}_$i.json 2>/dev/null; echo "unet++ $L $i rc=$?"
  done
done
unset MAU_LIB
python - <<'PY' | tee gpurun_out/r6_c6/summary.txt
import json, glob, statistics
for net, tag in (("U-Net B=32", "b"), ("U-Net++ B=16", "u")):
    for lib in ("base", "final"):
        v = []
        for f in sorted(glob.glob(f"gpurun_out/r6_c6/{tag}_{lib}_*.json")):
            for l in open(f):
                if l.startswith("{"):
                    d = json.loads(l)
            v.append(d["ms_per_step"])
        print(f"{net:14s} {lib:6s} ms_per_step {v}  median {statistics.median(v):.3f}  mean {statistics.mean(v):.3f}")
PY
