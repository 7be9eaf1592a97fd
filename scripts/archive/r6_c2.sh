#!/bin/bash
# round 6, call 2: head backward reduce on register pairs + a depth-two software pipeline (csrc/bn_fused.hip), the K-aware level-0 tile
# rule.  (1) bitwise fused-vs-unfused + golden head tests  (2) kernel times, previous library vs this one  (3) step A/B, alternating
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c2; mkdir -p $O
BASE=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_base.so
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused_bn_backward or head or g4" > $O/pytest_head.txt 2>&1; echo "head tests rc=$?"; tail -3 $O/pytest_head.txt
for i in 1 2; do
  echo "== base";  MAU_LIB=$BASE WHICH=head timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -v "^/opt"
  echo "== new";   WHICH=head timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -v "^/opt"
done | tee $O/head_ab.txt
for i in 1 2; do
  for L in base new; do
    if [ $L = base ]; then export MAU_LIB=$BASE; else unset MAU_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 15 > $O/b_${L}_$i.json 2>/dev/null; echo "unet $L rc=$?"
    timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 15 --model-type unet++ --batch 16 > $O/u_${L}_$i.json 2>/dev/null; echo "unet++ $L rc=$?"
  done
done
unset MAU_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_c2/[bu]_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
