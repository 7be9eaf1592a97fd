#!/bin/bash
# round 6, call 4: weight gradient -- an XCD-contiguous work-item order for ANY split count (the split-major (split, tile) list cut into
# one run per XCD), so the cost model may pick 5 / 10 / 42 / 85 splits.  (1) exact-integer tests  (2) per-layer wgrad, previous library
# vs this one, U-Net B=32 and the U-Net++'s shapes B=16  (3) step A/B  (4) single-tile inference with the fused input copy
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c4; mkdir -p $O
BASE=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_base.so
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "wgrad or k_groups or graphed_inference" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
X="x0_2.conv1:272:64:256,x0_4.conv1:400:64:256,x1_2.conv1:640:128:128,x1_3.conv1:768:128:128,x2_1.conv1:896:256:64,x2_2.conv1:1152:256:64"
for i in 1 2; do for L in base new; do
  if [ $L = base ]; then export MAU_LIB=$BASE; else unset MAU_LIB; fi
  echo "== $L B=32"; B=32 LAYERS=conv3_1.conv1,conv2_1.conv1,conv1_1.conv1,conv0_1.conv1,conv3_0.conv2 timeout -k 10 200 python scripts/conv_layer_bench.py 2>&1 | grep -E "^(conv|x)"
  echo "== $L B=16 (U-Net++ shapes)"; B=16 LAYERS=x0_,x1_,x2_ EXTRA=$X timeout -k 10 200 python scripts/conv_layer_bench.py 2>&1 | grep -E "^(conv|x)"
done; done | tee $O/wgrad_layers_ab.txt
for i in 1 2; do
  for L in base new; do
    if [ $L = base ]; then export MAU_LIB=$BASE; else unset MAU_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 15 > $O/b_${L}_$i.json 2>/dev/null; echo "unet $L rc=$?"
    timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 15 --model-type unet++ --batch 16 > $O/u_${L}_$i.json 2>/dev/null; echo "unet++ $L rc=$?"
  done
done
unset MAU_LIB
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 > $O/i_$i.json 2>/dev/null; echo "infer rc=$?"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_c4/[bui]_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("wgrad", {}).get("frac"), d["roofline"].get("wgrad", {}).get("frac_with_unpack"))
PY
