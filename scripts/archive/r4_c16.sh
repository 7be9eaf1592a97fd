#!/bin/bash
# round 4, check 16: under-filled 128-multiple layers on 64-channel workgroups + plain item order below 8 pixel tiles
# (single-tile inference): per-layer A/B at B=1 512x512, the app-shape inference line, then the GPU suite
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c16; mkdir -p $O
for NAR in 0 1 0 1; do
  MAU_CONV_NARROW=$NAR B=1 S=512 timeout -k 10 120 python scripts/conv_layer_bench.py 2>/dev/null | grep -E "^conv|TOTAL" | cut -c1-75 > $O/layers_b1_nar$NAR.txt; echo "== B=1 512 narrow=$NAR"; cat $O/layers_b1_nar$NAR.txt
done
for NAR in 0 1 0 1; do
  for args in "--batch 1 --channels 23 --meta 8 --precision fp16" "--batch 1 --precision bf16" "--batch 8 --precision bf16"; do
    MAU_CONV_NARROW=$NAR python bench.py --no-cpu-baseline --infer --size 512 $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('narrow=$NAR', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'])"
  done
done
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
