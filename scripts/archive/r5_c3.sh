#!/bin/bash
# (1) the new exact tests (K-group conv variants, production-shape wgrad16 incl. CU budgets, variant-by-name, 5-step tracking at width 64)
# (2) VERDICT r4 #3 priced: data gradient with / without epilogue sums against the BatchNorm-backward reduce pass
# (3) VERDICT r4 #6: single-tile inference, K groups off / on, same call, alternating
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "k_group or wgrad16_production or exact_integers or big_tile" > $O/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.txt
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "five_bf16" -s > $O/pytest_five.txt 2>&1; echo "five rc=$?"; tail -5 $O/pytest_five.txt
python scripts/dgrad_stats_probe.py 2>&1 | tee $O/dgrad_stats_probe.txt
for rep in 1 2; do
  for KG in 0 1; do
    export MAU_CONV_KG=$KG
    for args in "--batch 1 --channels 23 --meta 8 --precision fp16" "--batch 1 --precision bf16" "--batch 8 --precision bf16"; do
      python bench.py --no-cpu-baseline --infer --size 512 $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('KG=$KG', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'])"
    done
  done
done 2>&1 | tee $O/infer_ab.txt
unset MAU_CONV_KG
for KG in 0 1; do echo "== KG=$KG"; MAU_CONV_KG=$KG B=1 S=512 python scripts/conv_layer_bench.py 2>&1 | grep -v "^/opt"; done | tee $O/layers_b1.txt
