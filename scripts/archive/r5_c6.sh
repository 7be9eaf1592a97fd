#!/bin/bash
# the whole GPU suite, then the split-count model against round 4's rule on the captured step (same call, alternating)
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c6; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?"; tail -6 $O/pytest_gpu.txt
for rep in 1 2; do
  for M in 0 1; do
    for args in "" "--model-type unet++ --batch 16"; do
      MAU_WGRAD_SPLIT_MODEL=$M python bench.py --no-cpu-baseline --repeats 8 $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); w=r['roofline'].get('wgrad',{}); print('split_model=$M', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'], w.get('frac'), w.get('frac_with_unpack'), w.get('unpack_ms_per_step'), repr(r['final_loss']))"
    done
  done
done 2>&1 | tee $O/split_model_ab.txt
