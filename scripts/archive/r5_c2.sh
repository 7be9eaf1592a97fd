#!/bin/bash
# VERDICT r4 #1, second call: the chip is shared by compute units only in the layers whose main chain carries long HBM-bound
# passes behind the data gradient (256^2 / 128^2 images); everywhere else the two chains take turns as before.
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c2; mkdir -p $O
run() {  # W D minHW
  export MAU_WGRAD_CUS=$1 MAU_DGRAD_CUS=$2 MAU_SHARE_MIN_HW=$3
  python bench.py --no-cpu-baseline --repeats 8 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('W=$1 D=$2 minHW=$3', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r['final_loss']))"
}
for rep in 1 2; do
  run 0 0 0
  run 96 160 65536
  run 128 128 65536
  run 96 160 16384
  run 128 128 16384
  run 64 192 16384
  run 160 96 16384
done 2>&1 | tee $O/ab.txt
