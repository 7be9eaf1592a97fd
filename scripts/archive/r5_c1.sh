#!/bin/bash
# the backward's two chains share the chip by compute units.  MAU_WGRAD_CUS = W (the branch's weight gradient picks
# its split-K count for W workgroups), MAU_DGRAD_CUS = D (persistent grid of the data gradient beside it); same call, alternating.
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c1; mkdir -p $O
run() {  # W D order
  export MAU_WGRAD_CUS=$1 MAU_DGRAD_CUS=$2
  if [ "$3" = default ]; then unset MAU_BWD_DGRAD_FIRST; else export MAU_BWD_DGRAD_FIRST=$3; fi
  python bench.py --no-cpu-baseline --repeats 8 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('W=$1 D=$2 dgrad_first=$3', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r['final_loss']))"
}
for rep in 1 2; do
  run 0 0 default
  run 128 128 default
  run 96 160 default
  run 64 192 default
  run 128 128 1
  run 96 160 1
  run 96 0 default
  run 128 0 default
done 2>&1 | tee $O/ab.txt
