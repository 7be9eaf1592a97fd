#!/bin/bash
# round 6, call 8: the launch order inside a block's backward (data gradient before / after the weight-gradient branch's kernel) re-checked
# with this round's kernels, both networks, three alternations
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c8; mkdir -p $O
for i in 1 2 3; do for D in 0 1; do
  echo "== unet dgrad_first=$D $i"; timeout -k 10 200 python scripts/bench_with.py functional._DGRAD_FIRST="'$D'" -- --no-cpu-baseline --repeats 12 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): d=json.loads(l); print(d["ms_per_step"], d["value"])'
  echo "== unet++ dgrad_first=$D $i"; timeout -k 10 200 python scripts/bench_with.py functional._DGRAD_FIRST="'$D'" -- --no-cpu-baseline --repeats 12 --model-type unet++ --batch 16 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): d=json.loads(l); print(d["ms_per_step"], d["value"])'
done; done | tee $O/launch_order_ab.txt
