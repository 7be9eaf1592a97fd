#!/bin/bash
# round 4, check 3: the level-0 variant <64,4,4> (two independent 4-wave workgroups per CU): exactness, per-layer A/B, step A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c3; mkdir -p $O
MAU_CONV_L0=1 timeout -k 10 300 python scripts/conv_big_tile_check.py > $O/exact_l0.txt 2>&1; echo "exact rc=$?"; tail -3 $O/exact_l0.txt
for L0 in 0 1 0 1; do
  MAU_CONV_L0=$L0 LAYERS=conv0_0,conv0_1 timeout -k 10 200 python scripts/conv_layer_bench.py > $O/layers_l0_$L0.txt 2>&1; echo "layers L0=$L0 rc=$?"; cat $O/layers_l0_$L0.txt
done
for L0 in 0 1; do
  MAU_CONV_L0=$L0 python bench.py --no-cpu-baseline --repeats 12 > $O/bench_l0_$L0.json 2> $O/bench_l0_$L0.err; echo "bench L0=$L0 rc=$?"
  python -c "
import json
r=[json.loads(l) for l in open('$O/bench_l0_$L0.json') if l.startswith('{')][-1]
print('L0=$L0', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'], r['final_loss'])"
done
