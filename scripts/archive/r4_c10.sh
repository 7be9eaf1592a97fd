#!/bin/bash
# A/B of the first layer's weight-gradient kernel in the whole step (same call)
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c10; mkdir -p $O
for F in 0 1 0 1; do
  MAU_FIRST_WGRAD=$F python bench.py --no-cpu-baseline --repeats 12 > $O/bench_fw_$F.json 2> $O/bench_fw_$F.err; echo "bench FIRST_WGRAD=$F rc=$?"
  python -c "
import json
r=[json.loads(l) for l in open('$O/bench_fw_$F.json') if l.startswith('{')][-1]
print('FIRST_WGRAD=$F', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'])"
done
