#!/bin/bash
# round 4, check 13: the slab reductions with a group's loads in flight at once (bn.hip ordered_pair_sum) against the
# plain-loop form (variants/libmau_serial.so = -DMAU_REDUCE_SERIAL): same bits (final loss of 20 steps), step time A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c13; mkdir -p $O
V=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_serial.so
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bn or reduce or stats or finalize" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
for L in "" $V "" $V; do
  tag=$([ -z "$L" ] && echo new || echo serial)
  MAU_LIB=$L python bench.py --no-cpu-baseline --repeats 12 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$tag', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'], repr(r['final_loss']))"
done
