#!/bin/bash
# same-call A/B: static priority for waves 4-7 of the 8-wave convolution workgroups
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c11; mkdir -p $O
V=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_setprio.so
for r in 1 2; do
  for L in "" $V; do
    tag=$([ -z "$L" ] && echo base || echo setprio)
    MAU_LIB=$L LAYERS=conv1_0,conv2_0,conv3_1,conv2_1,conv1_1 timeout -k 10 200 python scripts/conv_layer_bench.py 2>/dev/null | grep -E "^conv|TOTAL" > $O/layers_${tag}_$r.txt; echo "== $tag $r"; cat $O/layers_${tag}_$r.txt
  done
done
for L in "" $V "" $V; do
  tag=$([ -z "$L" ] && echo base || echo setprio)
  MAU_LIB=$L python bench.py --no-cpu-baseline --repeats 12 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$tag', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'])"
done
