#!/bin/bash
# round 4, check 14: slab reductions -- tiered groups (new) vs the plain loop of the rounds before (variants/libmau_oldbn.so = bn.hip of
# commit 44c0403) vs -DMAU_REDUCE_SERIAL: per-layer kernel timing, bit-identity digest, step A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c14; mkdir -p $O
VO=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_oldbn.so
VS=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_serial.so
for L in "" $VO $VS; do
  tag=$([ -z "$L" ] && echo new || basename $L .so)
  MAU_LIB=$L timeout -k 10 120 python scripts/reduce_bench.py > $O/reduce_$tag.txt 2>&1; echo "== $tag rc=$?"; grep -v amdgpu.ids $O/reduce_$tag.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bn or reduce or stats or finalize" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.txt
for L in "" $VO "" $VO "" $VO; do
  tag=$([ -z "$L" ] && echo new || echo old)
  MAU_LIB=$L python bench.py --no-cpu-baseline --repeats 12 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$tag', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'], repr(r['final_loss']))"
done
