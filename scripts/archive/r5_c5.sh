#!/bin/bash
# the whole GPU suite on the round-5 library (split-count model, K groups, CU budget knob), then the default bench + U-Net++ + B=1 inference
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c5; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?"; tail -5 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python scripts/json_only.py < $O/bench_default.json | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline'].get('wgrad',{}).get('frac_with_unpack'), r['roofline'].get('wgrad',{}).get('unpack_ms_per_step'), r['fwd_ms_per_tile'], r.get('fwd_ms_per_tile_eager_unfrozen'))"
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 2>/dev/null | python scripts/json_only.py > $O/bench_unetpp_b16.json; python -c "
import json; r=json.load(open('$O/bench_unetpp_b16.json')); print('unet++', r['ms_per_step'], r['value'], r['roofline']['frac'])"
