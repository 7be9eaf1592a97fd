#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "first_layer" > $O/pytest_first.log 2>&1; echo "pytest first rc=$?"; tail -3 $O/pytest_first.log
timeout -k 10 200 python scripts/first_layer_bench.py > $O/first_bench.txt 2>&1; echo "first bench rc=$?"; cat $O/first_bench.txt
for F in 0 1 0 1; do
  MAU_CONV_FIRST=$F python bench.py --no-cpu-baseline --repeats 10 > $O/bench_first_$F.json 2> $O/bench_first_$F.err; echo "bench FIRST=$F rc=$?"
  python -c "
import json
r=[json.loads(l) for l in open('$O/bench_first_$F.json') if l.startswith('{')][-1]
print('FIRST=$F', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'], r['final_loss'])"
done
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --repeats 6 > $O/bench_upp.json 2> $O/bench_upp.err; echo "upp rc=$?"
MAU_CONV_L0=0 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --repeats 6 > $O/bench_upp_l0off.json 2> $O/bench_upp_l0off.err; echo "upp L0=0 rc=$?"
for f in bench_upp bench_upp_l0off; do python -c "
import json
r=[json.loads(l) for l in open('$O/$f.json') if l.startswith('{')][-1]
print('$f', r['ms_per_step'], r['timed_regions']['ms_per_step_min'], r['roofline']['frac'])"; done
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
