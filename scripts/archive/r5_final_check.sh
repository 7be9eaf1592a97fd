#!/bin/bash
# HEAD, as the driver will run it: build check, smoke, whole GPU suite, default bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?"; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python scripts/json_only.py < $O/bench_default.json | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['traffic'] is not None, r['cpu_baseline']['value'], 'b32_recorded' in r['cpu_baseline'])"
