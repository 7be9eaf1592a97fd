#!/bin/bash
# round 6, call 3: the whole GPU suite on the pruned library (ABI 5, switches removed), then a default bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c3; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
