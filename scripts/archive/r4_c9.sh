#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c9; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "first_layer" > $O/pytest_first.log 2>&1; echo "pytest first rc=$?"; tail -5 $O/pytest_first.log
timeout -k 10 200 python scripts/first_layer_bench.py > $O/first_bench.txt 2>&1; echo "first bench rc=$?"; cat $O/first_bench.txt
