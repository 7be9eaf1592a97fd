#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/rec5
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py -m gpu -q > gpurun_out/rec5/pytest_ops_full.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/rec5/pytest_ops_full.log
bash scripts/r5_records.sh
