#!/bin/bash
# full GPU suite, then the round's records (scripts/r5_records.sh) on the same box
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/rec5
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/rec5/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/rec5/pytest_gpu.log
bash scripts/r5_records.sh
