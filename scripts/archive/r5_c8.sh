#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/rec5
timeout -k 10 600 python -m pytest tests/test_gpu_properties_full_size.py -m gpu -q > gpurun_out/rec5/pytest_properties.log 2>&1; echo "properties rc=$?"; tail -4 gpurun_out/rec5/pytest_properties.log
bash scripts/r5_records2.sh
