#!/bin/bash
set -u
O=gpurun_out/r3_c16; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -x -k "row_buffer" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
export LAYERS=conv0_0.conv2,conv0_1.conv1,conv1_0.conv2
for th in 64 32 16; do
  echo "== TH_MAX=$th"; MAU_CONV_TH_MAX=$th timeout -k 10 120 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv"
done
