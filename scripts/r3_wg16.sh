#!/bin/bash
set -u
O=gpurun_out/r3_wg16; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "exact or two_tensor or g1 or g3 or big_tile" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
AB_VAR=MAU_WGRAD16 timeout -k 10 300 python scripts/wgrad_ab.py > $O/ab.txt 2>&1; cat $O/ab.txt | tail -21
