#!/bin/bash
set -u
O=gpurun_out/r3_c4; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_dist_rehearsal.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?"; tail -4 $O/pytest_dist.log
bash scripts/profile.sh r3 --no-graph
python scripts/summarize_profile.py gpurun_out/prof_r3 gpurun_out/r3_profiles unet_bf16_b32_s256_c6_train > $O/summ.txt 2>&1; echo "summ rc=$?"; tail -30 $O/summ.txt
