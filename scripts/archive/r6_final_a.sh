#!/bin/bash
# round 6, final call A: the whole GPU suite on the final library, then part `a` of the round's records (scripts/records.sh)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/rec_r6
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/rec_r6/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rec_r6/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/rec_r6/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/rec_r6/smoke.txt
bash scripts/records.sh r6 a
