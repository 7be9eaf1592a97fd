#!/bin/bash
# kernel trace of the eager train step: bash scripts/r3_trace.sh <tag> [env assignments...]   (through gpurun)
set -u
TAG=$1; shift
O=gpurun_out/r3_trace_$TAG; mkdir -p $O
export TMPDIR=/tmp
export MAU_OVERLAP_WGRAD=0      # one stream: per-kernel times and counters belong to one kernel at a time
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/trace.log 2>&1
echo "trace rc=$?"
python scripts/step_trace.py $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/step.txt; head -40 $O/step.txt; tail -1 $O/step.txt
