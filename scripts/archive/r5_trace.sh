#!/bin/bash
# kernel trace of ONE eager train step: bash scripts/r5_trace.sh <tag> [bench.py arguments] ; env assignments through the environment
set -u
TAG=$1; shift
O=gpurun_out/r5_trace_$TAG; mkdir -p $O
export TMPDIR=/tmp
export MAU_OVERLAP_WGRAD=0      # one stream: per-kernel times belong to one kernel at a time
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph --repeats 1 "$@" > $O/trace.log 2>&1
echo "trace $TAG rc=$?"
python scripts/step_trace.py $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/step.txt; tail -1 $O/step.txt
