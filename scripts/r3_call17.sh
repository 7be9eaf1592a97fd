#!/bin/bash
set -u
O=gpurun_out/r3_c17; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/dp2.json 2> $O/dp2.err; echo "dp2 rc=$?"; tail -1 $O/dp2.json | cut -c1-400
timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-sync-bn > $O/dp2_nosync.json 2> $O/dp2_nosync.err; echo "dp2 nosync rc=$?"; tail -1 $O/dp2_nosync.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $O/n1_eager.json 2> $O/n1_eager.err; echo "n1 eager rc=$?"; tail -1 $O/n1_eager.json | cut -c1-300
