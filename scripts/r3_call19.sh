#!/bin/bash
set -u
O=gpurun_out/r3_c19; mkdir -p $O
export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
MAU_DP_GRAPH=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist > $O/dp1_graph.json 2> $O/dp1_graph.err; echo "dp1 graph rc=$?"
tail -5 $O/dp1_graph.err | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist > $O/dp1_eager.json 2> $O/dp1_eager.err; echo "dp1 eager rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/n1_graph.json 2> $O/n1_graph.err; echo "n1 graph rc=$?"
python - <<'PY'
import json
for n in ("dp1_graph","dp1_eager","n1_graph"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c19/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["final_loss"], d["config"]["launch"])
    except Exception as e: print(n,"ERR",e)
PY
