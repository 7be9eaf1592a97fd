#!/bin/bash
# round 4, check 1: GPU suite + the data-parallel path's own cost on one GPU (1-rank RCCL group), direct RCCL vs ProcessGroupNCCL
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c1; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
MAU_DP_GRAPH=0 python bench.py --no-cpu-baseline --force-dist > $O/dp1_eager.json 2> $O/dp1_eager.err; echo "dp eager rc=$?"
MAU_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dist > $O/dp1_graph.json 2> $O/dp1_graph.err; echo "dp graph rc=$?"
MAU_RCCL_DIRECT=0 MAU_DP_GRAPH=0 python bench.py --no-cpu-baseline --force-dist > $O/dp1_eager_pg.json 2> $O/dp1_eager_pg.err; echo "dp eager pg rc=$?"
MAU_RCCL_DIRECT=0 MAU_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dist > $O/dp1_graph_pg.json 2> $O/dp1_graph_pg.err; echo "dp graph pg rc=$?"
python bench.py --no-cpu-baseline --no-graph > $O/bench_eager.json 2> $O/bench_eager.err; echo "eager rc=$?"
for f in bench_default dp1_eager dp1_graph dp1_eager_pg dp1_graph_pg bench_eager; do python - <<PY
import json
try:
    r=[json.loads(l) for l in open("$O/$f.json") if l.startswith("{")][-1]
    print("$f", r["ms_per_step"], r["timed_regions"], r.get("ms_per_step_with_loss_readback"), r["config"]["launch"], r["config"].get("collectives"))
except Exception as e: print("$f", "FAILED", e)
PY
done
