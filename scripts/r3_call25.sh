#!/bin/bash
set -u
O=gpurun_out/r3_c25; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for m in 2 0; do
MAU_OVERLAP_WGRAD=$m timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist > $O/dp1_$m.json 2> $O/dp1_$m.err; echo "dp1 m=$m rc=$?"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/n1_graph.json 2> $O/n1_graph.err; echo "n1 graph rc=$?"
python - <<'PY'
import json
for n in ("dp1_2","dp1_0","n1_graph"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c25/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["final_loss"], d["config"]["launch"])
    except Exception as e: print(n,"ERR",e)
PY
