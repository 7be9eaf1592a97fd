#!/bin/bash
# round 3, call 1: GPU tests of the launch-tail work + graph bench + wgrad order A/B   (through gpurun)
set -u
O=gpurun_out/r3_c1; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-graph > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?"
MAU_FUSED_REDUCE=0 MAU_PACK_MULTI=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-graph > $O/bench_eager_unfused.json 2> $O/bench_eager_unfused.err; echo "bench eager unfused rc=$?"
python - <<'PY'
import json
for n in ("bench_graph","bench_eager","bench_eager_unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c1/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("wgrad",{}).get("frac"))
    except Exception as e: print(n,"ERR",e)
PY
timeout -k 10 300 python scripts/wgrad_ab.py > $O/wgrad_ab.txt 2>&1; echo "wgrad_ab rc=$?"; cat $O/wgrad_ab.txt
timeout -k 10 120 python scripts/resize_bench.py > $O/resize.txt 2>&1; cat $O/resize.txt
