#!/bin/bash
set -u
O=gpurun_out/r3_c3; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_new_$i.json 2> $O/bench_new_$i.err; echo "bench rc=$?"
  MAU_WGRAD16=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_wg32_$i.json 2> $O/bench_wg32_$i.err; echo "bench wg32 rc=$?"
done
python - <<'PY'
import json
for n in ("bench_new_1","bench_wg32_1","bench_new_2","bench_wg32_2"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c3/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("wgrad",{}).get("frac"), d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
