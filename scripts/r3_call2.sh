#!/bin/bash
# round 3, call 2: fused BatchNorm variants -- GPU tests + same-box A/B of the train step   (through gpurun)
set -u
O=gpurun_out/r3_c2; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_fused_$i.json 2> $O/bench_fused_$i.err; echo "bench fused rc=$?"
  MAU_FUSED_UP=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_noup_$i.json 2> $O/bench_noup_$i.err; echo "bench noup rc=$?"
  MAU_FUSED_BN=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_unfused_$i.json 2> $O/bench_unfused_$i.err; echo "bench unfused rc=$?"
done
python - <<'PY'
import json
for n in ("bench_fused_1","bench_noup_1","bench_unfused_1","bench_fused_2","bench_noup_2","bench_unfused_2"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c2/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("wgrad",{}).get("frac"), d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
