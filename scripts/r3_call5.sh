#!/bin/bash
set -u
O=gpurun_out/r3_c5; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "adamw or graphed_train or multi_step or train_cli or single_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -6 $O/pytest.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_new_$i.json 2> $O/bench_new_$i.err; echo "bench rc=$?"
  timeout -k 10 200 python bench.py --no-cpu-baseline --torch-adamw > $O/bench_tadam_$i.json 2> $O/bench_tadam_$i.err; echo "bench torch-adamw rc=$?"
done
python - <<'PY'
import json
for n in ("bench_new_1","bench_tadam_1","bench_new_2","bench_tadam_2"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c5/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("wgrad",{}).get("frac"), d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
tail -3 $O/bench_new_1.err
