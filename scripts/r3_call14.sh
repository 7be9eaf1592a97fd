#!/bin/bash
set -u
O=gpurun_out/r3_c14; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 120 python scripts/dbg_upp_graph.py fb 2 64 > $O/fb_small.log 2>&1; echo "fb small rc=$? ok=$(grep -c ok $O/fb_small.log) warn=$(grep -c AccumulateGrad $O/fb_small.log)"
timeout -k 10 120 python scripts/dbg_upp_graph.py full > $O/full.log 2>&1; echo "full rc=$? ok=$(grep -c ok $O/full.log)"
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -q -x -k "graphed or row_buffers or virtual_concat" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > $O/upp.json 2> $O/upp.err; echo "upp graph rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --no-graph > $O/upp_nograph.json 2> $O/upp_nograph.err; echo "upp nograph rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 > $O/upp828.json 2> $O/upp828.err; echo "upp828 graph rc=$?"
python - <<'PY'
import json
for n in ("upp","upp_nograph","upp828"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c14/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], d["config"].get("launch"))
    except Exception as e: print(n,"ERR",e)
PY
