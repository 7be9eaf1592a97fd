#!/bin/bash
set -u
O=gpurun_out/r3_c15; mkdir -p $O
export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -q -x -k "graphed or row_buffers or side_stream" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 > $O/upp828.json 2> $O/upp828.err; echo "upp828 graph rc=$?"
MAU_OVERLAP_LSTM_GRAPH=0 timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 > $O/upp828_one.json 2> $O/upp828_one.err; echo "upp828 one-branch rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > $O/upp.json 2> $O/upp.err; echo "upp rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --temporal-embeddings --seq-len 828 > $O/unet828.json 2> $O/unet828.err; echo "unet828 rc=$?"
python - <<'PY'
import json
for n in ("upp828","upp828_one","upp","unet828"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c15/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], d["config"].get("launch"))
    except Exception as e: print(n,"ERR",e)
PY
