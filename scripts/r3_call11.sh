#!/bin/bash
set -u
O=gpurun_out/r3_c11; mkdir -p $O
export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > $O/upp.json 2> $O/upp.err; echo "upp graph rc=$?"
tail -5 $O/upp.err
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --no-graph > $O/upp_nograph.json 2> $O/upp_nograph.err; echo "upp nograph rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 > $O/upp828.json 2> $O/upp828.err; echo "upp828 graph rc=$?"
tail -5 $O/upp828.err
timeout -k 10 200 python bench.py --no-cpu-baseline --temporal-embeddings > $O/unet_temporal.json 2> $O/unet_temporal.err; echo "unet temporal rc=$?"
python - <<'PY'
import json
for n in ("upp","upp_nograph","upp828","unet_temporal"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c11/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], d["config"].get("launch"))
    except Exception as e: print(n,"ERR",e)
PY
