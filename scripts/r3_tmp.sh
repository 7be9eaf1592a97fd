#!/bin/bash
set -u
O=gpurun_out/r3_c33; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision fp16 > $O/i_fp16.json 2> $O/err.txt; echo rc=$?
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision fp16 --no-graph > $O/i_fp16_eager.json 2>> $O/err.txt; echo rc=$?
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 > $O/i_bf16.json 2>> $O/err.txt; echo rc=$?
python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 > $O/i_b1.json 2>> $O/err.txt; echo rc=$?
python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 --no-graph > $O/i_b1_eager.json 2>> $O/err.txt; echo rc=$?
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -x -k "config5 or inference or sweep" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_c33/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["launch"][:30])
    except Exception as e: print(f,"ERR",e)
PY
