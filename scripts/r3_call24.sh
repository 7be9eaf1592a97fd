#!/bin/bash
set -u
O=gpurun_out/r3_c24; mkdir -p $O
export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
MAU_OVERLAP_WGRAD=2 timeout -k 10 900 python -m pytest tests/test_gpu_model.py -q -x -k "graphed or reproduc or known or golden or multi_step or fused_adamw or packs_follow" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  for m in 0 2 1; do
    MAU_OVERLAP_WGRAD=$m timeout -k 10 200 python bench.py --no-cpu-baseline > $O/g_${m}_$i.json 2> $O/g_${m}_$i.err; echo "graph m=$m rc=$?"
  done
done
for m in 0 2; do
  MAU_OVERLAP_WGRAD=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-graph > $O/e_${m}.json 2> $O/e_${m}.err; echo "eager m=$m rc=$?"
done
MAU_OVERLAP_WGRAD=2 timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > $O/upp_2.json 2> $O/upp_2.err; echo "upp m=2 rc=$?"
MAU_OVERLAP_WGRAD=0 timeout -k 10 200 python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > $O/upp_0.json 2> $O/upp_0.err; echo "upp m=0 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_c24/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"], d["config"]["launch"][:12])
    except Exception as e: print(f,"ERR",e)
PY
