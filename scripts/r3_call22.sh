#!/bin/bash
set -u
O=gpurun_out/r3_c22; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "fused or head or graphed or reproduc or single_launch or golden or known" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in headold headnew headold headnew; do
  echo "== $v"; MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so WHICH=head timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -v libdrm
done
for i in 1 2; do
  for v in headnew headold; do
    MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err; echo "bench rc=$?"
  done
done
python - <<'PY'
import json
for i in (1,2):
  for v in ("headnew","headold"):
    n=f"bench_{v}_{i}"
    try:
        d=json.loads(open(f"gpurun_out/r3_c22/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
