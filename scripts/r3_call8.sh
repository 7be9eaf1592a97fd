#!/bin/bash
# epilogue rewrite of the 16x16x32 conv path: full suite on the in-tree build (packed math) + same-box A/B base / scalar / packed
set -u
O=gpurun_out/r3_c8; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
export LAYERS=conv0_0.conv2,conv0_1.conv1,conv1_0.conv2,conv2_1.conv1,conv3_1.conv1
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh base scal pk || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_pk_$i.json 2> $O/bench_pk_$i.err; echo "bench rc=$?"
  MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_base.so timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_base_$i.json 2> $O/bench_base_$i.err; echo "bench rc=$?"
  MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_scal.so timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_scal_$i.json 2> $O/bench_scal_$i.err; echo "bench rc=$?"
done
python - <<'PY'
import json
for n in ("bench_pk_1","bench_base_1","bench_scal_1","bench_pk_2","bench_base_2","bench_scal_2"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c8/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
