#!/bin/bash
set -u
O=gpurun_out/r3_c9; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
export LAYERS=conv0_0.conv2,conv0_1.conv1,conv1_0.conv2,conv2_1.conv1,conv3_1.conv1
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh base pk pipe || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
for i in 1 2; do
  for v in pipe pk base; do
    MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err; echo "bench rc=$?"
  done
done
python - <<'PY'
import json
for i in (1,2):
  for v in ("pipe","pk","base"):
    n=f"bench_{v}_{i}"
    try:
        d=json.loads(open(f"gpurun_out/r3_c9/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"])
    except Exception as e: print(n,"ERR",e)
PY
