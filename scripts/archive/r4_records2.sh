#!/bin/bash
# PMC records of the other BASELINE workloads (same library as scripts/r4_records.sh): U-Net++ B=16 training, 512x512 bf16 inference
set -u
export TMPDIR=/tmp
R=gpurun_out/rec4; mkdir -p $R
mkdir -p gpurun_out/r4_profiles && cp profiles/r4/pmc_summary.json gpurun_out/r4_profiles/pmc_summary.json
bash scripts/profile.sh r4upp --no-graph --repeats 1 --model-type unet++ --batch 16
python scripts/summarize_profile.py gpurun_out/prof_r4upp gpurun_out/r4_profiles unet++_bf16_b16_s256_c6_train > $R/profiles_summ_unetpp.txt 2>&1; echo "summ upp rc=$?"
bash scripts/profile.sh r4inf --no-graph --repeats 1 --infer --size 512 --batch 8
python scripts/summarize_profile.py gpurun_out/prof_r4inf gpurun_out/r4_profiles unet_bf16_b8_s512_c6_infer > $R/profiles_summ_infer.txt 2>&1; echo "summ inf rc=$?"
cp gpurun_out/r4_profiles/pmc_summary.json profiles/r4/pmc_summary.json
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 2>> $R/err.txt | python scripts/json_only.py > $R/bench_unetpp_b16.json
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision bf16 2>> $R/err.txt | python scripts/json_only.py > $R/bench_infer512_bf16_b8.json
python - <<'PY'
import json
for f in ("bench_unetpp_b16", "bench_infer512_bf16_b8"):
    d = json.loads(open(f"gpurun_out/rec4/{f}.json").read())
    print(f, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:80])
PY
