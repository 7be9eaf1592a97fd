#!/bin/bash
# round-5 records on ONE box.  Through gpurun; outputs -> gpurun_out/rec5/
set -u
export TMPDIR=/tmp
R=gpurun_out/rec5; mkdir -p $R
# 1. rocprofv3 passes of the headline workload (eager launches: a hipGraph replay is one dispatch to the profiler)
bash scripts/profile.sh r5 --no-graph --repeats 1
python scripts/summarize_profile.py gpurun_out/prof_r5 gpurun_out/r5_profiles unet_bf16_b32_s256_c6_train > $R/profiles_summ.txt 2>&1; echo "summ rc=$?"
mkdir -p profiles/r5 && cp gpurun_out/r5_profiles/pmc_summary.json profiles/r5/pmc_summary.json     # bench.py below reads this round's PMC record (same libmau_hip.so)
# 2. bench lines of every BASELINE config
jl() { python scripts/json_only.py; }
# (the one-iteration B=32 CPU baseline first: its record is what the default line quotes as cpu_baseline.b32_recorded)
python bench.py --cpu-baseline-b32 --repeats 4 2>> $R/err.txt | jl > $R/bench_default_cpu_b32.json; echo "cpu b32 rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/rec5/bench_default_cpu_b32.json"))["cpu_baseline"]
rec = dict(r["b32_one_iteration"], host_logical_cpus=r["host_logical_cpus"], b2_sample_images_s_same_run=r["value"])
json.dump(rec, open("profiles/r5/cpu_baseline_b32.json", "w"), indent=1)
print("cpu b32", rec)
PY
python bench.py 2>> $R/err.txt | jl > $R/bench_default.json; echo "default rc=$?"
python bench.py --no-cpu-baseline --no-graph 2>> $R/err.txt | jl > $R/bench_default_eager.json
MAU_DP_GRAPH=0 python bench.py --no-cpu-baseline --force-dist 2>> $R/err.txt | jl > $R/bench_dp1_forced_eager.json
MAU_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dist 2>> $R/err.txt | jl > $R/bench_dp1_forced_graph.json
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 2>> $R/err.txt | jl > $R/bench_unetpp_b16.json
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 2>> $R/err.txt | jl > $R/bench_unetpp_b16_T828.json
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision fp16 2>> $R/err.txt | jl > $R/bench_infer512_fp16.json
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision bf16 2>> $R/err.txt | jl > $R/bench_infer512_bf16_b8.json
python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 2>> $R/err.txt | jl > $R/bench_infer512_fp16_b1_c23.json
python bench.py --no-cpu-baseline --precision fp32 --batch 8 2>> $R/err.txt | jl > $R/bench_fp32_b8.json
echo "bench lines done"
# 3. per-layer timing and per-layer HBM traffic of the convolution kernels
OUT=$R/conv_layers.json timeout -k 10 300 python scripts/conv_layer_bench.py > $R/conv_layers.txt 2>&1; echo "layers rc=$?"
mkdir -p gpurun_out/layer_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/layer_pmc/pmc_fetch -- python3 scripts/conv_layer_bench.py > gpurun_out/layer_pmc/fetch.log 2>&1; echo "layer fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/layer_pmc/pmc_write -- python3 scripts/conv_layer_bench.py > gpurun_out/layer_pmc/write.log 2>&1; echo "layer write rc=$?"
python scripts/layer_traffic.py gpurun_out/layer_pmc $R/layer_traffic.json > $R/layer_traffic.txt 2>&1; echo "traffic rc=$?"; tail -4 $R/layer_traffic.txt
# 4. one-step kernel traces
bash scripts/r5_trace.sh unet > /dev/null 2>&1; cp gpurun_out/r5_trace_unet/step.txt $R/step_trace.txt
bash scripts/r5_trace.sh upp --model-type unet++ --batch 16 > /dev/null 2>&1; cp gpurun_out/r5_trace_upp/step.txt $R/step_trace_unetpp.txt
timeout -k 10 120 python scripts/first_layer_bench.py > $R/first_layer.txt 2>&1; echo "first rc=$?"
timeout -k 10 120 python scripts/fused_bn_bench.py > $R/fused_bn.txt 2>&1; echo "fused rc=$?"
