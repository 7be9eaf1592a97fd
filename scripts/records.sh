#!/bin/bash
# The committed records of a round, measured on ONE box with the library in the tree (through gpurun):
#     bash scripts/records.sh r6 [quick|a|b]    -> gpurun_out/rec_r6/ (copy what is to be judged into profiles/r6/)
# (a call through gpurun is limited to 20 minutes: `a` = the headline workload's profile passes, the default line with the one-iteration
#  B = 32 CPU baseline, per-layer timing, the step traces; `b` = every other config, their PMC passes, per-layer traffic.  `b` needs
#  profiles/<round>/pmc_summary.json of `a` IN THE TREE -- copy it there between the two calls.)
# 1. rocprofv3 --kernel-trace --stats + the PMC passes of the headline workload (scripts/profile.sh; eager launches: a hipGraph replay is
#    one dispatch to the profiler) -> pmc_summary.json keyed by workload, carrying the library's sha256 (bench.py quotes `traffic` from it
#    only when the hash matches the library it loaded);  2. the bench lines of every BASELINE config;  3. per-layer timing and per-layer
#    HBM traffic of the convolution kernels;  4. one-step kernel traces;  5. the small kernel benches.   `quick`: 1, the default line, 3 (timing) and 4 only.
set -u
export TMPDIR=/tmp
RND=${1:-r6}; PART=${2:-}
QUICK=; [ "$PART" = quick ] && QUICK=1
A=1; B=1; [ "$PART" = a ] && B=; [ "$PART" = b ] && A=; [ -n "$QUICK" ] && B=
R=gpurun_out/rec_$RND; mkdir -p $R profiles/$RND gpurun_out/${RND}_profiles
jl() { python scripts/json_only.py; }
if [ -n "$A" ]; then
bash scripts/profile.sh $RND --no-graph --repeats 1
python scripts/summarize_profile.py gpurun_out/prof_$RND gpurun_out/${RND}_profiles unet_bf16_b32_s256_c6_train > $R/profiles_summ.txt 2>&1; echo "summ rc=$?"
cp gpurun_out/${RND}_profiles/pmc_summary.json profiles/$RND/pmc_summary.json     # bench.py below reads this round's PMC record (same libmau_hip.so)
else
cp profiles/$RND/pmc_summary.json gpurun_out/${RND}_profiles/pmc_summary.json
fi
if [ -n "$A" ] && [ -z "$QUICK" ]; then
  python bench.py --cpu-baseline-b32 --repeats 4 2>> $R/err.txt | jl > $R/bench_default_cpu_b32.json; echo "cpu b32 rc=$?"
  RND=$RND python - <<'PY'
import json, os
rnd = os.environ["RND"]
r = json.load(open(f"gpurun_out/rec_{rnd}/bench_default_cpu_b32.json"))["cpu_baseline"]
rec = dict(r["b32_one_iteration"], host_logical_cpus=r["host_logical_cpus"], b2_sample_images_s_same_run=r["value"])
json.dump(rec, open(f"profiles/{rnd}/cpu_baseline_b32.json", "w"), indent=1)
print("cpu b32", rec)
PY
fi
[ -n "$A" ] && { python bench.py 2>> $R/err.txt | jl > $R/bench_default.json; echo "default rc=$?"; }
if [ -n "$B" ]; then
  python bench.py --no-cpu-baseline --no-graph 2>> $R/err.txt | jl > $R/bench_default_eager.json
  MAU_DP_GRAPH=0 python bench.py --no-cpu-baseline --force-dist 2>> $R/err.txt | jl > $R/bench_dp1_forced_eager.json
  MAU_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dist 2>> $R/err.txt | jl > $R/bench_dp1_forced_graph.json
  python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 2>> $R/err.txt | jl > $R/bench_unetpp_b16_T828.json
  python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision fp16 2>> $R/err.txt | jl > $R/bench_infer512_fp16.json
  python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 2>> $R/err.txt | jl > $R/bench_infer512_fp16_b1_c23.json
  python bench.py --no-cpu-baseline --precision fp32 --batch 8 2>> $R/err.txt | jl > $R/bench_fp32_b8.json
  # PMC records of the other BASELINE workloads, then their bench lines (which quote them)
  bash scripts/profile.sh ${RND}upp --no-graph --repeats 1 --model-type unet++ --batch 16
  python scripts/summarize_profile.py gpurun_out/prof_${RND}upp gpurun_out/${RND}_profiles unet++_bf16_b16_s256_c6_train > $R/profiles_summ_unetpp.txt 2>&1; echo "summ upp rc=$?"
  bash scripts/profile.sh ${RND}inf --no-graph --repeats 1 --infer --size 512 --batch 8
  python scripts/summarize_profile.py gpurun_out/prof_${RND}inf gpurun_out/${RND}_profiles unet_bf16_b8_s512_c6_infer > $R/profiles_summ_infer.txt 2>&1; echo "summ inf rc=$?"
  cp gpurun_out/${RND}_profiles/pmc_summary.json profiles/$RND/pmc_summary.json
  python bench.py --no-cpu-baseline --model-type unet++ --batch 16 2>> $R/err.txt | jl > $R/bench_unetpp_b16.json
  python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision bf16 2>> $R/err.txt | jl > $R/bench_infer512_bf16_b8.json
fi
echo "bench lines done"
[ -n "$A" ] && { OUT=$R/conv_layers.json timeout -k 10 300 python scripts/conv_layer_bench.py > $R/conv_layers.txt 2>&1; echo "layers rc=$?"; }
if [ -n "$B" ]; then
  mkdir -p gpurun_out/layer_pmc
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/layer_pmc/pmc_fetch -- python3 scripts/conv_layer_bench.py > gpurun_out/layer_pmc/fetch.log 2>&1; echo "layer fetch rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/layer_pmc/pmc_write -- python3 scripts/conv_layer_bench.py > gpurun_out/layer_pmc/write.log 2>&1; echo "layer write rc=$?"
  python scripts/layer_traffic.py gpurun_out/layer_pmc $R/layer_traffic.json > $R/layer_traffic.txt 2>&1; echo "traffic rc=$?"; tail -4 $R/layer_traffic.txt
fi
if [ -n "$A" ]; then
bash scripts/trace_step.sh unet > /dev/null 2>&1; cp gpurun_out/trace_unet/step.txt $R/step_trace.txt
bash scripts/trace_step.sh upp --model-type unet++ --batch 16 > /dev/null 2>&1; cp gpurun_out/trace_upp/step.txt $R/step_trace_unetpp.txt
fi
if [ -n "$B" ]; then
  timeout -k 10 120 python scripts/first_layer_bench.py > $R/first_layer.txt 2>&1; echo "first rc=$?"
  timeout -k 10 120 python scripts/fused_bn_bench.py > $R/fused_bn.txt 2>&1; echo "fused rc=$?"
fi
for f in $R/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['roofline'].get('frac'), d['roofline'].get('traffic'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
