#!/bin/bash
# round-3 records on ONE box: rocprofv3 passes of the headline workload (eager launches: a hipGraph replay shows up as one
# dispatch), bench lines of every BASELINE config, per-layer conv timing, a step trace.  Through gpurun; outputs -> gpurun_out/
set -u
export TMPDIR=/tmp
bash scripts/profile.sh r3 --no-graph
python scripts/summarize_profile.py gpurun_out/prof_r3 gpurun_out/r3_profiles unet_bf16_b32_s256_c6_train > gpurun_out/r3_profiles_summ.txt 2>&1; echo "summ rc=$?"
mkdir -p profiles/r3 && cp gpurun_out/r3_profiles/pmc_summary.json profiles/r3/pmc_summary.json     # bench.py below reads this round's PMC record
bash scripts/refresh_records.sh
# the data-parallel code path on ONE GPU under a 1-rank RCCL group (SyncBN + bucketed all-reduce): eager (the default at N > 1) and captured
python bench.py --no-cpu-baseline --force-dist > gpurun_out/rec/bench_dp1_forced_eager.json 2>> gpurun_out/rec/err.txt
MAU_DP_GRAPH=1 python bench.py --no-cpu-baseline --force-dist > gpurun_out/rec/bench_dp1_forced_graph.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --no-graph > gpurun_out/rec/bench_default_eager.json 2>> gpurun_out/rec/err.txt
OUT=gpurun_out/rec/conv_layers.json timeout -k 10 300 python scripts/conv_layer_bench.py > gpurun_out/rec/conv_layers.txt 2>&1; echo "layers rc=$?"
bash scripts/r3_trace.sh final > /dev/null 2>&1; cp gpurun_out/r3_trace_final/step.txt gpurun_out/rec/step_trace.txt
timeout -k 10 120 python scripts/fused_bn_bench.py > gpurun_out/rec/fused_bn.txt 2>&1; echo "fused rc=$?"
AB_VAR=MAU_WGRAD16 timeout -k 10 120 python scripts/wgrad_ab.py > gpurun_out/rec/wgrad_layers.txt 2>&1; echo "wgrad rc=$?"
