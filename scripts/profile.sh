#!/bin/bash
# Collect the rocprofv3 evidence of one bench workload on the GPU box (run through gpurun):
#   scripts/profile.sh <tag> [bench.py arguments ...]     -> gpurun_out/prof_<tag>/{stats,pmc_*}/...
# kernel-trace/--stats and every --pmc set run as SEPARATE passes (gpurun refuses --pmc together
# with sys/hip/hsa tracing, and FETCH_SIZE / WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md).
set -u
TAG=${1:-r3}; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
export MAU_OVERLAP_WGRAD=0      # one stream: per-kernel times and counters belong to one kernel at a time
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
echo "$BENCH" > "$OUT/command.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
echo "stats pass rc=$?"
i=0
# (GRBM_GUI_ACTIVE rides with the SQ set -- the GRBM block has its own slots: MFMA-busy and clock per dispatch come from ONE pass)
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc_$i" -- $BENCH > "$OUT/pmc_$i.log" 2>&1
  echo "pmc pass $i ($PMC) rc=$?"
done
