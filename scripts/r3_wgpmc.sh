#!/bin/bash
set -u
O=gpurun_out/r3_wgpmc; mkdir -p $O; export TMPDIR=/tmp
for L in ${LAYERS:-conv3_1.conv1 conv0_1.conv2}; do for M in 0 1; do
  export LAYER=$L MAU_WGRAD16=$M
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/a_${L}_$M -- python3 scripts/wg_one.py > $O/log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/b_${L}_$M -- python3 scripts/wg_one.py >> $O/log 2>&1
  echo "== $L wgrad16=$M"
  python scripts/pmc_summ.py $(ls $O/a_${L}_$M/*/*counter_collection.csv) wgrad
  python scripts/pmc_summ.py $(ls $O/b_${L}_$M/*/*counter_collection.csv) wgrad
  python - <<PY
import csv,glob
f=glob.glob("$O/b_${L}_$M/*/*kernel_trace.csv")[0]
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'wgrad' in r['Kernel_Name']]
print("avg us", sum(d)/len(d)/1e3, "n", len(d))
PY
done; done
