#!/bin/bash
# full GPU suite on the current tree + level-0 ablation timing of the conv kernel (timing-only builds) + one PMC pass
set -u
O=gpurun_out/r3_c7; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
export LAYERS=conv0_0.conv2,conv0_1.conv1,conv1_0.conv2
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh base nodma noread noepi nostats nostore || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
LAYERS=conv0_0.conv2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc0 -- python3 scripts/conv_layer_bench.py > $O/pmc0.log 2>&1
LAYERS=conv0_0.conv2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc1 -- python3 scripts/conv_layer_bench.py > $O/pmc1.log 2>&1
echo pmc rc=$?
