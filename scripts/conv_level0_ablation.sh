#!/bin/bash
# Level-0 ablation of the conv kernel's 16x16x32 path (timing-only builds; results are garbage, the instruction streams are measured).
# Build first, here:   bash scripts/build_variants.sh conv3x3_bf16.hip base:"" nodma:"-DMAU_CONV_ABL_NODMA" noread:"-DMAU_CONV_ABL_NOREAD" \
#     noepi:"-DMAU_CONV_ABL_NOEPI -DMAU_CONV_NO_COUNTED_EPI" nostats:"-DMAU_CONV_ABL_NOSTATS" nostore:"-DMAU_CONV_ABL_NOSTORE -DMAU_CONV_NO_COUNTED_EPI"
# then through gpurun: bash scripts/conv_level0_ablation.sh     -> gpurun_out/conv_level0_ablation/
set -u
O=gpurun_out/conv_level0_ablation; mkdir -p $O
export TMPDIR=/tmp
export LAYERS=${LAYERS:-conv0_0.conv2,conv0_1.conv1,conv1_0.conv2}
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh base nodma noread noepi nostats nostore || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
# instruction mix and waits of the 64->64 layer (two PMC passes)
LAYERS=conv0_0.conv2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc0 -- python3 scripts/conv_layer_bench.py > $O/pmc0.log 2>&1
LAYERS=conv0_0.conv2 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc1 -- python3 scripts/conv_layer_bench.py > $O/pmc1.log 2>&1
echo pmc rc=$?
