#!/bin/bash
# Is the conv kernel WAITING for its LDS-DMAs, or paying for issuing them?  Timing-only builds of the 16x16x32 path:
#   bash scripts/build_variants.sh conv3x3_bf16.hip base:"" nowait:"-DMAU_CONV_ABL_NOWAIT" nodma:"-DMAU_CONV_ABL_NODMA" noepi:"-DMAU_CONV_ABL_NOEPI -DMAU_CONV_NO_COUNTED_EPI"
# then through gpurun: bash scripts/conv_nowait_ablation.sh   (round 3: nowait == base on every layer, nodma -20 %: DESIGN.md section 4)
set -u
O=gpurun_out/conv_nowait_ablation; mkdir -p $O
export TMPDIR=/tmp
export LAYERS=${LAYERS:-conv0_0.conv2,conv0_1.conv1,conv1_0.conv2,conv2_1.conv1}
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh base nowait nodma noepi || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
