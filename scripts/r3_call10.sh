#!/bin/bash
set -u
O=gpurun_out/r3_c10; mkdir -p $O
export TMPDIR=/tmp
export LAYERS=conv0_0.conv2,conv0_1.conv1,conv1_0.conv2,conv2_1.conv1
rm -f gpurun_out/conv_abl.txt
for rep in 1 2; do
  bash scripts/conv_ablation.sh pipe nowait nodma noepi || exit 1
done
cp gpurun_out/conv_abl.txt $O/conv_abl.txt; cat $O/conv_abl.txt
