#!/bin/bash
# round 6, call 5 (TIMING ONLY): what BatchNorm-apply + ReLU fused into the consuming convolution's load would cost with an LDS-DMA loader --
# a fix-up pass over each stage's halo image in LDS + one more barrier per stage (-DMAU_CONV_PROBE_LDSFIX, variants/libmau_ldsfix.so,
# garbage results) against the library, same call, alternating; forward column = statistics epilogue, as in the training step
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c5; mkdir -p $O
P=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_ldsfix.so
for i in 1 2; do for L in lib probe; do
  if [ $L = probe ]; then export MAU_LIB=$P; else unset MAU_LIB; fi
  echo "== $L"; B=32 LAYERS=conv0_0.conv2,conv0_1.conv1,conv0_1.conv2,conv1_0.conv2,conv1_1.conv1,conv2_0.conv2,conv2_1.conv1,conv3_1.conv1 timeout -k 10 300 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv"
done; done | tee $O/lds_fixup_probe.txt
unset MAU_LIB
timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -E "^(pool|up|head)" | tee $O/fused_bn.txt
timeout -k 10 120 python scripts/elementwise_bench.py 2>&1 | grep -v "^/opt" | tail -30 | tee $O/elementwise.txt
