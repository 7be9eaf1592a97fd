#!/bin/bash
# round 4, check 15 (TIMING ONLY): the convolution without its weight-slab DMAs (variants/libmau_nowdma.so = -DMAU_CONV_ABL_NOWDMA) --
# an upper bound on what weights RESIDENT in LDS (K = 64 layers: 72 KB) would buy the level-0 launches; MAU_CONV_L0 = 1 / 0
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c15; mkdir -p $O
V=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_nowdma.so
for L0 in 1 0; do
  for L in "" $V "" $V; do
    tag=$([ -z "$L" ] && echo base || echo nowdma)
    MAU_CONV_L0=$L0 MAU_LIB=$L LAYERS=conv0_0,conv0_1,conv1_0.conv1 timeout -k 10 200 python scripts/conv_layer_bench.py 2>/dev/null | grep -E "^conv|TOTAL" > $O/layers_${tag}_l0$L0.txt; echo "== L0=$L0 $tag"; cat $O/layers_${tag}_l0$L0.txt
  done
done
