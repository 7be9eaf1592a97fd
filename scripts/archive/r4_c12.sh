#!/bin/bash
# round 4, check 12 (TIMING ONLY): what a channel-blocked input layout ([C/16][pixels][16]: every 16-channel stage of the
# convolution's loader reads contiguous 576-byte halo rows instead of 32 bytes of each pixel's 128...384-byte line) would buy the
# level-0 / level-1 forward and data-gradient launches.  MAU_CONV_BLK_PROBE makes the loader address the SAME buffers as if blocked
# (valid addresses, meaningless values).
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c12; mkdir -p $O
for P in 0 1 0 1; do
  if [ $P = 1 ]; then export MAU_CONV_BLK_PROBE=1; else unset MAU_CONV_BLK_PROBE; fi
  LAYERS=conv0_0,conv0_1,conv1_0,conv1_1 timeout -k 10 200 python scripts/conv_layer_bench.py > $O/layers_p$P.txt 2>&1; echo "probe=$P rc=$?"; grep -v amdgpu.ids $O/layers_p$P.txt
done
