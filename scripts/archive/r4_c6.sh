#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c6; mkdir -p $O
for v in "" u1 u4; do
  if [ -z "$v" ]; then L=""; else L="metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so"; fi
  echo "== variant '${v:-default}'"
  MAU_LIB=$L timeout -k 10 200 python scripts/first_layer_bench.py 2>/dev/null | head -2
done
