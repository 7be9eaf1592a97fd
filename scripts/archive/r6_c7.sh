#!/bin/bash
# round 6, call 7: head_bn_bwd_apply -- pixels per workgroup 1024 (library) / 2048 / 4096, scalar or register-pair arithmetic; same call, alternating
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c7; mkdir -p $O
V=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants
for i in 1 2; do for L in lib hap2k hap4k happ happ2k; do
  if [ $L = lib ]; then unset MAU_LIB; else export MAU_LIB=$V/libmau_$L.so; fi
  echo "== $L"; WHICH=head timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -E "^head"
done; done | tee $O/head_apply_variants.txt
