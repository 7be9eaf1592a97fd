#!/bin/bash
set -u
O=gpurun_out/r3_c13; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 120 python scripts/dbg_upp_graph.py fb 2 64 > $O/$name.log 2>&1; echo "$name rc=$? ok=$(grep -c ok $O/$name.log)"; }
run bf32 BF=32
run wg16off MAU_WGRAD16=0
run fusedbn0 MAU_FUSED_BN=0
run fusedup0 MAU_FUSED_UP=0
run fusedred0 MAU_FUSED_REDUCE=0
run m16off MAU_CONV_M16=0
run packmulti0 MAU_PACK_MULTI=0
run allold MAU_WGRAD16=0 MAU_FUSED_BN=0 MAU_FUSED_UP=0 MAU_FUSED_REDUCE=0 MAU_CONV_M16=0 MAU_PACK_MULTI=0
AMD_LOG_LEVEL=3 timeout -k 10 120 python scripts/dbg_upp_graph.py fb 2 64 2>&1 | tail -n 400 > $O/amdlog_tail.log; echo "amdlog done"
