#!/bin/bash
# round 6, call 1: (1) the new full-size parity tests (fp32 1e-3 at B = 32, batch moments) + the tests touched by ABI 5
# (2) the round's baseline bench line on this box  (3) VERDICT r5 #1(i): the level-0 tile form by K -- <64,4,4> (two 4-wave workgroups
# per CU) against <64,4,8> (one 8-wave workgroup, 64 x 16 tile) on the U-Net's and the U-Net++'s full-resolution layers, same call, alternating
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -s -k "config2" > $O/pytest_full.txt 2>&1; echo "full rc=$?"; grep -E "config 2|passed|failed|Error|error" $O/pytest_full.txt | tail -20
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_capi_and_host.py -q -x -k "k_group or wgrad16_production or capi or abi or symbols" > $O/pytest_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/pytest_ops.txt
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/bench_default.json; echo
X="x0_1.conv1:208:64:256,x0_2.conv1:272:64:256,x0_3.conv1:336:64:256,x0_4.conv1:400:64:256"
for i in 1 2; do for L0 in 1 0; do
  echo "== MAU_CONV_L0=$L0 B=16 (U-Net++ full-resolution layers)"; MAU_CONV_L0=$L0 B=16 LAYERS=conv0_0.conv2,conv0_1.conv1,x0_ EXTRA=$X timeout -k 10 200 python scripts/conv_layer_bench.py 2>&1 | grep -E "^(conv|x0)"
  echo "== MAU_CONV_L0=$L0 B=32"; MAU_CONV_L0=$L0 B=32 LAYERS=conv0_0.conv2,conv0_1.conv1,conv0_1.conv2 timeout -k 10 200 python scripts/conv_layer_bench.py 2>&1 | grep -E "^(conv|x0)"
done; done | tee $O/l0_by_k.txt
