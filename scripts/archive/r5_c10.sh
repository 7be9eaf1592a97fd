#!/bin/bash
# VERDICT r4 #5, measured: the transposed level-0 epilogue (weights as the A operand, stores straight from registers) for the launches
# WITHOUT statistics (data gradients, inference): exactness first, then per layer and on the step / the 512x512 inference, MAU_CONV_TR=0/1
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c10; mkdir -p $O
MAU_CONV_TR=1 timeout -k 10 600 python scripts/conv_big_tile_check.py > $O/big_tile_tr.txt 2>&1; echo "big tile rc=$?"; tail -2 $O/big_tile_tr.txt
MAU_CONV_TR=1 timeout -k 10 600 python -m pytest tests/test_gpu_properties_full_size.py -m gpu -q -k "forward_and_data" > $O/pytest_prop_tr.txt 2>&1; echo "properties rc=$?"; tail -2 $O/pytest_prop_tr.txt
for rep in 1 2; do for TR in 0 1; do
  echo "== TR=$TR training shapes (dgrad column = plain epilogue)"; MAU_CONV_TR=$TR LAYERS=conv0_0.conv2,conv0_1,conv1_0.conv1 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv"
  echo "== TR=$TR inference shapes B=8 512 (fwd column = inference epilogue)"; MAU_CONV_TR=$TR EPI=post B=8 S=512 LAYERS=conv0_0.conv2,conv0_1,conv1_0.conv1 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv"
done; done 2>&1 | tee $O/layers_ab.txt
for rep in 1 2; do for TR in 0 1; do
  for args in "--repeats 8" "--infer --size 512 --batch 8"; do
    MAU_CONV_TR=$TR python bench.py --no-cpu-baseline $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('TR=$TR', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r.get('final_loss')))"
  done
done; done 2>&1 | tee $O/step_ab.txt
