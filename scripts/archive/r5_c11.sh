#!/bin/bash
# the loader's slot coordinates recomputed per item instead of hoisted-and-spilled (conv3x3_bf16.hip setup()): exactness, then
# (before the call: cp the library of the commit to compare against to metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_hip_head.so -- the directory is git-ignored and travels with gpurun)
# HEAD's library (variants/libmau_hip_head.so) against the new one, same call, alternating: per layer, step, inference
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c11; mkdir -p $O
V=metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_hip_head.so
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties_full_size.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
for rep in 1 2; do for L in head new; do
  if [ $L = head ]; then export MAU_LIB=$PWD/$V; else unset MAU_LIB; fi
  echo "== $L training shapes B=32 256"; python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv|^TOTAL"
  echo "== $L inference epilogue B=8 512"; EPI=post B=8 S=512 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv|^TOTAL"
done; done 2>&1 | tee $O/layers_ab.txt
for rep in 1 2 3; do for L in head new; do
  if [ $L = head ]; then export MAU_LIB=$PWD/$V; else unset MAU_LIB; fi
  for args in "--repeats 8" "--model-type unet++ --batch 16 --repeats 8" "--infer --size 512 --batch 8" "--infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16"; do
    python bench.py --no-cpu-baseline $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$L', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r.get('final_loss')))"
  done
done; done 2>&1 | tee $O/step_ab.txt
