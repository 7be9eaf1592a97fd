#!/bin/bash
# round 6, call 9: weight gradient -- the pixel tiles walked band by band, column-major (vertical neighbours back to back: the halo rows they
# share are re-read from L2) against the row-major walk of the record library (variants/libmau_rec6.so).  (1) exact / property tests
# (2) per-layer wgrad, alternating  (3) per-layer wgrad traffic (FETCH_SIZE / WRITE_SIZE passes) of the new walk  (4) step A/B, 4 alternations
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c9; mkdir -p $O
REC=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_rec6.so
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties_full_size.py -m gpu -q -x -k "wgrad or weight_gradient" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
bash scripts/ab_libs.sh 2 rec6=$REC colmajor= -- env B=32 LAYERS=conv0_0.conv2,conv1_0.conv1,conv1_0.conv2,conv2_0.conv2,conv1_1.conv1,conv0_1.conv1,conv0_1.conv2,conv2_1.conv1 python scripts/conv_layer_bench.py | grep -E "^(==|conv)" | tee $O/wgrad_walk_layers.txt
bash scripts/ab_libs.sh 4 rec6=$REC colmajor= -- python bench.py --no-cpu-baseline --repeats 12 | tee $O/step_unet.txt
bash scripts/ab_libs.sh 3 rec6=$REC colmajor= -- python bench.py --no-cpu-baseline --repeats 12 --model-type unet++ --batch 16 | tee $O/step_unetpp.txt
