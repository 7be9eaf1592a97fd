# timing A/B of conv kernel builds (scripts/build_variants.sh) on representative layers: bash scripts/conv_ablation.sh v1 v2 ...
for v in "$@"; do
  echo "== $v" >> gpurun_out/conv_abl.txt
  MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so LAYERS=${LAYERS:-conv0_0.conv2,conv0_1.conv1,conv3_1.conv1,conv2_0.conv2,conv1_1.conv1} timeout -k 10 120 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv" >> gpurun_out/conv_abl.txt || exit 1
done
