# same-box A/B of two environment settings on the conv layer bench: bash scripts/run_ab.sh "ENV_A=1" "ENV_B=1"   (in-tree library)
rm -f gpurun_out/conv_abl.txt
for i in 1 2; do
  for e in "$@"; do
    echo "== [$e]" >> gpurun_out/conv_abl.txt
    env $e LAYERS=${LAYERS:-conv0_0.conv2,conv0_1.conv1,conv3_1.conv1,conv2_0.conv2,conv1_1.conv1} timeout -k 10 120 python scripts/conv_layer_bench.py 2>&1 | grep -E "^conv" >> gpurun_out/conv_abl.txt || exit 1
  done
done
cat gpurun_out/conv_abl.txt
