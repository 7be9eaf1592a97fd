# clock / MFMA-busy of conv kernel builds on one layer: bash scripts/conv_ablation_pmc.sh v1 v2 ...   (PMC pass per build)
export TMPDIR=/tmp
for v in "$@"; do
  export MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_$v.so
  export LAYERS=${LAYERS:-conv3_1.conv1}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/abl_pmc/$v -- python3 scripts/conv_layer_bench.py > gpurun_out/abl_pmc_$v.log 2>&1 || exit 1
done
