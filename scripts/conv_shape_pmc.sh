# Clock / MFMA-busy of the two multiply loops of the in-tree library on one layer (PMC pass each): bash scripts/conv_shape_pmc.sh
export TMPDIR=/tmp
export LAYERS=${LAYERS:-conv3_1.conv1}
rm -rf gpurun_out/abl_pmc
for m in 0 1; do
  export MAU_CONV_M16=$m
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/abl_pmc/m16_$m -- python3 scripts/conv_layer_bench.py > gpurun_out/abl_pmc_m16_$m.log 2>&1 || exit 1
done
python3 scripts/conv_ablation_pmc_summary.py
