export MAU_LIB=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_big.so
MAU_CONV_BIGWAVE=1 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" > gpurun_out/conv_test_big.log 2>&1; tail -n 3 gpurun_out/conv_test_big.log
rm -f gpurun_out/conv_abl.txt
bash scripts/conv_ablation.sh big && MAU_CONV_BIGWAVE=1 bash scripts/conv_ablation.sh big && bash scripts/conv_ablation.sh big && MAU_CONV_BIGWAVE=1 bash scripts/conv_ablation.sh big; cat gpurun_out/conv_abl.txt
