# Re-measure the committed bench / profile records of a round on one box: bash scripts/refresh_records.sh  (through gpurun)
set -u
mkdir -p gpurun_out/rec
python bench.py > gpurun_out/rec/bench_default.json 2> gpurun_out/rec/bench_default.err
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 > gpurun_out/rec/bench_unetpp_b16.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --model-type unet++ --batch 16 --seq-len 828 > gpurun_out/rec/bench_unetpp_b16_T828.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 --precision fp16 > gpurun_out/rec/bench_infer512_fp16.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --infer --size 512 --batch 8 > gpurun_out/rec/bench_infer512_bf16_b8.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 > gpurun_out/rec/bench_infer512_fp16_b1_c23.json 2>> gpurun_out/rec/err.txt
python bench.py --no-cpu-baseline --precision fp32 --batch 8 > gpurun_out/rec/bench_fp32_b8.json 2>> gpurun_out/rec/err.txt
tail -n 3 gpurun_out/rec/err.txt
for f in gpurun_out/rec/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d['roofline'].get('achieved'), d['roofline'].get('frac'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
