#!/bin/bash
# round 6, call 10: MaxPool2d(2,2) inside the inference epilogue (mau_conv3x3_fwd_pool).  (1) bitwise against conv + maxpool, the inference
# model tests  (2) 512 x 512 inference, bf16 B=8 / fp16 B=8 / the app's shape B=1, two runs each (the previous library has no such entry
# point: its session runs the separate maxpool launches -- that is round 6's record, profiles/r6/bench_infer512_*.json)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c10; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "fwd_pool or inference or infer or config5 or graphed or eval or sweep or freeze" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
for i in 1 2; do
  for a in "--batch 8 --precision bf16" "--batch 8 --precision fp16" "--batch 1 --channels 23 --meta 8 --precision fp16"; do
    echo "== $a"; timeout -k 10 200 python bench.py --no-cpu-baseline --infer --size 512 $a 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): d=json.loads(l); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'
  done
done | tee $O/infer.txt
