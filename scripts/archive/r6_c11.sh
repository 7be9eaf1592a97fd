#!/bin/bash
# round 6, call 11: MaxPool2d(2,2) inside the inference epilogue, off / on in the same call (functional._INFER_POOL_FUSED through
# scripts/bench_with.py), three alternations: 512 x 512 bf16 B=8, fp16 B=8, the app's shape (B=1, 23 channels, fp16)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c11; mkdir -p $O
for i in 1 2 3; do for F in False True; do
  for a in "--batch 8 --precision bf16" "--batch 8 --precision fp16" "--batch 1 --channels 23 --meta 8 --precision fp16"; do
    echo "== fused=$F $a"; timeout -k 10 200 python scripts/bench_with.py functional._INFER_POOL_FUSED=$F -- --no-cpu-baseline --infer --size 512 $a 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): d=json.loads(l); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'
  done
done; done | tee $O/infer_pool_ab.txt
