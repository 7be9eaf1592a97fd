#!/bin/bash
# round 6, call 12: the pooled inference epilogue in EVERY 16-bit tiling (the 32x32x16 epilogue too: single-tile inference at the deep levels).
# (1) bitwise tests + inference model tests  (2) off / on, same call, three alternations
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_c12; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "fwd_pool or inference or infer or config5 or graphed or eval or sweep or freeze or k_group" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
for i in 1 2 3; do for F in False True; do
  for a in "--batch 8 --precision bf16" "--batch 1 --channels 23 --meta 8 --precision fp16"; do
    echo "== fused=$F $a"; timeout -k 10 200 python scripts/bench_with.py functional._INFER_POOL_FUSED=$F -- --no-cpu-baseline --infer --size 512 $a 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): d=json.loads(l); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])'
  done
done; done | tee $O/infer_pool_ab.txt
