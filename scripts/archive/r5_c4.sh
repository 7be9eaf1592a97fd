#!/bin/bash
# (1) fixed tests (K-group exact, 6-step tracking)  (2) wgrad split-count probe  (3) B=1 inference: per-layer table with the
# inference epilogue, K groups off / on, and a kernel trace of the B=1 fp16 23-channel session
set -u
export TMPDIR=/tmp
O=gpurun_out/r5_c4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "k_group" > $O/pytest_kg.txt 2>&1; echo "kg rc=$?"; tail -3 $O/pytest_kg.txt
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "six_bf16" -s > $O/pytest_six.txt 2>&1; echo "six rc=$?"; grep -E "losses hip|eval output|passed|failed" $O/pytest_six.txt
python scripts/wgrad_split_probe.py 2>&1 | grep -v "^/opt" | tee $O/wgrad_split_probe.txt
for KG in 0 1; do echo "== KG=$KG (forward column: inference epilogue)"; EPI=post MAU_CONV_KG=$KG B=1 S=512 python scripts/conv_layer_bench.py 2>&1 | grep -v "^/opt"; done | tee $O/layers_b1_post.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b1 -- python3 $GRAFT_REPO_ROOT/bench.py --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 --no-cpu-baseline --steps 50 --repeats 2 > $GRAFT_REPO_ROOT/$O/prof_b1.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/r5_c4/prof_b1/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:10.1f} us  avg {float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:110]}")
print("total kernel time us", tot / 1e3)
PY
