#!/bin/bash
# kernel trace of the single-tile inference call (B=1, 512x512, 23 channels, fp16), eager launches
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c18; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-graph --repeats 1 --infer --size 512 --batch 1 --channels 23 --meta 8 --precision fp16 > $O/trace.log 2>&1
echo "trace rc=$?"
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/r4_c18/trace/*/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'first_fwd' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{d:8.1f} us  {r['Kernel_Name'][:110]}")
print("kernels", b - a, "kernel time", round(tot, 1), "us; span", (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
PY
