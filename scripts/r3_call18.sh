#!/bin/bash
set -u
O=gpurun_out/r3_c18; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist > $O/dp1_$i.json 2> $O/dp1_$i.err; echo "dp1 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist --no-sync-bn > $O/dp1_nosync_$i.json 2> $O/dp1_nosync_$i.err; echo "dp1 nosync rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/n1_eager_$i.json 2> $O/n1_eager_$i.err; echo "n1 eager rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/n1_graph_$i.json 2> $O/n1_graph_$i.err; echo "n1 graph rc=$?"
done
python - <<'PY'
import json
for i in (1,2):
  for n in ("dp1","dp1_nosync","n1_eager","n1_graph"):
    try:
        d=json.loads(open(f"gpurun_out/r3_c18/{n}_{i}.json").read().strip().splitlines()[-1]); print(n, i, d["ms_per_step"], d["value"], d["final_loss"], d["config"]["parallelism"], d["config"]["sync_bn"])
    except Exception as e: print(n,"ERR",e)
PY
