#!/bin/bash
# full GPU suite + default bench (graph, fused AdamW) + smoke on the current tree
set -u
O=gpurun_out/r3_c6; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"
done
tail -1 $O/bench_2.json
