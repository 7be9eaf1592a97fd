#!/bin/bash
# launch order inside ConvBNReLU.backward: the network's default (U-Net: weight-gradient branch first; U-Net++: data gradient first)
# against MAU_BWD_DGRAD_FIRST=0 / 1 forced, same call, alternating; then the model / full-size tests
set -u
export TMPDIR=/tmp
O=gpurun_out/r4_c20; mkdir -p $O
for P in default 0 1 default 0 1; do
  for args in "--model-type unet++ --batch 16" ""; do
    if [ $P = default ]; then unset MAU_BWD_DGRAD_FIRST; else export MAU_BWD_DGRAD_FIRST=$P; fi
    python bench.py --no-cpu-baseline --repeats 8 $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('dgrad_first=$P', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r['final_loss']))"
  done
done 2>&1 | tee $O/ab.txt
unset MAU_BWD_DGRAD_FIRST
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py tests/test_gpu_dist_rehearsal.py -m gpu -q -x > $O/pytest_model.txt 2>&1; echo "model rc=$?"; tail -3 $O/pytest_model.txt
