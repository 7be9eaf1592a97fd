#!/bin/bash
# round 4, check 17: the narrow-workgroup rules must not cost the training configs anything (U-Net++ B=16: its 16 x 16 level now
# runs <64,2,4>; U-Net B=32: untouched by construction)
set -u
export TMPDIR=/tmp
for NAR in 0 1 0 1; do
  for args in "--model-type unet++ --batch 16" ""; do
    MAU_CONV_NARROW=$NAR python bench.py --no-cpu-baseline --repeats 8 $args 2>/dev/null | python scripts/json_only.py | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('narrow=$NAR', '$args', r['ms_per_step'], r['value'], r['roofline']['frac'], repr(r['final_loss']))"
  done
done
