#!/bin/bash
set -u
O=gpurun_out/r3_c12; mkdir -p $O
export TMPDIR=/tmp
for m in fwd fwdtrain fb full; do
  timeout -k 10 120 python scripts/dbg_upp_graph.py $m > $O/$m.log 2>&1; echo "$m rc=$? $(grep -c ok $O/$m.log)"
done
timeout -k 10 120 python scripts/dbg_upp_graph.py fb 2 64 > $O/fb_small.log 2>&1; echo "fb small rc=$? $(grep -c ok $O/fb_small.log)"
BF=16 timeout -k 10 120 python scripts/dbg_upp_graph.py fb 2 64 > $O/fb_small16.log 2>&1; echo "fb small bf16 rc=$? $(grep -c ok $O/fb_small16.log)"
MT=unet timeout -k 10 120 python scripts/dbg_upp_graph.py fb > $O/fb_unet.log 2>&1; echo "fb unet rc=$? $(grep -c ok $O/fb_unet.log)"
