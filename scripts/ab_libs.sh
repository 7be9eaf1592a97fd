#!/bin/bash
# Same-call A/B of two (or more) builds of libmau_hip.so with the same ABI, alternating (through gpurun: boxes differ by ~5 %, so only
# numbers of ONE call are ever compared):
#     bash scripts/ab_libs.sh <rounds> <label>=<path|""> [<label>=<path> ...] -- <command ...>
#   e.g. bash scripts/ab_libs.sh 5 base=$PWD/metadata-augmented-unet-for-lst-ndvi_amd/variants/libmau_base.so final= -- python bench.py --no-cpu-baseline --repeats 12
# An empty path = the in-tree library.  Variant builds: scripts/build_variants.sh <file.hip> name:"-DFLAG ...".  The command's stdout is
# printed behind a "== <label> <round>" line; a bench.py JSON line is condensed to ms_per_step / value / conv frac.
set -u
N=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
shift
for i in $(seq 1 $N); do
  for spec in "${LIBS[@]}"; do
    label=${spec%%=*}; path=${spec#*=}
    if [ -n "$path" ]; then export MAU_LIB=$path; else unset MAU_LIB; fi
    echo "== $label $i"
    "$@" 2>/dev/null | python -c '
import sys, json
for l in sys.stdin:
    l = l.rstrip("\n")
    if l.startswith("{"):
        try:
            d = json.loads(l); print("ms_per_step", d["ms_per_step"], "value", d["value"], "conv frac", d.get("roofline", {}).get("frac")); continue
        except ValueError:
            pass
    if l and not l.startswith("/opt"): print(l)
'
  done
done
unset MAU_LIB
