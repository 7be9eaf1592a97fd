#!/bin/bash
set -u
export TMPDIR=/tmp
for B in 4 8 16 32 64; do echo "== B=$B"; B=$B timeout -k 10 200 python scripts/first_layer_bench.py 2>/dev/null | head -2; done
