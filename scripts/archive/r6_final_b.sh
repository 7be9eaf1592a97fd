#!/bin/bash
# round 6, final call B: part `b` of the round's records (the other BASELINE configs, their PMC passes, per-layer traffic); needs
# profiles/r6/pmc_summary.json of call A in the tree
set -u
export TMPDIR=/tmp
bash scripts/records.sh r6 b
