#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_c8; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/pmc1 -- python3 scripts/first_layer_bench.py > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc2 -- python3 scripts/first_layer_bench.py > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_c8"
for d in ("pmc1","pmc2"):
    fs=glob.glob(f"{O}/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        if "first" in k or "conv3x3_bf16" in k:
            print(d,k,len(n[k]),{c:round(v/len(n[k])) for c,v in agg[k].items()})
PY
