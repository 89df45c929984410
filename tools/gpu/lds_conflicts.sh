#!/bin/bash
# LDS bank-conflict share per kernel: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE over the conv_bench shapes in CB_ONLY (all if unset)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/ldsc; mkdir -p $OUT; rm -f $OUT/*.csv
cd /tmp && export TMPDIR=/tmp
CB_WARM=2 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o m -- $ROOT/tools/bin/conv_bench 2 64 > $OUT/run.log 2>&1
cd $ROOT
python3 - "$(find $OUT -name '*counter_collection.csv' | head -1)" <<PY
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"][:70]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE",0)):
    a=v.get("SQ_LDS_IDX_ACTIVE",0)
    if a: print("%-72s active %.3g conflict %.3g share %.2f" % (k, a, v.get("SQ_LDS_BANK_CONFLICT",0), v.get("SQ_LDS_BANK_CONFLICT",0)/a))
PY
