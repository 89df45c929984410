#!/bin/bash
# first v4 run: correctness samples + timing, A/B against the round-1 route
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
for pat in "256->256 @40" "256->256 @80" "512->512 @40" "512->512 @20" "1x1 256->256" "256->512"; do
  CB_ONLY="$pat" CB_CHECK=1 timeout -k 5 120 tools/bin/conv_bench 20 64 >> $O/v4_check.log 2>&1 || echo "FAILED $pat rc=$?" >> $O/v4_check.log
done
DY_NO_CONV_V4=1 timeout -k 5 200 tools/bin/conv_bench 20 64 > $O/v2_all.log 2>&1
timeout -k 5 200 tools/bin/conv_bench 20 64 > $O/v4_all.log 2>&1
echo done
