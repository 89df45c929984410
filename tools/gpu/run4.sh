#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
rm -f $O/n128_check.log
for pat in "128->128 @80" "320->128" "1280->512" "256->256 @40"; do
  CB_ONLY="$pat" CB_CHECK=1 timeout -k 5 120 tools/bin/conv_bench 20 64 >> $O/n128_check.log 2>&1 || echo "FAILED $pat rc=$?" >> $O/n128_check.log
done
timeout -k 5 120 tools/bin/v4_diag 64 128 80 3 > $O/diag_128_80.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 256 40 3 > $O/diag_256_40b.log 2>&1
timeout -k 5 300 tools/bin/conv_bench 20 64 > $O/v4_all2.log 2>&1
echo done
