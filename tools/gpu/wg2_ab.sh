#!/bin/bash
# same-box A/B of the pipelined weight gradient (wgrad_v2) between library builds: the n-scale shapes it serves + the C2 step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "n 3x3 128->128 @20" "n 1x1 384->256 @20" "n 3x3 64->64 @40" "n 1x1 192->128 @40" "n 3x3s2 128->256 @40"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 32 n 2>&1 | grep -v "^$" | tail -1
    done
  done
done
