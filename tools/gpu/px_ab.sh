#!/bin/bash
# pixel-streaming 1x1 kernel (conv_px.hip) between library builds: the short-K layers of the 160x160 stage
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "1x1 128->128 @160" "1x1 320->128 @160" "1x1 64->64 @160" "1x1 1024->256 @80" "1x1 256->256 @80"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
