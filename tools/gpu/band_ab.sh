#!/bin/bash
# band_kernel<64|128> between library builds (dedark_yolo_amd/lib/var*/ against the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "s1 3x3 64->64 @160" "s2 3x3 128->128 @80" "d 3x3 256->64 @80"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
