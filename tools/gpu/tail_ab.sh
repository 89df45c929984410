#!/bin/bash
# dead K-step DMAs no longer issued at the end of a tile (conv_v4 / conv_v5 / band kernel): library builds side by side
# (dedark_yolo_amd/lib/var*/ against the in-tree library), forward / data gradient columns of tools/conv_bench, B = 64
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "s3 3x3 256->256 @80" "s4 3x3 512->512 @40" "s3 3x3 256->256 @40" "1x1 256->256 @80" "1x1 1024->256 @80" "1x1 2048->512 @40" "3x3s2 256->512 @80" "s2 3x3 128->128 @80" "s1 3x3 64->64 @160" "d 3x3 256->64 @80"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1 | cut -c1-112
    done
  done
done
