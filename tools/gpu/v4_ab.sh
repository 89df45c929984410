#!/bin/bash
# same-box A/B/C of conv_v4 variants built side by side (dedark_yolo_amd/lib/var*/): interleaved repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "3x3 256->256 @40" "3x3 512->512 @40" "3x3 256->256 @80" "1x1 2048->512 @40"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" timeout -k 10 120 $ROOT/tools/bin/conv_bench 40 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
