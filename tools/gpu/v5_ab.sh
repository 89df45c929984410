#!/bin/bash
# same-box A/B of the conv_v5 / band / wgrad kernels: library variants built side by side (dedark_yolo_amd/lib/var*/), interleaved repetitions
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "s2 3x3 128->128 @80" "s1 3x3 64->64 @160" "1x1 256->256 @80" "3x3s2 64->128 @320" "1x1 320->128 @160" "3x3s2 128->256 @160" "1x1 128->128 @160"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
