#!/bin/bash
# non-temporal stores in the adding epilogues (CB_ADD: data gradient + a second view): library builds side by side, B = 64
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "s1 3x3 64->64 @160" "s3 3x3 256->256 @80"; do
      for m in CB_ADD CB_ACC; do
        echo -n "rep $rep lib ${v:-current} $m | "
        env $m=1 CB_ONLY="$shape" timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1 | cut -c1-112
      done
    done
  done
done
