#!/bin/bash
# conv_kernel<64> between library builds: the stride-2 data gradient with 64 gradient channels (its main C3 customer)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "3x3s2 64->128 @320"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 20 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
