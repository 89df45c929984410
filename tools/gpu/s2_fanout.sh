#!/bin/bash
# stride-2 data gradients as the graph issues them (plain / accumulating into dx / with an addend view) between library builds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
  export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
  for shape in "3x3s2 256->512 @80" "3x3s2 512->512 @40" "3x3s2 256->256 @80" "3x3s2 128->256 @160" "3x3s2 64->128 @320"; do
    for mode in plain acc add; do
      unset CB_ACC CB_ADD
      [ $mode = acc ] && export CB_ACC=1
      [ $mode = add ] && export CB_ADD=1
      echo -n "lib ${v:-current} $mode | "
      CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
