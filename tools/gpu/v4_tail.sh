#!/bin/bash
# A/B of conv_v4's tail tiles (DY_V4_TAIL: 0 = off, 2 / 3 = forced height, unset = the launcher's choice) on the diagnostics library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/diag
B=${1:-64}
for shape in "3x3 256->256 @40" "3x3 256->256 @80" "3x3 512->512 @40" "1x1 1280->512 @40" "1x1 2048->512 @40" "3x3s2 256->512 @80"; do
  for mode in 0 auto 3 2; do
    if [ $mode = auto ]; then unset DY_V4_TAIL; else export DY_V4_TAIL=$mode; fi
    echo "== $shape tail=$mode"
    CB_ONLY="$shape" CB_CHECK=1 timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 $B 2>&1 | grep -v "^$" | tail -4
  done
done
