#!/bin/bash
# same-box A/B of the BatchNorm streaming kernels (tools/bin/bn_bench, L shapes) between library builds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in $(cd $ROOT/dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    echo "== rep $rep lib ${v:-current}"
    BB_ONLY="L " timeout -k 10 200 $ROOT/tools/bin/bn_bench 20 2>&1 | grep -v "^$"
  done
done
