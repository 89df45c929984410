#!/bin/bash
# what bounds the band weight gradient: the in-tree library against variants built with -DWG3_ABLATE=1|2|3 into dedark_yolo_amd/lib/varAbl<n>/
# (wgrad_v3.hip; results of the variants are wrong by construction, only the wgrad time matters)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in "" $(cd $ROOT/dedark_yolo_amd/lib && ls -d varAbl* 2>/dev/null); do
    export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/$v
    for shape in "s2 3x3 128->128 @80" "s1 3x3 64->64 @160"; do
      echo -n "rep $rep lib ${v:-current} | "
      CB_ONLY="$shape" timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1
    done
  done
done
