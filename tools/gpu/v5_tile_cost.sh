#!/bin/bash
# per-tile time split of conv_v5 on the short-K / high-resolution layers (s_memtime stamps of the first and a late block; DIAG library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/diag
for only in "3x3s2 64->128 @320" "3x3s2 128->256 @160" "1x1 320->128 @160" "1x1 128->128 @160" "1x1 1024->256 @80"; do
  echo "== $only"
  CB_V5=1 CB_ONLY="$only" DY_ABLATE=32 timeout -k 10 120 $ROOT/tools/bin/conv_bench 20 64 2>&1 | grep -v "^$"
done
