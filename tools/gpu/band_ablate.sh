#!/bin/bash
# what bounds the activation-band kernel: K-loop ablations of the DIAG build (make DIAG=1 OUT=../lib/diag/libdedark_yolo.so OBJDIR=../lib/diag/obj)
# DY_ABLATE bits: 2 band DMA out of range (nothing fetched, zeros land), 4 the same for the weight tiles, 8 no MFMA, 16 no fragment reads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export LD_LIBRARY_PATH=$ROOT/dedark_yolo_amd/lib/diag
for rep in 1 2; do
  for shape in "s1 3x3 64->64 @160" "s2 3x3 128->128 @80"; do
    for ab in 0 8 24 16 6 22 30; do
      echo -n "rep $rep DY_ABLATE=$ab | "
      CB_ONLY="$shape" DY_ABLATE=$ab timeout -k 10 120 $ROOT/tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" | tail -1 | cut -c1-110
    done
  done
done
