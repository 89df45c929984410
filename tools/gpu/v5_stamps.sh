#!/bin/bash
# conv_v5 / band kernel: s_memtime stamps per phase and K-loop ablations (needs `make -C dedark_yolo_amd/csrc clean && make DIAG=1`
# and `make -C tools`).  DY_ABLATE bits: 1 A taps != 0 out of range, 2 all A out of range, 4 all B out of range, 8 no MFMA,
# 16 no fragment reads, 32 stamps, 64 no A DMA on taps != 0.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/v5_stamps.txt; : > $out
for only in "128->128 @80" "64->64 @160" "1x1 1024->256 @80"; do
  for ab in 32 62; do
    echo "== $only DY_ABLATE=$ab" >> $out
    CB_V5=1 CB_ONLY="$only" DY_ABLATE=$ab timeout -k 10 120 tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" >> $out || echo "rc=$?" >> $out
  done
done
cat $out
