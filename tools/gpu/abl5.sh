#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/abl5b.txt; : > $out
for only in "128->128 @80" "64->64 @160"; do
  for ab in 32 62 94; do
    echo "== $only DY_ABLATE=$ab" >> $out
    CB_V5=1 CB_ONLY="$only" DY_ABLATE=$ab timeout -k 10 120 tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" >> $out || echo "rc=$?" >> $out
  done
done
cat $out
