#!/bin/bash
# DIAG build: weight-gradient routing sweep on the narrow-output layers
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
for sel in "256->64 @80" "512->64 @40" "320->128 @160" "64->128 @320" "128->128 @160" "128->256 @160"; do
  echo "== $sel default"; CB_ONLY="$sel" tools/bin/conv_bench 10 64 2>&1 | grep -v "^$" | tail -1
  echo "== $sel wg4 min cout 64, any K"; DY_WG4_MIN_COUT=64 DY_WG4_ANY_K=1 CB_CHECK=1 CB_ONLY="$sel" tools/bin/conv_bench 10 64 2>&1 | grep -v "^$" | tail -2
done > gpurun_out/r2/wg4sweep.log 2>&1
cat gpurun_out/r2/wg4sweep.log | cut -c1-200
