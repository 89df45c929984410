#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout -k 5 120 tools/bin/v4_diag 64 128 80 3 > $O/diag_128_80_b.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 256 40 3 > $O/diag_256_40_c.log 2>&1
CB_CHECK=1 timeout -k 5 400 tools/bin/conv_bench 20 64 > $O/v4_all3.log 2>&1
echo done
