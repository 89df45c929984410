#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
CB_CHECK=1 timeout -k 5 500 tools/bin/conv_bench 20 64 > $O/wg4_all.log 2>&1
DY_NO_WGRAD_V4=1 CB_ONLY="256->256 @40" timeout -k 5 100 tools/bin/conv_bench 20 64 > $O/wg4_off.log 2>&1
echo done
