#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
CB_CHECK=1 timeout -k 5 500 tools/bin/conv_bench 20 64 > $O/epi_all.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 256 40 3 > $O/diag_256_40_d.log 2>&1
timeout -k 5 120 tools/bin/v4_diag 64 128 80 3 > $O/diag_128_80_d.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_conv_kernels.py tests/test_gpu_parity.py -x -q > $O/pytest_conv.log 2>&1; echo "rc=$?" >> $O/pytest_conv.log
echo done
