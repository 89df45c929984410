#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py -m gpu -q -s -k "l_graph or full_size" > gpurun_out/r2/pytest_lp.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2/pytest_lp.log
grep -n "L graph\|permuted\|eval prediction\|passed\|failed" gpurun_out/r2/pytest_lp.log | cut -c1-700
