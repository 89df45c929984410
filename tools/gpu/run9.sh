#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_val.py -x -q > $O/pytest_gpu3.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu3.log
tail -15 $O/pytest_gpu3.log
