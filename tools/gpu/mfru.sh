#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lowprec.py tests/test_gpu_trainer.py -m gpu -q -s -k "scconv or mfru or yolov8_3 or l_graph or validates_on_ema or initial_bn" > gpurun_out/r2/pytest_mfru.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2/pytest_mfru.log
grep -n "passed\|failed\|^E  \|Error\|g2_scconv\|g2_mfru\|yolov8-3\|L graph" gpurun_out/r2/pytest_mfru.log | cut -c1-400 | tail -40
