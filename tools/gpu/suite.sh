#!/bin/bash
# full GPU suite + default bench (C3); logs under gpurun_out/r2/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/pytest_gpu.log
tail -4 gpurun_out/r2/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err && tail -1 gpurun_out/r2/bench_c3.json
