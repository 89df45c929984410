#!/bin/bash
# new low-precision / routed / scaler tests, then the C5 bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_lowprec.py tests/test_gpu_conv_kernels.py tests/test_gpu_trainer.py -m gpu -q -s -k "low_precision or f16 or fp16 or large_tile or full_size or ciou or reduces" > gpurun_out/r2/pytest_fp16.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2/pytest_fp16.log
tail -5 gpurun_out/r2/pytest_fp16.log
timeout -k 10 500 python bench.py --imgsz 1280 --batch 16 --dtype fp16 --steps 10 --warmup 3 > gpurun_out/r2/bench_c5.json 2> gpurun_out/r2/bench_c5.err
echo "bench rc=$?"; tail -1 gpurun_out/r2/bench_c5.json | cut -c1-900; tail -3 gpurun_out/r2/bench_c5.err
