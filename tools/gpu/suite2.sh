#!/bin/bash
# conv kernel tests + C3 / C2 / C5 bench lines
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q > gpurun_out/r2/pytest_conv.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/pytest_conv.log
tail -3 gpurun_out/r2/pytest_conv.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err && tail -1 gpurun_out/r2/bench_c3.json | cut -c1-330
timeout -k 10 400 python bench.py --model yolov8n-lowlight.yaml --batch 32 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r2/bench_c2.json 2> gpurun_out/r2/bench_c2.err && tail -1 gpurun_out/r2/bench_c2.json | cut -c1-330
