#!/bin/bash
# GPU test suite + C3 / C2 bench lines with the current build
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout -k 10 300 python bench.py --model yolov8l.yaml --batch 64 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c3_a.json 2> $O/bench_c3_a.err
DY_NO_CONV_V4=1 timeout -k 10 300 python bench.py --model yolov8l.yaml --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > $O/bench_c3_nov4.json 2> $O/bench_c3_nov4.err
tail -3 $O/pytest_gpu.log
