#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c3_b.json 2> $O/bench_c3_b.err
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu2.log
timeout -k 10 300 python tools/layer_profile.py --model yolov8l.yaml --batch 64 > $O/c3_layer_profile.txt 2>&1
tail -3 $O/pytest_gpu2.log
