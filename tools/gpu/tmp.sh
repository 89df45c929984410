#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_conv_kernels.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 600 python bench.py --model yolov8n-lowlight.yaml --batch 32 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 600 python tools/layer_profile.py --model yolov8l.yaml --batch 64 --top 200 > gpurun_out/c3_layers.txt 2>&1; grep "3->64\|total event" gpurun_out/c3_layers.txt | cut -c1-200
