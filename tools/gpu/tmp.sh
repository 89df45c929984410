#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_conv_kernels.py tests/test_gpu_parity.py tests/test_gpu_lowprec.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 600 python bench.py --model yolov8n-lowlight.yaml --batch 32 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
