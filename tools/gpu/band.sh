#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/band4.txt; : > $out
timeout -k 10 600 tools/bin/conv_bench 30 64 2>&1 | grep -v "^$" >> $out || echo "rc=$?" >> $out
cat $out
timeout -k 10 900 python -m pytest tests/test_gpu_conv_kernels.py -x -q -m gpu 2>&1 | tail -3
