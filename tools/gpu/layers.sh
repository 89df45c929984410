#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 600 python tools/layer_profile.py --model yolov8l.yaml --batch 64 --top 90 > gpurun_out/r2/c3_layer_profile.txt 2>&1
echo "rc=$?"; head -3 gpurun_out/r2/c3_layer_profile.txt
