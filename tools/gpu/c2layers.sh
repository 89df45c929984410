#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 600 python tools/layer_profile.py --top 70 > gpurun_out/r2/c2_layer_profile.txt 2>&1
echo "rc=$?"
