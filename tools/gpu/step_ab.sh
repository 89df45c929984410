#!/bin/bash
# same-box A/B of the whole training step between library builds: dedark_yolo_amd/lib/var*/ against the in-tree library, interleaved
# repetitions of the bench command (no CPU baseline, no roofline leg).  Usage: bash tools/gpu/step_ab.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2 3; do
  for v in $(cd dedark_yolo_amd/lib && ls -d var* 2>/dev/null) ""; do
    if [ -n "$v" ]; then export DY_LIB_DIR=$ROOT/dedark_yolo_amd/lib/$v; else unset DY_LIB_DIR; fi
    echo -n "rep $rep lib ${v:-current} | "
    timeout -k 10 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['unit'], b['ms_per_step'], 'ms')"
  done
done
