#!/bin/bash
# final tree: the default bench line + rocprofv3 kernel stats of the same command (no CPU baseline under the profiler)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3/final3; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-200
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c3 -- python3 $R/bench.py --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_c3_kernel_stats.csv && head -4 $O/bench_c3_kernel_stats.csv | cut -c1-150
tail -1 $O/prof.log | cut -c1-200
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null; true
