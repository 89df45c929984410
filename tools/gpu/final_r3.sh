#!/bin/bash
# end-of-round check: smoke(), full GPU suite, the C3 / C5 / C2 bench lines, rocprofv3 kernel stats of the C3 command
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.json | cut -c1-400
timeout -k 10 400 python bench.py --imgsz 1280 --batch 16 --dtype fp16 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.json | cut -c1-200
timeout -k 10 400 python bench.py --model yolov8n-lowlight.yaml --batch 32 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-200
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c3 -- python3 $R/bench.py --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_c3_kernel_stats.csv && head -12 $O/bench_c3_kernel_stats.csv | cut -c1-150
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null; true
