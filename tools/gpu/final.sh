#!/bin/bash
# end-of-round check: build() + smoke(), full GPU suite, the three bench lines
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/pytest_gpu.log
tail -4 gpurun_out/r2/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/r2/bench_c3.json 2> gpurun_out/r2/bench_c3.err; tail -1 gpurun_out/r2/bench_c3.json | cut -c1-260
timeout -k 10 400 python bench.py --imgsz 1280 --batch 16 --dtype fp16 --no-cpu-baseline > gpurun_out/r2/bench_c5.json 2> gpurun_out/r2/bench_c5.err; tail -1 gpurun_out/r2/bench_c5.json | cut -c1-260
