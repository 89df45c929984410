#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of the two bench commands + separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# Usage (from the repo root, through gpurun): bash tools/profile_round.sh <out_dir under gpurun_out/>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C3="--model yolov8l.yaml --batch 64 --steps 10 --warmup 3"
SHORT="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2_stats" -o c2 -- python3 "$ROOT/bench.py" > "$OUT/bench_c2_under_rocprof.log" 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3_stats" -o c3 -- python3 "$ROOT/bench.py" $C3 --no-cpu-baseline > "$OUT/bench_c3_under_rocprof.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2_f" -o f -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c2_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2_w" -o w -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c2_w.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c3_f" -o f -- python3 "$ROOT/bench.py" --model yolov8l.yaml --batch 64 $SHORT > "$OUT/pmc_c3_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c3_w" -o w -- python3 "$ROOT/bench.py" --model yolov8l.yaml --batch 64 $SHORT > "$OUT/pmc_c3_w.log" 2>&1 &&
cd "$ROOT" &&
python3 tools/pmc_traffic.py $(find "$OUT/c2_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c2_w" -name "*counter_collection.csv" | head -1) "$OUT/c2_pmc_traffic.json" > "$OUT/pmc_c2_summary.log" 2>&1 &&
python3 tools/pmc_traffic.py $(find "$OUT/c3_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c3_w" -name "*counter_collection.csv" | head -1) "$OUT/c3_pmc_traffic.json" > "$OUT/pmc_c3_summary.log" 2>&1 &&
cp $(find "$OUT/c2_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c2_kernel_stats.csv" &&
cp $(find "$OUT/c3_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c3_kernel_stats.csv" &&
rm -rf "$OUT"/c2_stats "$OUT"/c3_stats "$OUT"/c2_f "$OUT"/c2_w "$OUT"/c3_f "$OUT"/c3_w &&
python3 bench.py > "$OUT/bench_c2.json.log" 2>&1 &&
python3 bench.py $C3 > "$OUT/bench_c3.json.log" 2>&1
echo "profile_round exit $?"
ls -la "$OUT"
