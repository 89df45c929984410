#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of the two bench commands (C3 = the default, C2), separate PMC passes
# (FETCH_SIZE / WRITE_SIZE) of the same commands, the SQ counter pass over the named conv shapes, the plain bench lines.
# Usage (from the repo root, through gpurun): bash tools/profile_round.sh <out_dir under gpurun_out/>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C2="--model yolov8n-lowlight.yaml --batch 32"
C5="--imgsz 1280 --batch 16 --dtype fp16"
STATS="--steps 10 --warmup 3 --no-cpu-baseline"
SHORT="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
step() { echo "[profile_round] $1 ($(date +%T))"; }
step "kernel stats C3" &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3_stats" -o c3 -- python3 "$ROOT/bench.py" $STATS > "$OUT/bench_c3_under_rocprof.log" 2>&1 &&
step "kernel stats C3, one stream (the schedule bench.py's roofline leg times: averages comparable with its avg_launch_us)" &&
DY_WGRAD_STREAM=0 DY_BRANCH_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3s_stats" -o c3s -- python3 "$ROOT/bench.py" $STATS > "$OUT/bench_c3_single_stream_under_rocprof.log" 2>&1 &&
step "kernel stats C2" &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2_stats" -o c2 -- python3 "$ROOT/bench.py" $C2 $STATS > "$OUT/bench_c2_under_rocprof.log" 2>&1 &&
step "PMC C3" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c3_f" -o f -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c3_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c3_w" -o w -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c3_w.log" 2>&1 &&
step "kernel stats + PMC C5 (BASELINE configs[4]: the bandwidth-regime config, rocprof GB/s vs peak)" &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_stats" -o c5 -- python3 "$ROOT/bench.py" $C5 $STATS > "$OUT/bench_c5_under_rocprof.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c5_f" -o f -- python3 "$ROOT/bench.py" $C5 $SHORT > "$OUT/pmc_c5_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c5_w" -o w -- python3 "$ROOT/bench.py" $C5 $SHORT > "$OUT/pmc_c5_w.log" 2>&1 &&
python3 "$ROOT/tools/pmc_traffic.py" $(find "$OUT/c5_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c5_w" -name "*counter_collection.csv" | head -1) "$OUT/c5_pmc_traffic.json" 5 > "$OUT/pmc_c5_summary.log" 2>&1 &&
cp $(find "$OUT/c5_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c5_kernel_stats.csv" &&
rm -rf "$OUT"/c5_stats "$OUT"/c5_f "$OUT"/c5_w &&
step "PMC C2" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c2_f" -o f -- python3 "$ROOT/bench.py" $C2 $SHORT > "$OUT/pmc_c2_f.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c2_w" -o w -- python3 "$ROOT/bench.py" $C2 $SHORT > "$OUT/pmc_c2_w.log" 2>&1 &&
step "SQ counters over the named conv shapes" &&
for shape in "256->256 @40" "128->128 @80" "256->256 @80" "64->64 @160" "512->512 @20" "512->512 @40"; do
  tag=$(echo "$shape" | tr -d ' >@-')
  CB_WARM=3 CB_ONLY="3x3 $shape" rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$OUT/sq_$tag" -o m -- "$ROOT/tools/bin/conv_bench" 3 64 > "$OUT/sq_$tag.log" 2>&1 || exit 1
  python3 "$ROOT/tools/pmc_mfma.py" $(find "$OUT/sq_$tag" -name "*counter_collection.csv" | head -1) "$OUT/sq_$tag.csv" "3x3 $shape B=64" >> "$OUT/mfma_util.log" 2>&1
done &&
cd "$ROOT" &&
( head -1 "$OUT"/sq_25625640.csv; for f in "$OUT"/sq_*.csv; do tail -n +2 "$f"; done ) > "$OUT/conv_mfma_util.csv" &&
python3 tools/pmc_traffic.py $(find "$OUT/c2_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c2_w" -name "*counter_collection.csv" | head -1) "$OUT/c2_pmc_traffic.json" 5 > "$OUT/pmc_c2_summary.log" 2>&1 &&
python3 tools/pmc_traffic.py $(find "$OUT/c3_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c3_w" -name "*counter_collection.csv" | head -1) "$OUT/c3_pmc_traffic.json" 5 > "$OUT/pmc_c3_summary.log" 2>&1 &&
cp $(find "$OUT/c2_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c2_kernel_stats.csv" &&
cp $(find "$OUT/c3_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c3_kernel_stats.csv" &&
cp $(find "$OUT/c3s_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c3_single_stream_kernel_stats.csv" &&
rm -rf "$OUT"/c2_stats "$OUT"/c3_stats "$OUT"/c3s_stats "$OUT"/c2_f "$OUT"/c2_w "$OUT"/c3_f "$OUT"/c3_w "$OUT"/sq_*/ &&
step "plain bench lines (with this run's PMC summaries as their traffic source)" &&
cp "$OUT/c3_pmc_traffic.json" profiles/r03_c3_pmc_traffic.json && cp "$OUT/c2_pmc_traffic.json" profiles/r03_c2_pmc_traffic.json &&
python3 bench.py > "$OUT/bench_c3.json.log" 2> "$OUT/bench_c3.err" &&
python3 bench.py $C2 --steps 100 --warmup 20 > "$OUT/bench_c2.json.log" 2> "$OUT/bench_c2.err" &&
cp "$OUT/c5_pmc_traffic.json" profiles/r03_c5_pmc_traffic.json && python3 bench.py $C5 > "$OUT/bench_c5.json.log" 2> "$OUT/bench_c5.err"
echo "profile_round exit $?"
ls -la "$OUT"
