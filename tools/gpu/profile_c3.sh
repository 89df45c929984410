#!/bin/bash
# reduced tools/profile_round.sh for the final tree: C3 one-stream kernel stats (comparable with bench.py's avg_launch_us), the C3 PMC
# traffic passes, the SQ counter pass over the named conv shapes.  Usage (through gpurun): bash tools/gpu/profile_c3.sh <out_dir under gpurun_out/>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STATS="--steps 10 --warmup 3 --no-cpu-baseline"
SHORT="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
SQ="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
step() { echo "[profile_c3] $1 ($(date +%T))"; }
step "kernel stats C3, one stream" &&
DY_WGRAD_STREAM=0 DY_BRANCH_STREAMS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3s_stats" -o c3s -- python3 "$ROOT/bench.py" $STATS > "$OUT/bench_c3_single_stream_under_rocprof.log" 2>&1 &&
step "PMC C3" &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/c3_f" -o f -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c3_f.log" 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/c3_w" -o w -- python3 "$ROOT/bench.py" $SHORT > "$OUT/pmc_c3_w.log" 2>&1 &&
step "SQ counters over the named conv shapes" &&
for shape in "256->256 @40" "128->128 @80" "256->256 @80" "64->64 @160" "512->512 @20" "512->512 @40"; do
  tag=$(echo "$shape" | tr -d ' >@-')
  CB_WARM=3 CB_ONLY="3x3 $shape" timeout -k 10 100 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$OUT/sq_$tag" -o m -- "$ROOT/tools/bin/conv_bench" 3 64 > "$OUT/sq_$tag.log" 2>&1 || exit 1
  python3 "$ROOT/tools/pmc_mfma.py" $(find "$OUT/sq_$tag" -name "*counter_collection.csv" | head -1) "$OUT/sq_$tag.csv" "3x3 $shape B=64" >> "$OUT/mfma_util.log" 2>&1
done &&
cd "$ROOT" &&
( head -1 "$OUT"/sq_25625640.csv; for f in "$OUT"/sq_*.csv; do tail -n +2 "$f"; done ) > "$OUT/conv_mfma_util.csv" &&
python3 tools/pmc_traffic.py $(find "$OUT/c3_f" -name "*counter_collection.csv" | head -1) $(find "$OUT/c3_w" -name "*counter_collection.csv" | head -1) "$OUT/c3_pmc_traffic.json" 5 > "$OUT/pmc_c3_summary.log" 2>&1 &&
cp $(find "$OUT/c3s_stats" -name "*kernel_stats.csv" | head -1) "$OUT/c3_single_stream_kernel_stats.csv" &&
rm -rf "$OUT"/c3s_stats "$OUT"/c3_f "$OUT"/c3_w "$OUT"/sq_*/ &&
step "done" && ls "$OUT"
