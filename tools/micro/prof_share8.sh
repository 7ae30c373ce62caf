# One rank's share of an 8-way walker-sharded step (tools/gpu_shard_sim.py 8) under rocprofv3: per-kernel medians of the half-step's
# launches -> gpurun_out/prof_share8_summary.csv (+ the tool's own line).  usage: bash tools/micro/prof_share8.sh [extra tool args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_share8"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_share8" -o run -- python3 "$R/tools/gpu_shard_sim.py" 8 --c-only --ball=1e-13 "$@" > "$R/gpurun_out/prof_share8.json" 2> "$R/gpurun_out/prof_share8.err"
python3 "$R/tools/kernel_trace_summary.py" "$(find "$R/gpurun_out/prof_share8" -name '*kernel_trace.csv' | head -1)" 16 > "$R/gpurun_out/prof_share8_summary.csv"
python3 "$R/tools/trace_timeline.py" "$(find "$R/gpurun_out/prof_share8" -name '*kernel_trace.csv' | head -1)" 400 24 > "$R/gpurun_out/prof_share8_timeline.txt"
cat "$R/gpurun_out/prof_share8.json" "$R/gpurun_out/prof_share8_summary.csv" "$R/gpurun_out/prof_share8_timeline.txt"
rm -rf "$R/gpurun_out/prof_share8"
