#!/bin/bash
# Round profile session (one gpurun call): rocprofv3 kernel trace + stats of the default bench command and of the headline's
# timed region alone, the same for the int8 variant (bench.py --sliced), PMC passes on both predict kernels.  Outputs under
# gpurun_out/prof_<tag>/ (scratch): copy the summaries you quote into profiles/.   usage: tools/prof_round.sh [tag=r06]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r06}
O=$R/gpurun_out/prof_$tag
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
summ() { f=$(find "$1" -name '*kernel_trace.csv' | head -1); python3 "$R/tools/kernel_trace_summary.py" "$f" 25 > "$2"; }
echo "== full bench under rocprofv3"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/full" -o run -- python3 "$R/bench.py" > "$O/full_bench.json" 2> "$O/full_bench.err"
cp "$(find "$O/full" -name '*kernel_stats.csv' | head -1)" "$O/full_kernel_stats.csv"; summ "$O/full" "$O/full_kernel_trace_summary.csv"
echo "== timed region alone"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/timed" -o run -- python3 "$R/bench.py" --preheat 0 --no-extras --no-uniform --no-cpu-baseline > "$O/timed_bench.json" 2> "$O/timed_bench.err"
cp "$(find "$O/timed" -name '*kernel_stats.csv' | head -1)" "$O/timed_kernel_stats.csv"; summ "$O/timed" "$O/timed_kernel_trace_summary.csv"
echo "== timed region alone, int8 variant"; date
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/timed8" -o run -- python3 "$R/bench.py" --sliced --preheat 0 --no-extras --no-uniform --no-cpu-baseline > "$O/timed_sliced_bench.json" 2> "$O/timed_sliced_bench.err"
cp "$(find "$O/timed8" -name '*kernel_stats.csv' | head -1)" "$O/timed_sliced_kernel_stats.csv"; summ "$O/timed8" "$O/timed_sliced_kernel_trace_summary.csv"
echo "== PMC passes on k_predict (timed region: skip the first 7 launches)"; date
cd "$R" && bash tools/pmc_passes.sh prof_$tag/pmc "k_predict<128" 7 -- python3 bench.py --steps 10 --warmup 3 --preheat 0 --no-cpu-baseline --no-extras --no-uniform > "$O/pmc_passes.log" 2>&1
echo "== PMC passes on k_predict_sliced"; date
cd "$R" && bash tools/pmc_passes.sh prof_$tag/pmc8 "k_predict_sliced" 7 -- python3 bench.py --sliced --steps 10 --warmup 3 --preheat 0 --no-cpu-baseline --no-extras --no-uniform > "$O/pmc8_passes.log" 2>&1
rm -rf "$O"/full/* "$O"/timed/* "$O"/timed8/* "$O"/pmc/pass*/ "$O"/pmc8/pass*/ 2>/dev/null      # the raw traces are tens of MB: keep the summaries
tail -3 "$O/pmc_passes.log" "$O/pmc8_passes.log"; date
