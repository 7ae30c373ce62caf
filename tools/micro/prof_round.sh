# Round profile session (one gpurun call): rocprofv3 kernel trace + stats of the default bench command and of the headline's
# timed region alone, PMC passes on k_predict, kernel trace of the nine-emulator chain.  Outputs under gpurun_out/prof_rN/.
R=$GRAFT_REPO_ROOT
tag=${1:-r05}
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
summ() { f=$(find $1 -name '*kernel_trace.csv' | head -1); python3 $R/tools/kernel_trace_summary.py $f 25 > $2; }
echo "== full bench under rocprofv3"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -o run -- python3 $R/bench.py > $O/full_bench.json 2> $O/full_bench.err
cp $(find $O/full -name '*kernel_stats.csv' | head -1) $O/full_kernel_stats.csv; summ $O/full $O/full_kernel_trace_summary.csv
echo "== timed region alone"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/timed -o run -- python3 $R/bench.py --preheat 0 --no-extras --no-uniform --no-cpu-baseline > $O/timed_bench.json 2> $O/timed_bench.err
cp $(find $O/timed -name '*kernel_stats.csv' | head -1) $O/timed_kernel_stats.csv; summ $O/timed $O/timed_kernel_trace_summary.csv
echo "== nine-emulator chain"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi -o run -- python3 $R/tools/gpu_multi_chain_profile.py 30 > $O/multi.json 2> $O/multi.err
summ $O/multi $O/multi_kernel_trace_summary.csv
cat $O/multi.json; head -8 $O/multi_kernel_trace_summary.csv
echo "== PMC passes on k_predict (timed region: skip the first 7 launches)"; date
cd $R && bash tools/pmc_passes.sh prof_$tag/pmc "k_predict<128" 7 -- python3 bench.py --steps 10 --warmup 3 --preheat 0 --no-cpu-baseline --no-extras --no-uniform > $O/pmc_passes.log 2>&1
rm -rf $O/full/* $O/timed/* $O/multi/* $O/pmc/pass*/ 2>/dev/null      # the raw traces are tens of MB: keep the summaries
tail -3 $O/pmc_passes.log; date
