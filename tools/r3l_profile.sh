mkdir -p gpurun_out/r3l
for x in -1 3 -1 3; do python tools/gpu_shard_sim.py 1 --c-only --ball=1e-13 --tune=xcd:$x >> gpurun_out/r3l/xcd_ab.txt 2>gpurun_out/r3l/err.txt; done
grep -h "ranks_sim\|tune" gpurun_out/r3l/xcd_ab.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3l/bench_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-uniform > $GRAFT_REPO_ROOT/gpurun_out/r3l/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3l/bench_rocprof.err
cd $GRAFT_REPO_ROOT
tools/pmc_passes.sh r3l/pmc_bench "k_predict<128" 6 -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-uniform > gpurun_out/r3l/pmc_bench.log 2>&1
cat gpurun_out/r3l/pmc_bench/summary.jsonl
find gpurun_out/r3l/bench_trace -name "*kernel_stats.csv" | head -2
