cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o run -- python3 $R/bench.py --preheat 0 --no-extras --no-uniform --no-cpu-baseline > $R/gpurun_out/prof_final_bench.json 2> $R/gpurun_out/prof_final.err
ls -R $R/gpurun_out/prof_final | head -20
