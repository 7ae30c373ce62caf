# rocprofv3 kernel trace of the nine-emulator chain's step loop (tools/gpu_multi_chain_profile.py): per-kernel medians
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
rm -rf $R/gpurun_out/prof_multi
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_multi -o run -- python3 $R/tools/gpu_multi_chain_profile.py 30 "$@" > $R/gpurun_out/prof_multi.json 2> $R/gpurun_out/prof_multi.err
python3 $R/tools/kernel_trace_summary.py $(find $R/gpurun_out/prof_multi -name "*kernel_trace.csv" | head -1) 12 > $R/gpurun_out/prof_multi_summary.csv
cat $R/gpurun_out/prof_multi.json $R/gpurun_out/prof_multi_summary.csv
