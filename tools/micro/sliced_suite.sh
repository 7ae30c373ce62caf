# The whole GPU suite, and the randomised soaks, with the int8 predict kernel forced on for every engine (GPB_PREDICT_SLICED=1):
# what passes and what misses the fp64 path's bars, with magnitudes -> gpurun_out/sliced_suite.txt (copy to profiles/rNN_sliced_suite.txt)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$R/gpurun_out/sliced_suite.txt
cd "$R"
{
echo "# GPB_PREDICT_SLICED=1 python3 -m pytest tests -m gpu -q   (every GPEngine of the process: gpb_ctx_option 51 = 1)"
GPB_PREDICT_SLICED=1 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_ranks.py 2>&1 | grep -E "^FAILED|passed|failed|^E  +assert [0-9.e-]+ < " | sed 's/ - .*//'
echo
echo "# the soak tools under the same setting (one JSON line per violation, then the summary)"
for t in "gpu_parity_soak.py 300 61" "gpu_posterior_soak.py 300 61" "gpu_multi_chain_soak.py 100 61" "gpu_sampler_soak.py 100 61" "gpu_shard_soak.py 100 61"; do
  echo "## tools/$t"
  GPB_PREDICT_SLICED=1 python3 tools/$t 2>/dev/null | grep -v '"done"' | tail -12
done
} > "$out" 2>&1
cat "$out"
