# the randomised parity soaks of tools/gpu_*_soak.py with this round's seeds; one JSON line each into gpurun_out/soak_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
tag=${1:-r05}; s0=${2:-10}
out=$R/gpurun_out/soak_$tag.txt
: > $out
cd $R
for s in $s0 $((s0 + 1)); do echo "parity $s" >> $out; python3 tools/gpu_parity_soak.py 500 $s 2>/dev/null | tail -1 >> $out; done
for s in $s0 $((s0 + 1)); do echo "posterior $s" >> $out; python3 tools/gpu_posterior_soak.py 500 $s 2>/dev/null | tail -1 >> $out; done
echo "multi_chain $s0" >> $out; python3 tools/gpu_multi_chain_soak.py 300 $s0 2>/dev/null | tail -1 >> $out
echo "sampler $s0" >> $out; python3 tools/gpu_sampler_soak.py 200 $s0 2>/dev/null | tail -1 >> $out
echo "shard $s0" >> $out; python3 tools/gpu_shard_soak.py 300 $s0 2>/dev/null | tail -1 >> $out
cat $out
