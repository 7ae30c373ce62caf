mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_multi_emulator.py tests/test_gpu_engine.py tests/test_gpu_sampler.py -q -m gpu -x 2>&1 | tail -3
for x in -1 4 -1 4; do python tools/gpu_shard_sim.py 1 --c-only --ball=1e-13 --tune=xcd:$x >> gpurun_out/r3m/xcd_ab.txt 2>gpurun_out/r3m/err.txt; done
grep -h "ranks_sim\|tune" gpurun_out/r3m/xcd_ab.txt | cut -c1-330
for b in 0 1 0 1; do python tools/gpu_multi_emulator.py 1000 --tune=chain_batch:$b >> gpurun_out/r3m/multi.txt 2>>gpurun_out/r3m/err.txt; python tools/gpu_multi_emulator.py 1000 --tune=chain_batch:$b --ball >> gpurun_out/r3m/multi_ball.txt 2>>gpurun_out/r3m/err.txt; done
grep -h "emulators\|tune" gpurun_out/r3m/multi.txt gpurun_out/r3m/multi_ball.txt | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3m/fetch4 -- python3 /root/repo/tools/gpu_shard_sim.py 1 --c-only --ball=1e-13 --tune=xcd:4 > $GRAFT_REPO_ROOT/gpurun_out/r3m/fetch4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $(find gpurun_out/r3m/fetch4 -name '*counter_collection.csv' | head -1) "k_predict<128" 6
