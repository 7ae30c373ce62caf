# rocprofv3 kernel trace of ONE rank's step loop of a BASELINE config (tools/gpu_shard_sim.py 1 --c-only --cfg=N, burnt-in): the
# launch sequence of the last half-steps with every kernel's duration and the idle gap in front of it.   usage: prof_cfg_step.sh 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-3}
rm -rf $R/gpurun_out/prof_cfg$CFG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cfg$CFG -o run -- python3 $R/tools/gpu_shard_sim.py 1 --c-only --cfg=$CFG --ball=1e-13 > $R/gpurun_out/prof_cfg$CFG.json 2> $R/gpurun_out/prof_cfg$CFG.err
T=$(find $R/gpurun_out/prof_cfg$CFG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 400 24 > $R/gpurun_out/prof_cfg${CFG}_timeline.txt
cat $R/gpurun_out/prof_cfg$CFG.json $R/gpurun_out/prof_cfg${CFG}_timeline.txt
