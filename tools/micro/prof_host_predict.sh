# kernel + memory-copy trace of Emulator.predict(numpy in, numpy out) on 10 000 points (tools/gpu_host_predict_timing.py): do the
# chunks' device-to-host copies run under the next chunk's kernels?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_hp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_hp -o run -- python3 $R/tools/gpu_host_predict_timing.py > $R/gpurun_out/prof_hp.txt 2> $R/gpurun_out/prof_hp.err
ls $R/gpurun_out/prof_hp/*
cat $R/gpurun_out/prof_hp.txt
