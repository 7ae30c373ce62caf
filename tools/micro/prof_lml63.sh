cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_lml63
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lml63 -o run -- python3 $R/tools/gpu_lml_profile.py 1000 20 63 > $R/gpurun_out/prof_lml63.txt 2>&1
python3 $R/tools/kernel_trace_summary.py $(find $R/gpurun_out/prof_lml63 -name "*kernel_trace.csv" | head -1) 16 > $R/gpurun_out/prof_lml63_summary.csv
tail -1 $R/gpurun_out/prof_lml63.txt; cat $R/gpurun_out/prof_lml63_summary.csv
rm -rf $R/gpurun_out/prof_lml63
