# rocprofv3 kernel trace of LML + gradient evaluations: bash tools/micro/prof_lml.sh N reps P  -> gpurun_out/prof_lml_<N>_<P>_summary.csv
cd /tmp && export TMPDIR=/tmp
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
N=${1:-1024}; reps=${2:-20}; P=${3:-10}
D=$R/gpurun_out/prof_lml_${N}_${P}
rm -rf $D
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -o run -- python3 $R/tools/gpu_lml_profile.py $N $reps $P > ${D}.txt 2>&1
python3 $R/tools/kernel_trace_summary.py $(find $D -name "*kernel_trace.csv" | head -1) 20 > ${D}_summary.csv
grep ms_per_lml ${D}.txt; cat ${D}_summary.csv
python3 - <<PY
import csv,glob
f=glob.glob("$D/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)/1e3
print("kernel time summed over the run: %.1f us over %d evaluations (+1 warm-up) = %.1f us per evaluation" % (tot, $reps, tot/($reps+1)))
PY
rm -rf $D
