#!/bin/bash
# Collect hardware counters for one or more kernels of one command, ONE counter group per rocprofv3 run (separate --pmc passes
# with --kernel-trace only: the pool refuses --pmc combined with the sys / hip / hsa trace domains), then print the median per
# dispatch of every counter (tools/pmc_summary.py), one JSON line per (pass, kernel).
#   tools/pmc_passes.sh <out_dir under gpurun_out> <kernel-name-substring[,substring...]> <skip_first_n> -- python3 <script> [args]
# (<script> may be given relative to the repo root: the passes run from /tmp, as the profiling recipe asks.)
# Counter groups: MFMA pipe, wave / wait cycles, LDS, L2 hits, fabric reads, fabric writes, clock.
set -u
out=$1; kerns=$2; skip=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out/$out"
cmd=()
for a in "$@"; do                       # repo-relative paths in the command become absolute before the cd
    if [ -e "$root/$a" ] && [ "${a#/}" = "$a" ]; then cmd+=("$root/$a"); else cmd+=("$a"); fi
done
cd /tmp && export TMPDIR=/tmp
i=0
groups=("SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE")
# PMC_VALU=1 adds the vector / scalar instruction counters (the pair kernels: K build, cross kernel)
if [ "${PMC_VALU:-0}" = "1" ]; then
    groups+=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY")
fi
for grp in "${groups[@]}"; do
    i=$((i + 1))
    d="$root/gpurun_out/$out/pass$i"
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- "${cmd[@]}" > "$d.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$d.log"; continue; }
    f=$(find "$d" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] || continue
    IFS=',' read -ra ks <<< "$kerns"
    for k in "${ks[@]}"; do
        python3 "$root/tools/pmc_summary.py" "$f" "$k" "$skip" "$k" | tee -a "$root/gpurun_out/$out/summary.jsonl"
    done
done
