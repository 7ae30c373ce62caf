#!/bin/bash
# Collect hardware counters for one kernel of one command, ONE counter group per rocprofv3 run (separate --pmc passes with
# --kernel-trace only: the pool refuses --pmc combined with the sys / hip / hsa trace domains), then print the median per
# dispatch of every counter (tools/pmc_summary.py).
#   tools/pmc_passes.sh <out_dir under gpurun_out> <kernel-name-substring> <skip_first_n> -- python3 <script> [args]
# Counter groups: MFMA pipe, wave / wait cycles, LDS, L2 hits, fabric reads, fabric writes, clock.
set -u
out=$1; kern=$2; skip=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out/$out"
cd /tmp && export TMPDIR=/tmp
i=0
# PMC_VALU=1 adds the vector / scalar instruction counters (the pair kernels: K build, cross kernel)
extra=()
[ "${PMC_VALU:-0}" = "1" ] && extra=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY")
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "${extra[@]}"; do
    i=$((i + 1))
    d="$root/gpurun_out/$out/pass$i"
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -- "$@" > "$d.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$d.log"; continue; }
    f=$(find "$d" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 "$root/tools/pmc_summary.py" "$f" "$kern" "$skip" | tee -a "$root/gpurun_out/$out/summary.jsonl"
done
