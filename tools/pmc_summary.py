#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for one kernel: median over its dispatches of every counter column.

    python tools/pmc_summary.py <counter_collection.csv> <kernel-name-substring> [skip_first_n] [label]
Prints a JSON object {counter: median, ..., "dispatches": n} (with "kernel": label when given; "sum_<counter>": the counter
summed over the kept dispatches, for kernels whose launches differ in size).  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports
them; the gfx950 correction (FETCH_SIZE counts 128-byte requests at 64 bytes: x2) is applied by the caller."""
import csv
import json
import statistics
import sys
from collections import defaultdict

path, name = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per_disp = defaultdict(dict)
with open(path) as f:
    for r in csv.DictReader(f):
        if name not in r["Kernel_Name"]:
            continue
        per_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
disp = [per_disp[k] for k in sorted(per_disp)][skip:]
out = {"dispatches": len(disp)}
if len(sys.argv) > 4:
    out["kernel"] = sys.argv[4]
for c in sorted({c for d in disp for c in d}):
    vals = [d[c] for d in disp if c in d]
    out[c] = statistics.median(vals)
    out["sum_" + c] = sum(vals)
print(json.dumps(out))
