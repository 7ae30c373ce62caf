#!/usr/bin/env python3
"""Per-kernel launch count, mean, median, min and max duration from a rocprofv3 --kernel-trace CSV:
    python tools/kernel_trace_summary.py <kernel_trace.csv> [top_n]
(rocprofv3 --stats gives the mean only; for bench.py the mean of k_predict includes the one 4096-row evaluation of the
starting positions, the median is the timed region's launch.)"""
import csv
import statistics
import sys
from collections import defaultdict

dur = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("kernel,calls,total_us,mean_us,median_us,min_us,max_us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (k, len(v), sum(v), statistics.mean(v), statistics.median(v), min(v), max(v)))
