#!/usr/bin/env python3
"""Per-kernel duration and the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV (in-order stream):
    python tools/trace_timeline.py <kernel_trace.csv> [last_n_kernels]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
    rows.sort()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
    rows = rows[-n:]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for i in range(1, len(rows)):
        s, e, k = rows[i]
        dur[k] += e - s
        gap[k] += max(0, s - rows[i - 1][1])
        cnt[k] += 1
    span = rows[-1][1] - rows[1][0]
    print("%-72s %7s %9s %9s %7s" % ("kernel", "count", "dur_us", "gap_us", "share"))
    for k in sorted(cnt, key=lambda k: -(dur[k] + gap[k])):
        print("%-72s %7d %9.2f %9.2f %6.1f%%" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3,
                                                 100.0 * (dur[k] + gap[k]) / span))
    print("span_us %.1f  busy %.1f%%" % (span / 1e3, 100.0 * sum(dur.values()) / span))
    if len(sys.argv) > 3:                      # the raw sequence of the last few launches
        m = int(sys.argv[3])
        for i in range(len(rows) - m, len(rows)):
            s, e, k = rows[i]
            print("%9.2f +%8.2f  gap %7.2f  %s" % ((s - rows[-m][0]) / 1e3, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3, k))


if __name__ == "__main__":
    main()
