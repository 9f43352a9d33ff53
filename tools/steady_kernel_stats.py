#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run restricted to its STEADY-STATE launches.

rocprofv3's own --stats averages every launch of the process, including the first ones (code-object load, clock
ramp after idling: ~0.29 ms against 0.195 ms for the bench's kernel), so its average sits above the bench's own
figure for the timed launches.  This tool reads the same run's kernel trace and summarises, per kernel, only the last
N launches (N = the timed steps of the traced command), in the column layout of rocprofv3's *_kernel_stats.csv.

usage: steady_kernel_stats.py TRACE_DIR N [OUT.csv]        (TRACE_DIR holds <pid>_kernel_trace.csv)"""
import csv
import glob
import math
import os
import sys


def main():
    trace_dir, n = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    files = sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + trace_dir)
    rows = list(csv.DictReader(open(files[0])))
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    lines = [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "LaunchesInRun", "Window"]]
    stats = []
    for name, v in by.items():
        v.sort()
        d = [x[1] for x in v[-n:]] if len(v) > n else [x[1] for x in v]
        avg = sum(d) / len(d)
        sd = math.sqrt(sum((x - avg) ** 2 for x in d) / len(d))
        stats.append((name, len(d), sum(d), avg, min(d), max(d), sd, len(v)))
    total = sum(s[2] for s in stats) or 1
    for s in sorted(stats, key=lambda s: -s[2]):
        lines.append([s[0], s[1], s[2], "%.3f" % s[3], "%.2f" % (100.0 * s[2] / total), s[4], s[5], "%.3f" % s[6], s[7],
                      "last %d launches of %d" % (s[1], s[7])])
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerows(lines)


if __name__ == "__main__":
    main()
