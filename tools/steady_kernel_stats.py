#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run restricted to its STEADY-STATE launches.

rocprofv3's own --stats averages every launch of the process, including the first ones (code-object load, clock
ramp after idling: ~0.29 ms against 0.195 ms for the bench's kernel), so its average sits above the bench's own
figure for the timed launches.  This tool reads the same run's kernel trace and summarises, per kernel, only the N
launches of the traced command's timed region, in the column layout of rocprofv3's *_kernel_stats.csv.  Which launches
those are: BENCH.json (the traced run's own line) says how many pre-heat and warm-up launches precede them
("roofline.launches_timed": "200 launches after 320 pre-heat + 100 warm-up launches") -- the bench's check launches
BEHIND the timed region (parity, cold-clock figure: ~46, after host-side pauses) must not be counted either.  Without
BENCH.json: the last N launches of every kernel.  Kernels other than the timed one: their last N.

usage: steady_kernel_stats.py TRACE_DIR N [OUT.csv [BENCH.json]]        (TRACE_DIR holds <pid>_kernel_trace.csv)"""
import csv
import glob
import math
import os
import sys


def main():
    trace_dir, n = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    first = None
    if len(sys.argv) > 4:
        import json
        import re
        text = open(sys.argv[4]).read()
        try:                                        # the full record (bench.py --detail-out: one indented JSON document) ...
            rec = json.loads(text)
        except ValueError:                          # ... or a file whose last line is a record (earlier rounds' stdout)
            rec = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
        nums = [int(v) for v in re.findall(r"\d+", rec["roofline"]["launches_timed"])]
        assert nums[0] == n, (nums, n)
        first = sum(nums[1:])                       # launches before the timed region
    files = sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)      # the newest process
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
        timed = first is not None and len(v) >= first + n and "lcrc_fused_kernel" in name
        d = [x[1] for x in v[first:first + n]] if timed else [x[1] for x in v[-n:]] if len(v) > n else [x[1] for x in v]
        window = ("launches %d..%d of %d (the timed region)" % (first + 1, first + n, len(v))) if timed else "last %d launches of %d" % (len(d), len(v))
        avg = sum(d) / len(d)
        sd = math.sqrt(sum((x - avg) ** 2 for x in d) / len(d))
        stats.append((name, len(d), sum(d), avg, min(d), max(d), sd, len(v), window))
    total = sum(s[2] for s in stats) or 1
    for s in sorted(stats, key=lambda s: -s[2]):
        lines.append([s[0], s[1], s[2], "%.3f" % s[3], "%.2f" % (100.0 * s[2] / total), s[4], s[5], "%.3f" % s[6], s[7],
                      s[8]])
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerows(lines)


if __name__ == "__main__":
    main()
