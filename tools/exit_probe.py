#!/usr/bin/env python3
"""The floor under any HIP process on this box, and what the PREVIOUS process's exit does to the next one's start:
tools/ubench/hip_startup (hipInit, a stream, a 6.6 MB upload, one kernel, nothing else) timed as a process, runs in a row
with a pause of 0 / 0.25 / 0.5 / 1 / 2 s between them (a process's GPU state is torn down by the kernel after it has
exited; a process that starts meanwhile waits for that inside hipInit).  usage: exit_probe.py"""
import os
import subprocess
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def wall(cmd):
    t = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True)
    return time.perf_counter() - t, p


for pause in (0.0, 0.25, 0.5, 1.0, 2.0, 0.0):
    rows = []
    for i in range(5):
        time.sleep(pause)
        w, p = wall([ROOT + "/tools/ubench/hip_startup"])
        init = [l for l in p.stdout.splitlines() if l.startswith("hipInit")]
        tot = [l for l in p.stdout.splitlines() if l.startswith("total since main")]
        rows.append((w, float(init[-1].split()[-2]) if init else -1, float(tot[-1].split()[-2]) if tot else -1))
    print("pause %.2f s:  process %s s;  hipInit %s ms;  main %s ms" % (
        pause, " ".join("%.3f" % r[0] for r in rows), " ".join("%.0f" % r[1] for r in rows), " ".join("%.0f" % r[2] for r in rows)), flush=True)
