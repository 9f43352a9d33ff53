#!/usr/bin/env python3
"""Where the first milliseconds of a list go: `phnrec -l` on BASELINE configs[3]'s list (HU, 10 000 files) with
PHNREC_TRACE_PIPELINE=1 (the workers' steps, time-stamped) and LCRC_TRACE_SLOW_US (library calls that took longer than
that, with the step that took it), followed by the process wall clock of each mode.
usage: pipeline_trace.py [n_files]      env TRACE_CHARS: how much of the trace to print; TRACE_FLAGS: "-F;-F -D;" (modes, ';'-separated)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench

exe = "phnrec_amd/bin/phnrec"
mdir = os.path.join("tests", "golden", "models", bench.HU)
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, n_files)
    for flags in [f.split() for f in os.environ.get("TRACE_FLAGS", "-F;-F -D;").split(";")]:
        for rep in range(2):
            env = dict(os.environ, PHNREC_STATS="1", PHNREC_TRACE_PIPELINE="1", LCRC_TRACE_SLOW_US=os.environ.get("LCRC_TRACE_SLOW_US", "5000"))
            if not flags:
                env["PHNREC_NO_AUTO_E"] = "1"          # (the host front-end: a list of this length would take -F by itself)
            p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-g", "1"] + flags, env=env, capture_output=True, text=True)
        print("=====", flags)
        print(p.stderr[:int(os.environ.get("TRACE_CHARS", "9000"))])
        print([l for l in p.stderr.splitlines() if l.startswith("phnrec:")])
    if os.environ.get("TRACE_ONLY"):
        sys.exit(0)
    print("===== process wall clock, best of 5 (list wall in brackets)")
    for flags in (["-F"], ["-F", "-D"], ["-E"], []):
        for extra in ({},):
            best = None
            for rep in range(5):
                env = dict(os.environ, PHNREC_STATS="1", **extra)
                if not flags:
                    env["PHNREC_NO_AUTO_E"] = "1"      # (host): a list of this length would take -F by itself
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-g", "1"] + flags, env=env, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                st = [l for l in p.stderr.splitlines() if l.startswith("phnrec:")][-1]
                lw = float(st.split("wall_s=")[1].split()[0])
                su = float(st.split("setup_s=")[1].split()[0])
                if best is None or dt < best[0]:
                    best = (dt, lw, su)
            print("%-8s %-22s process %.3f s  [list %.3f s, set-up %.3f s]" % (" ".join(flags) or "(host)", extra or "", best[0], best[1], best[2]), flush=True)
