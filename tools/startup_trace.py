#!/usr/bin/env python3
"""Where the first 0.3 s of a list run go: the CLI's pipeline time line (PHNREC_TRACE_PIPELINE) and the library's
start-up phases (LCRC_TRACE_STARTUP) of `phnrec -l <configs[3] list> [flags]`.  usage: startup_trace.py [n_files] [flags...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
flags = sys.argv[2:]
exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
mdir = os.path.join(ROOT, "tests", "golden", "models", "PHN_HU_SPDAT_LCRC_N1500")
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, n_files)
    env = dict(os.environ, PHNREC_STATS="1", PHNREC_TRACE_PIPELINE="1", LCRC_TRACE_STARTUP="1")
    for k in range(3):
        p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf")] + flags, env=env, capture_output=True, text=True)
        if k == 0:
            continue
        lines = p.stderr.splitlines()
        keep = [l for l in lines if "ctx:" in l or "worker" in l or "lcrc" in l.lower() or l.startswith("phnrec:")]
        firsts = [l for l in lines if "took launch" in l][:6] + [l for l in lines if "decoded" in l][:3]
        print("---- run %d: %s" % (k, " ".join(flags)))
        print("\n".join(sorted(set(keep + firsts), key=lambda l: lines.index(l))[:70]))
