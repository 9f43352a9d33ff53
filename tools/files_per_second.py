#!/usr/bin/env python3
"""How many FILES per second the CLI's list pipeline sustains when the files are so short that the GPU has next to nothing
to do (0.25 s each = 23 frames): the per-file cost of feeder, stage 1, launch assembly, decoding and the in-order writer.
BASELINE configs[3]'s files average 894 frames: N GPUs at 28 M frames/s each need 31 k x N files per second.
usage: files_per_second.py [n_files] [flags...]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
flags = sys.argv[2:] or ["-F"]
exe = "phnrec_amd/bin/phnrec"
mdir = os.path.join("tests", "golden", "models", bench.HU)
rng = np.random.default_rng(5)
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    sig = np.clip(rng.normal(0, 3000, 2000), -32768, 32767).astype("<i2")
    names = []
    for i in range(n):
        p = os.path.join(td, "s%06d.raw" % i)
        sig.tofile(p)
        names.append(p)
    lst = os.path.join(td, "l.scp")
    open(lst, "w").write("".join(x + "\n" for x in names))
    env = dict(os.environ, PHNREC_STATS="1")
    for rep in range(2):
        r, pr = bench.run_cli(exe, ["-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-g", "1"] + flags, env)
        print(flags, "files/s %.0f  (%.2f M frames/s, wall %.3f s)  host_cpu_s %.3f %s"
              % (n / r["list_wall_s"], r["value"] / 1e6, r["list_wall_s"], r["host_cpu_s"], r["cpu_s_by_stage"]), flush=True)
