#!/usr/bin/env python3
"""A/B of the CLI's three list modes (host front-end, -F, -F -D) over two builds of the library on the same box, alternating:
phnrec_amd/lib/ab/libbase.so (tools/build_ab_lib.sh <ref> base) against the library as it stands; the file next to the
CLI is swapped in place and restored.  usage: ab_cli_modes.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench, os, subprocess, tempfile, shutil, sys
exe = "phnrec_amd/bin/phnrec"
lib = "phnrec_amd/lib/libphnrec_lcrc.so"
shutil.copyfile(lib, "/tmp/new.so")
mdir = os.path.join("tests", "golden", "models", bench.HU)
try:
  with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, 10000)
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
      for flags in ([], ["-F"], ["-F", "-D"]):
        row = []
        for tag, so in (("A", "phnrec_amd/lib/ab/libbase.so"), ("B", "/tmp/new.so")):
            shutil.copyfile(so, lib)
            env = dict(os.environ, PHNREC_STATS="1")
            p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o%s.mlf" % tag)] + flags, env=env, capture_output=True, text=True)
            line = [l for l in p.stderr.splitlines() if l.startswith("phnrec:")][-1]
            row.append("%s %.1f M" % (tag, float(line.split("frames_per_s=")[1].split()[0]) / 1e6))
        print(rep, flags, " ".join(row), "same MLF", open(os.path.join(td, "oA.mlf")).read() == open(os.path.join(td, "oB.mlf")).read(), flush=True)
finally:
    shutil.copyfile("/tmp/new.so", lib)
