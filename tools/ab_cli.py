#!/usr/bin/env python3
"""A/B of the CLI over two builds of the library on the same box, alternating: the library file next to the CLI is swapped in
place (and restored at the end).  Per build and round: the wall clock of a one-file run and the list loop's figures on the
configs[3] list with -F.        usage: ab_cli.py LIB_A LIB_B [rounds]      ("-" = the library as it stands)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = os.path.join(ROOT, "phnrec_amd", "lib", "libphnrec_lcrc.so")
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", bench.HU)
    raw = os.path.join(ROOT, "tests", "golden", "test.raw")
    keep = "/tmp/ab_cli_keep.so"
    shutil.copyfile(lib, keep)
    builds = [keep if a == "-" else os.path.join(ROOT, a) for a in sys.argv[1:3]]
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            lst, names, frames = bench.synthetic_list(td, 10000)
            env = dict(os.environ, PHNREC_STATS="1")
            for r in range(rounds):
                for tag, b in zip("AB", builds):
                    shutil.copyfile(b, lib)
                    t0 = time.perf_counter()
                    subprocess.run([exe, "-c", mdir, "-i", raw, "-o", os.path.join(td, "x.rec")], check=True, capture_output=True)
                    one = time.perf_counter() - t0
                    p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-F"], env=env, capture_output=True, text=True)
                    line = [l for l in p.stderr.splitlines() if l.startswith("phnrec:")][-1]
                    print("%s round %d  one file %.3f s   %s" % (tag, r, one, line[line.index("wall_s"):line.index("host_cpu_s")]), flush=True)
    finally:
        shutil.copyfile(keep, lib)


if __name__ == "__main__":
    main()
