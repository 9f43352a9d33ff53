#!/usr/bin/env python3
"""A/B of two builds of the library on configs[3] AS A PROCESS (exec to exit), alternating, each run 0.3 s after the previous
one's exit: the start and the end of a run are what the variants differ in (tools/ab_cli_list.py compares list loops of
seconds).  The variant is a copy named libphnrec_lcrc.so in its own directory, selected through LD_LIBRARY_PATH.
    ab_process.py LIB_A LIB_B [rounds = 7] [flags ...]          ("-" = the library as it stands)"""
import os
import shutil
import statistics
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    flags = sys.argv[4:]
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", bench.HU)
    libdir = tempfile.mkdtemp(prefix="ab_process_")
    dirs = {}
    for tag, x in zip("AB", (a, b)):
        d = os.path.join(libdir, tag)
        os.mkdir(d)
        shutil.copyfile(os.path.join(ROOT, "phnrec_amd", "lib", "libphnrec_lcrc.so") if x == "-" else os.path.join(ROOT, x),
                        os.path.join(d, "libphnrec_lcrc.so"))
        dirs[tag] = d
    res = {"A": [], "B": []}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            lst, names, frames = bench.synthetic_list(td, 10000)
            mlfs = {}
            for r in range(rounds + 1):
                for tag in "AB":
                    env = dict(os.environ, PHNREC_STATS="1")
                    env["LD_LIBRARY_PATH"] = dirs[tag] + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
                    mlf = os.path.join(td, tag + ".mlf")
                    v, _pr = bench.run_cli(exe, ["-c", mdir, "-l", lst, "-m", mlf] + flags, env)
                    if "error" in v:
                        print(tag, v["error"])
                        continue
                    mlfs[tag] = open(mlf, "rb").read()
                    if r > 0:
                        res[tag].append(v)
            for tag in "AB":
                med = lambda k: statistics.median(x[k] for x in res[tag])
                print("%s  process %.3f s (min %.3f)  main %.3f  first ctx %.3f  all ctx %.3f  list from first line %.3f  outside main %.3f" % (
                    tag, med("process_wall_s"), min(x["process_wall_s"] for x in res[tag]), med("main_s"), med("first_ctx_s"), med("create_s"),
                    med("list_from_first_line_s"), statistics.median(x["process_wall_s"] - x["main_s"] for x in res[tag])))
            print("same MLF:", mlfs.get("A") == mlfs.get("B"))
    finally:
        shutil.rmtree(libdir, ignore_errors=True)


if __name__ == "__main__":
    main()
