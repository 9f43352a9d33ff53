#!/usr/bin/env python3
"""A/B of the CLI's list modes over two builds of the library on the same box, alternating, on the configs[3] list with every
file listed REPS times (a list loop of seconds, not of 0.3 s).  Each build is a copy named libphnrec_lcrc.so in its own
temporary directory, selected through LD_LIBRARY_PATH (the CLI finds its library by RUNPATH, which LD_LIBRARY_PATH
precedes): the installed library is never touched, whatever kills this tool.
    ab_cli_list.py LIB_A LIB_B [reps = 4] [rounds = 3] [mode ...]        ("-" = the library as it stands; a mode is the flag
                                                                        string, e.g. "-F" "-F -D" "-E -D" "" (the host
                                                                        front-end: run with PHNREC_NO_AUTO_E=1); default -F, -F -D)
An environment variant instead of a second library: LIB_B = "env:NAME=VALUE[,NAME=VALUE]" runs the same library with it set."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench  # noqa: E402


def main():
    a, b = sys.argv[1], sys.argv[2]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    modes = sys.argv[5:] or ["-F", "-F -D"]
    lib = os.path.join(ROOT, "phnrec_amd", "lib", "libphnrec_lcrc.so")
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", bench.HU)
    libdir = tempfile.mkdtemp(prefix="ab_cli_list_")
    builds = []
    for tag, x in zip("AB", (a, b)):
        extra, src = {}, lib
        if x.startswith("env:"):
            extra = dict(kv.split("=", 1) for kv in x[4:].split(","))
        elif x != "-":
            src = os.path.join(ROOT, x)
        d = os.path.join(libdir, tag)
        os.mkdir(d)
        shutil.copyfile(src, os.path.join(d, "libphnrec_lcrc.so"))
        builds.append((d, extra))
    res = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            lst, names, frames = bench.synthetic_list(td, 10000)
            rep_lst = os.path.join(td, "rep.scp")
            with open(rep_lst, "w") as f:
                for _ in range(reps):
                    f.write("".join(n + "\n" for n in names))
            for r in range(rounds):
                for mode in modes:
                    row, mlfs = [], []
                    for tag, (d, extra) in zip("AB", builds):
                        mlf = os.path.join(td, "o%s.mlf" % tag)
                        env = dict(os.environ, PHNREC_STATS="1", **extra)
                        env["LD_LIBRARY_PATH"] = d + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
                        if not mode.split():
                            env.setdefault("PHNREC_NO_AUTO_E", "1")
                        v, _pr = bench.run_cli(exe, ["-c", mdir, "-l", rep_lst, "-m", mlf] + mode.split(), env)
                        if "error" in v:
                            print(tag, mode, v["error"], flush=True)
                            continue
                        res.setdefault((mode, tag), []).append(v["value"])
                        row.append("%s %.2f M (kernels %.0f ms, host %.2f CPU-s, process %.2f s)"
                                   % (tag, v["value"] / 1e6, v["gpu_kernel_ms"], v["host_cpu_s"], v["process_wall_s"]))
                        mlfs.append(open(mlf, "rb").read())
                    print(r, repr(mode), "  ".join(row), " same MLF:", len(mlfs) == 2 and mlfs[0] == mlfs[1], flush=True)
    finally:
        shutil.rmtree(libdir, ignore_errors=True)
    for (mode, tag), v in sorted(res.items()):
        v = sorted(v)
        print("median %-8s %s %.2f M  (min %.2f, max %.2f)" % (repr(mode), tag, v[len(v) // 2] / 1e6, v[0] / 1e6, v[-1] / 1e6))


if __name__ == "__main__":
    main()
