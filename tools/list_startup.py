#!/usr/bin/env python3
"""BASELINE configs[3] as a PROCESS: the HU system over the 10 000-file list, exec to exit.

What a caller of `phnrec -l list -m out.mlf` waits for, by mode and by -g (with -g 8 mapped onto this box's one GPU:
PHNREC_DEVICE_MAP=0 x 8 -- 24 contexts, one device), with the CLI's own break-down: setup_s (in front of the list),
first_ctx_s / create_s (until the first / the last context could take a launch: beside the list), list (from its first
line to its last, that start-up included), contexts that came up (of those planned).  Every MLF is compared with the first one byte for byte.

Every process starts bench.SETTLE_S (0.3 s) after the previous one has exited (pass 0 as the third argument for runs back to back:
each then waits in hipInit for the kernel to finish tearing down its predecessor's GPU state, 0.1-0.2 s).

usage: list_startup.py [n_files] [runs] [pause_s]
"""
import os
import statistics
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HU = "PHN_HU_SPDAT_LCRC_N1500"


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    if len(sys.argv) > 3:
        bench.SETTLE_S = float(sys.argv[3])
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", HU)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        lst, names, frames = bench.synthetic_list(td, n_files)
        print("%s, %d files, %d frames; median of %d runs after one discarded run, %.2f s between processes" % (HU, n_files, frames, runs, bench.SETTLE_S))
        print("%-26s %8s %8s %8s %8s %8s %8s %5s  %s" % ("run", "process", "main", "1st ctx", "all ctx", "list", "M fr/s", "ctxs", "mode"))
        ref = None
        for g, dmap in ((1, "0"), (2, "0,0"), (8, ",".join(["0"] * 8))):
            for name, extra, env_extra in (("default flags", [], {}), ("host front-end", [], {"PHNREC_NO_AUTO_E": "1", "PHNREC_NO_AUTO_D": "1"}),
                                           ("-F", ["-F"], {"PHNREC_NO_AUTO_D": "1"}), ("-F -D", ["-F", "-D"], {})):
                env = dict(os.environ, PHNREC_STATS="1", PHNREC_DEVICE_MAP=dmap, **env_extra)
                mlf = os.path.join(td, "out.mlf")
                rs = []
                for k in range(runs + 1):
                    r, pr = bench.run_cli(exe, ["-c", mdir, "-l", lst, "-m", mlf, "-g", str(g)] + extra, env)
                    if "error" in r:
                        print("-g %d %s: %s" % (g, name, r["error"]))
                        break
                    if k > 0:
                        rs.append(r)
                    data = open(mlf, "rb").read()
                    if ref is None:
                        ref = data
                    elif data != ref:
                        print("-g %d %s: MLF DIFFERS from the first run's" % (g, name))
                if not rs:
                    continue
                med = lambda key: statistics.median(x[key] for x in rs)
                print("-g %d %-21s %8.3f %8.3f %8.3f %8.3f %8.3f %8.2f %5d  %s" % (
                    g, name, med("process_wall_s"), med("main_s"), med("first_ctx_s"), med("create_s"),
                    med("list_from_first_line_s"), frames / med("process_wall_s") / 1e6, rs[-1]["contexts"], rs[-1]["mode"]))
        print("MLFs: every run's equals the first run's byte for byte" if ref is not None else "no run")


if __name__ == "__main__":
    main()
