#!/usr/bin/env python3
"""configs[3] as a process with two against three contexts per GPU (PHNREC_CTX_PER_GPU), alternating, by mode; and the
8 x list (steady state).  usage: ctx_count_ab.py [rounds = 6]"""
import os, sys, statistics, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
m = os.path.join(ROOT, "tests", "golden", "models", bench.HU)
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, 10000)
    rep = os.path.join(td, "x8.scp")
    open(rep, "w").write("".join(n + "\n" for n in names) * 8)
    for label, l, fr, rr in (("1 x list", lst, frames, rounds), ("8 x list", rep, 8 * frames, 2)):
        for flags, extra in ((["-F"], {}), (["-F", "-D"], {}), ([], {"PHNREC_NO_AUTO_E": "1"})):
            res = {}
            for r in range(rr + 1):
                for n in ("2", "3"):
                    v, _ = bench.run_cli(exe, ["-c", m, "-l", l, "-m", td + "/o.mlf"] + flags, dict(os.environ, PHNREC_STATS="1", PHNREC_CTX_PER_GPU=n, **extra), timeout=900)
                    if r > 0 and "error" not in v:
                        res.setdefault(n, []).append(v)
            print("%s %-8s  " % (label, " ".join(flags) or "host") + "   ".join(
                "%s contexts: process %.3f s, list %.2f M fr/s" % (n, statistics.median(x["process_wall_s"] for x in v), statistics.median(x["value"] for x in v) / 1e6)
                for n, v in sorted(res.items())), flush=True)
