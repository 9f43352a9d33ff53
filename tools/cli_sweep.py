#!/usr/bin/env python3
"""Sweep of the CLI's list modes over pipeline settings on BASELINE configs[3]'s list (HU, 10 000 files):
contexts per GPU (PHNREC_CTX_PER_GPU), frames per launch (-b).
usage: cli_sweep.py [n_files] ["ctx,batch;ctx,batch;..."]   (batch 0 = the CLI's default)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import tempfile

import bench

exe = "phnrec_amd/bin/phnrec"
mdir = os.path.join("tests", "golden", "models", bench.HU)
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
grid = sys.argv[2] if len(sys.argv) > 2 else "2,0;3,0;4,0;3,16384;3,65536"
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, n_files)
    print("files %d frames %d cores %d" % (n_files, frames, bench.usable_cpus()), flush=True)
    for flags in ([], ["-E"], ["-F"], ["-F", "-D"]):
        for cfg in grid.split(";"):
            ctx, batch = (int(x) for x in cfg.split(","))
            env = dict(os.environ, PHNREC_STATS="1", PHNREC_CTX_PER_GPU=str(ctx))
            if not flags:
                env["PHNREC_NO_AUTO_E"] = "1"          # (host fe): a list of this length would take -F by itself
            extra = ["-b", str(batch)] if batch else []
            best = None
            for _ in range(3):
                r, _pr = bench.run_cli(exe, ["-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-g", "1"] + flags + extra, env)
                if "error" in r:
                    best = r
                    break
                if best is None or r["value"] > best["value"]:
                    best = r
            if "error" in best:
                print(flags, cfg, best, flush=True)
                continue
            print("%-12s ctx %d batch %6d: %6.2f M frames/s  wall %.3f  kernel_ms %6.1f  host_cpu_s %.3f %s"
                  % (" ".join(flags) or "(host fe)", ctx, batch, best["value"] / 1e6, best["list_wall_s"], best["gpu_kernel_ms"],
                     best["host_cpu_s"], best.get("cpu_s_by_stage")), flush=True)
