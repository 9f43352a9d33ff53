#!/usr/bin/env python3
"""BASELINE configs[4] on ONE device (four `phnrec -g 2` processes share it): which flag set should the arrangement take
there?  The launch-order gate of -D is per process; four processes that each run "one posterior kernel at a time" still
run against each other.  Variants: the default flags, -F, -F -D (ordered), -F -D with PHNREC_LAUNCH_ORDER=0 (shared).
usage: four_systems_ab.py [files_per_system = 2500] [rounds = 3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = (("default flags", [], {}), ("-F", ["-F"], {"PHNREC_NO_AUTO_D": "1"}), ("-F -D ordered", ["-F", "-D"], {}),
            ("-F -D shared", ["-F", "-D"], {"PHNREC_LAUNCH_ORDER": "0"}))
for r in range(rounds):
    out = bench.four_systems_leg(1, [0], n, variants=variants)
    for key, _f, _e in variants:
        v = out.get(key, {})
        if "error" in v:
            print(r, key, v["error"])
            continue
        print("%d %-16s whole script %.2f M frames/s (%.3f s)  list loops %.2f M  per system: %s" % (
            r, key, v["value"] / 1e6, v["process_wall_s"], v["list_loops_frames_per_s"] / 1e6,
            "  ".join("%s %.1f M %s ctx %d" % (k[4:6], p["frames_per_s"] / 1e6, p["mode"], p["contexts"]) for k, p in sorted(v["per_system"].items()))), flush=True)
