#!/usr/bin/env python3
"""Where does the host Viterbi's CPU time go in a list run?  (dev aid)  Posterior dumps of the first files of
configs[3]'s list, then `phnrec -s post -l ... -m` with 1 / 4 / 16 pool threads: viterbi CPU seconds per frame by thread
count; and the bare decoder micro-benchmark run as 1 and as 16 simultaneous processes."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import bench

exe = "phnrec_amd/bin/phnrec"
mdir = os.path.join("tests", "golden", "models", bench.HU)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, n)
    sub = os.path.join(td, "sub.scp")
    open(sub, "w").write("".join("%s %s\n" % (names[i], os.path.join(td, "p%05d.lop" % i)) for i in range(n)))
    env = dict(os.environ, PHNREC_STATS="1")
    r, pr = bench.run_cli(exe, ["-c", mdir, "-l", sub, "-t", "post", "-F"], env)
    print("dumps:", r.get("value"), flush=True)
    lop = os.path.join(td, "lop.scp")
    open(lop, "w").write("".join("%s\n" % os.path.join(td, "p%05d.lop" % i) for i in range(n)))
    for j in (1, 4, 16):
        for extra_env in ({}, {"PHNREC_NO_AVX512": "1"}):
            r, pr = bench.run_cli(exe, ["-c", mdir, "-s", "post", "-l", lop, "-m", os.path.join(td, "v.mlf"), "-j", str(j)], dict(env, **extra_env))
            st = r.get("cpu_s_by_stage", {})
            print("-s post -j %2d %s: %.2f M frames/s, viterbi %.3f CPU-s = %.1f ns/frame, stage1 %.3f decode_write %.3f"
                  % (j, extra_env, r["value"] / 1e6, st.get("viterbi", 0), st.get("viterbi", 0) / frames * 1e9, st.get("stage1", 0),
                     st.get("decode_write", 0)), flush=True)
ub = os.path.join("tools", "ubench", "phndec_host_bench")
for procs in (1, 16):
    ps = [subprocess.Popen([ub, "61", "900", "400"], stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    outs = [p.communicate()[0].strip() for p in ps]
    print("%d simultaneous micro-benchmarks:" % procs, outs[0], "|", outs[-1], flush=True)
