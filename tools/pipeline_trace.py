import os, sys, tempfile, subprocess
sys.path.insert(0, os.getcwd())
import bench
exe = "phnrec_amd/bin/phnrec"
mdir = os.path.join("tests", "golden", "models", bench.HU)
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    lst, names, frames = bench.synthetic_list(td, 10000)
    for flags in (["-F"], ["-F", "-D"], []):
        for rep in range(2):
            env = dict(os.environ, PHNREC_STATS="1", PHNREC_TRACE_PIPELINE="1")
            p = subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf"), "-g", "1"] + flags, env=env, capture_output=True, text=True)
        print("=====", flags)
        print(p.stderr[:12000])
