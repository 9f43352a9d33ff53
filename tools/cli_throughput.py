#!/usr/bin/env python3
"""End-to-end throughput of the drop-in CLI on a synthetic file list (BASELINE configs[3]-like):
raw 8 kHz lin16 files of 3-15 s -> MLF, everything included (file reads, host front-end, GPU posteriors,
host Viterbi, output).  Prints the CLI's own PHNREC_STATS line per run."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    system = sys.argv[2] if len(sys.argv) > 2 else "PHN_CZ_SPDAT_LCRC_N1500"
    mdir = os.path.join(ROOT, "tests", "golden", "models", system)
    rng = np.random.default_rng(1236)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        names = []
        total = 0
        for i in range(n_files):
            secs = rng.uniform(3.0, 15.0)
            n = int(secs * 8000)
            t = np.arange(n) / 8000.0
            sig = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) for f in (200, 700, 1300, 2100, 3400))
            sig = sig + rng.normal(0, 1000, n)
            p = os.path.join(td, "f%05d.raw" % i)
            np.clip(sig, -32768, 32767).astype("<i2").tofile(p)
            names.append(p)
            total += (n - 200) // 80 + 1
        lst = os.path.join(td, "list.scp")
        open(lst, "w").write("".join(n + "\n" for n in names))
        print("files %d, frames %d (%.1f h of audio)" % (n_files, total, total / 360000.0), flush=True)
        # how fast can this box read the list at all?  (16 reader threads, page cache)
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.time()
        with ThreadPoolExecutor(16) as ex:
            nbytes = sum(ex.map(lambda p: len(open(p, "rb").read()), names))
        dt = time.time() - t0
        print("plain read of the list: %.1f MB in %.3f s (%.2f GB/s); usable cores: %d" %
              (nbytes / 1e6, dt, nbytes / dt / 1e9, len(os.sched_getaffinity(0))), flush=True)
        try:
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            print("cgroup cpu.max:", q, flush=True)
        except OSError:
            pass
        env = dict(os.environ, PHNREC_STATS="1")
        if len(sys.argv) > 3:
            env["PHNREC_CTX_PER_GPU"] = sys.argv[3]
        if os.environ.get("CLI_ROCPROF"):      # per-kernel times of one configuration (dev aid)
            out = os.path.join(ROOT, "gpurun_out", "cli_prof")
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--",
                   BIN, "-c", mdir, "-l", lst] + os.environ["CLI_ROCPROF"].split() + ["-m", os.path.join(td, "out.mlf")]
            subprocess.run(cmd, env=dict(env, TMPDIR="/tmp"), capture_output=True, text=True)
            import glob
            for f in glob.glob(os.path.join(out, "*", "*_kernel_stats.csv")):
                print(open(f).read())
            return
        for extra, label in ((["-t", "post"], "wf->post (HTK dumps)"), (["-m", os.path.join(td, "out.mlf")], "wf->str (MLF)"), (["-F", "-m", os.path.join(td, "out.mlf")], "wf->str, GPU front-end (-F)"),
                             (["-F", "-D", "-m", os.path.join(td, "out.mlf")], "wf->str, GPU front-end + decoder (-F -D)"),
                             (["-F", "-H", "-m", os.path.join(td, "out.mlf")], "wf->str, -F, split-f16 arithmetic (-H)"),
                             (["-F", "-D", "-H", "-m", os.path.join(td, "out.mlf")], "wf->str, -F -D -H"),
                             (["-F", "-D", "-b", "131072", "-m", os.path.join(td, "out.mlf")], "wf->str, -F -D, 131072 frames per launch"),
                             (["-F", "-b", "131072", "-m", os.path.join(td, "out.mlf")], "wf->str, -F, 131072 frames per launch"),
                             (["-m", os.path.join(td, "out.mlf"), "-j", "8"], "wf->str, 8 host threads")):
            t0 = time.time()
            p = subprocess.run([BIN, "-c", mdir, "-l", lst] + extra, env=env, capture_output=True, text=True)
            dt = time.time() - t0
            print("%-40s rc=%d wall %.2fs  %s" % (label, p.returncode, dt, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else ""), flush=True)


if __name__ == "__main__":
    main()
