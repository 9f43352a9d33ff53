#!/usr/bin/env python3
"""Where a CLI list run's wall clock goes on the GPU: runs `phnrec` on the configs[3] list under rocprofv3 --kernel-trace
and reports, per kernel, calls / total / average, and the UNION of the kernels' busy intervals against the span from the
first launch to the last (idle share = the host could not keep the device fed).  usage: cli_timeline.py [files] [flags...]
env TIMELINE_REPS=n: every file listed n times (a list loop of seconds); TIMELINE_ROWS=k: the first k launches one by one;
TIMELINE_PAR=1: the list's files are first turned into parameter files (`-t par`, host front-end) and the traced run reads
those (`-s par`): no front-end kernels on the device, only the posterior kernels (and the decoder's with -D)."""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    flags = sys.argv[2:] or ["-F"]
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", bench.HU)
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        lst, names, frames = bench.synthetic_list(td, n_files)
        reps = int(os.environ.get("TIMELINE_REPS", "1"))
        if os.environ.get("TIMELINE_PAR"):
            par_lst = os.path.join(td, "par.scp")
            with open(par_lst, "w") as f:
                f.write("".join("%s %s.mel\n" % (n, n) for n in names))
            subprocess.run([exe, "-c", mdir, "-l", par_lst, "-t", "par"], check=True, capture_output=True)
            names = [n + ".mel" for n in names]
            lst = os.path.join(td, "mel.scp")
            with open(lst, "w") as f:
                f.write("".join(n + "\n" for n in names))
            flags = ["-s", "par"] + flags
        if reps > 1:
            lst = os.path.join(td, "rep.scp")
            with open(lst, "w") as f:
                for _ in range(reps):
                    f.write("".join(n + "\n" for n in names))
            frames *= reps
        out = os.path.join(td, "prof")
        env = dict(os.environ, PHNREC_STATS="1", TMPDIR="/tmp")
        subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "w.mlf")] + flags, env=env, capture_output=True)   # page cache
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                            exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "o.mlf")] + flags, env=env, capture_output=True, text=True)
        print([l for l in p.stderr.splitlines() if l.startswith("phnrec:")][-1:])
        rows = []
        for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows)
        if os.environ.get("TIMELINE_ROWS"):          # the first launches one by one, and every idle gap above 0.3 ms
            t0 = iv[0][0]
            q = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): r.get("Queue_Id", "?") for r in rows}
            for s_, e_, n_ in iv[:int(os.environ["TIMELINE_ROWS"])]:
                print("    +%9.3f ms  %8.1f us  queue %-3s %s" % ((s_ - t0) / 1e6, (e_ - s_) / 1e3, q[(s_, e_)], n_[-40:]))
            end = iv[0][1]
            for s_, e_, n_ in iv[1:]:
                if s_ - end > 300000:
                    print("    idle %7.3f ms before +%9.3f ms (%s)" % ((s_ - end) / 1e6, (s_ - t0) / 1e6, n_[-30:]))
                end = max(end, e_)
        per = {}
        for s, e, n in iv:
            c = per.setdefault(n, [0, 0])
            c[0] += 1
            c[1] += e - s
        span = iv[-1][1] - iv[0][0]
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for s, e, _ in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print("files %d frames %d flags %s" % (n_files, frames, " ".join(flags)))
        print("span first launch -> last end: %.1f ms; device busy (union of kernels): %.1f ms = %.0f %%; sum of kernel times %.1f ms"
              % (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(c[1] for c in per.values()) / 1e6))
        for n, (calls, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print("  %-62s calls %5d  total %8.2f ms  avg %8.1f us" % (n, calls, tot / 1e6, tot / calls / 1e3))
        # the posterior kernel alone: how much of the span at least one / exactly one / two or more of its launches run, and how
        # much of its running time a decoder kernel runs beside it
        ev = []
        for s, e, n in iv:
            kind = 0 if "lcrc_fused_kernel" in n else 1 if "phndec" in n else -1
            if kind >= 0:
                ev += [(s, 0, kind), (e, 1, kind)]
        ev.sort()
        cnt, last, acc = [0, 0], ev[0][0] if ev else 0, {}
        for t, typ, kind in ev:
            key = (min(cnt[0], 2), min(cnt[1], 1))
            acc[key] = acc.get(key, 0) + t - last
            last = t
            cnt[kind] += 1 if typ == 0 else -1
        tot = float(sum(acc.values())) or 1.0
        print("  posterior kernels in flight x decoder kernel in flight, share of the span between the first and last of them:")
        for key in sorted(acc):
            print("    posterior %s, decoder %s: %7.1f ms = %4.1f %%" % (("0", "1", ">=2")[key[0]], ("no", "yes")[key[1]], acc[key] / 1e6, 100.0 * acc[key] / tot))


if __name__ == "__main__":
    main()
