#!/usr/bin/env python3
"""Randomised run of the CLI's list pipeline (GPU): random lists -- files of 0 bytes, less than a frame, exactly a frame, up to
20 s, optionally one unreadable name --, random batch sizes (-b), logical GPU counts (-g N on the one device), host threads
(-j), modes (host front-end, -E, -E -D, -F, -F -D), contexts per GPU (created beside the list: those the list lives to see, or
all of them), launch order and decoder overlap on / off, over the four
shipped systems (8 kHz lin16 and A-law, 16 kHz lin16): every
configuration must write the MLF the plain host-front-end run writes, byte for byte (every mode's features and labels are
the host's), or fail the same way, and none may hang (each run has a time limit).     usage: fuzz_cli.py [seed [lists]]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
MODELS = [os.path.join(ROOT, "tests", "golden", "models", m) for m in
          ("PHN_CZ_SPDAT_LCRC_N1500", "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500")]


def run(args, env=None, limit=120):
    p = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=limit,
                       env=dict(os.environ, **(env or {})))
    return p.returncode, p.stderr


def fuzz(seed=0, n_lists=12, log=print):
    rng = np.random.default_rng(seed)
    runs = 0
    for it in range(n_lists):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            # the system of this list (the first lists: CZ as ever, then all four in turn), its frame size, and the waveform
            # format: lin16, or A-law bytes for the 8 kHz systems
            MODEL = MODELS[0] if it < 2 else MODELS[int(rng.integers(0, 4))]
            vs = 400 if "PHN_EN" in MODEL else 200
            alaw = it >= 2 and vs == 200 and int(rng.integers(0, 3)) == 0
            wfmt = ["-w", "alaw"] if alaw else []
            n_files = int(rng.integers(1, 70))
            names = []
            for i in range(n_files):
                kind = int(rng.integers(0, 10))
                n = 0 if kind == 0 else int(rng.integers(1, vs)) if kind == 1 else vs if kind == 2 else int(rng.integers(vs + 1, 800 * vs))
                p = os.path.join(td, "f%03d.raw" % i)
                if alaw:
                    rng.integers(0, 256, n, dtype=np.uint8).tofile(p)
                else:
                    (rng.normal(0, 2500, n).clip(-32768, 32767).astype("<i2")).tofile(p)
                names.append(p)
            bad = int(rng.integers(0, 4)) == 0
            if bad:
                names.insert(int(rng.integers(0, len(names) + 1)), os.path.join(td, "missing.raw"))
            lst = os.path.join(td, "list.scp")
            open(lst, "w").write("".join(n + "\n" for n in names))
            ref_mlf = os.path.join(td, "ref.mlf")
            rc0, err0 = run(["-c", MODEL, "-l", lst, "-m", ref_mlf] + wfmt, env={"PHNREC_NO_AUTO_E": "1", "PHNREC_NO_AUTO_D": "1"})
            assert (rc0 != 0) == bad, (seed, it, rc0, err0[-300:])
            want = open(ref_mlf).read() if os.path.exists(ref_mlf) else None
            for mode in ([], ["-E"], ["-E", "-D"], ["-F"], ["-F", "-D"]):
                for _ in range(3):
                    g = int(rng.choice([1, 1, 2, 3, 4]))
                    b = int(rng.choice([64, 700, 5000, 32768, 200000]))
                    j = int(rng.choice([1, 2, 5, 16]))
                    out = os.path.join(td, "o.mlf")
                    if os.path.exists(out):
                        os.remove(out)
                    env = {"PHNREC_DEVICE_MAP": ",".join(["0"] * g)}
                    for name, choices in (("PHNREC_CTX_PER_GPU", ["", "1", "2", "3", "4"]), ("PHNREC_LAUNCH_ORDER", ["", "0", "1"]),
                                          ("PHNREC_DECODER_OVERLAP", ["", "0", "1"]), ("PHNREC_ALL_CONTEXTS", ["", "", "1"])):
                        v = str(rng.choice(choices))
                        if v:
                            env[name] = v
                    rc, err = run(["-c", MODEL, "-l", lst, "-m", out, "-g", g, "-b", b, "-j", j] + wfmt + mode, env=env)
                    got = open(out).read() if os.path.exists(out) else None
                    assert rc == rc0 and got == want, (seed, it, os.path.basename(MODEL), wfmt, mode, g, b, j, env, rc, err[-300:])
                    runs += 1
    log("cli fuzz ok: %d lists, %d configurations" % (n_lists, runs))
    return runs


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    fuzz(*a)
