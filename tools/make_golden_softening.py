#!/usr/bin/env python3
"""Fixtures of the posterior writer path ("next" row f2) from the REAL reference (build container only):
`phnrec_ref -t post` dumps with posteriors/softening_func = log / igor / gmm_bypass (srec.cpp:164-176,
1062-1070) on a 1.25 s cut of the bundled test.raw, CZ system.  The model directory is used in place through
symlinks; only its `config` is a modified copy.  Writes tests/golden/cli/soft_<name>.lop (68 KB each)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402

REF = "/root/reference"
CZ = "PHN_CZ_SPDAT_LCRC_N1500"
CASES = {"log": "log 0 0 0", "igor": "igor 0.5 10 10", "igor_asym": "igor 0.3 2.718282 10", "gmm_bypass": "gmm_bypass 0 0 0"}


def main():
    cli = ob.ref_cli_path()
    if cli is None or not os.path.isdir(REF):
        sys.exit("need /root/reference and oracle/_ref (cd oracle && make)")
    raw = open(os.path.join(REF, "test.raw"), "rb").read()[:20000]
    out_dir = os.path.join(ROOT, "tests", "golden", "cli")
    for name, value in CASES.items():
        with tempfile.TemporaryDirectory() as td:
            mdir = os.path.join(td, CZ)
            os.makedirs(mdir)
            for sub in os.listdir(os.path.join(REF, CZ)):
                if sub != "config":
                    os.symlink(os.path.join(REF, CZ, sub), os.path.join(mdir, sub))
            cfg = open(os.path.join(REF, CZ, "config")).read().splitlines(True)
            section, done = None, False
            for i, line in enumerate(cfg):
                if line.startswith("["):
                    section = line.strip()
                if section == "[posteriors]" and line.startswith("softening_func="):
                    cfg[i] = "softening_func=%s\n" % value
                    done = True
            assert done
            open(os.path.join(mdir, "config"), "w").writelines(cfg)
            open(os.path.join(td, "x.raw"), "wb").write(raw)
            subprocess.check_call([cli, "-c", mdir, "-i", os.path.join(td, "x.raw"), "-t", "post",
                                   "-o", os.path.join(out_dir, "soft_%s.lop" % name)], stdout=subprocess.DEVNULL)
        print("soft_%s.lop" % name, os.path.getsize(os.path.join(out_dir, "soft_%s.lop" % name)), "bytes")


if __name__ == "__main__":
    main()
