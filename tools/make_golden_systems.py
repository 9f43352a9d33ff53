#!/usr/bin/env python3
"""Goldens for the non-LCRC `posteriors/system` variants (1BT_DCT, 1BT, 3BT): outputs of the REAL reference
(oracle/_ref, built from /root/reference by oracle/Makefile) on seeded synthetic models -- the reference
ships no model and holds no test vector for them.  Run in the build container; writes
  tests/golden/systems.npz            <case>/{cfg, off, mel, post}   reference Traps, naive loop, bunch 5
  tests/golden/systems/1bt_dct.{lop,rec}   the reference CLI on the bundled test.raw with a synthetic
                                           1BT_DCT model directory (modelgen.write_traps_dir, seed 31)
  tests/golden/geometry.npz           the same for geometries no shipped model uses: posteriors/length other than 31
                                      (odd and even), LCRC without C0 / with another number of coefficients per band
  tests/golden/systems/lcrc_len21.lop the reference CLI on test.raw with an LCRC model of length 21, add_c0=false
The model directories are regenerated from the same seeds by the tests (phnrec_amd/modelgen.py).
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from phnrec_amd import modelgen  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# name, system, nbanks, hidden, n_out, seed, kwargs, utterance lengths
CASES = [
    ("dct_c0", "1BT_DCT", 15, 120, 45, 31, dict(coefs=6), [40, 1, 17, 70]),
    ("dct_noc0_hamm", "1BT_DCT", 23, 64, 30, 32, dict(coefs=5, add_c0=False, hamming=True), [33, 64]),
    ("dct_wide", "1BT_DCT", 23, 200, 120, 33, dict(coefs=16), [50, 20]),
    ("1bt", "1BT", 15, 90, 42, 34, dict(band_out=12, band_hidden=40), [48, 3, 30]),
    ("1bt_hamm", "1BT", 11, 50, 27, 35, dict(band_out=20, band_hidden=17, hamming=True), [31, 32]),
    ("3bt", "3BT", 15, 70, 33, 36, dict(band_out=9, band_hidden=33), [45, 16]),
]
CLI_CASE = dict(system="1BT_DCT", nbanks=15, hidden=100, n_out=138, seed=31, coefs=6)
# ... and posteriors/length (last field); system LCRC: modelgen.write_model_dir with `coefs` inputs per band
GEOM_CASES = [
    ("lcrc_21_noc0", "LCRC", 15, 96, 45, 41, dict(coefs=5, add_c0=False), [40, 1, 17, 70], 21),
    ("lcrc_31_c0_8", "LCRC", 23, 64, 30, 42, dict(coefs=8), [33, 64], 31),
    ("lcrc_31_noc0_11", "LCRC", 15, 80, 36, 43, dict(coefs=11, add_c0=False), [20, 47], 31),
    ("lcrc_30_even", "LCRC", 11, 50, 27, 44, dict(coefs=7), [31, 32, 2], 30),
    ("lcrc_9", "LCRC", 7, 40, 21, 45, dict(coefs=3, add_c0=False), [12, 5], 9),
    ("lcrc_51", "LCRC", 15, 72, 33, 46, dict(coefs=14), [60, 26], 51),
    ("1bt_21_hamm", "1BT", 15, 90, 42, 47, dict(band_out=12, band_hidden=40, hamming=True), [48, 3, 30], 21),
    ("3bt_25", "3BT", 11, 70, 33, 48, dict(band_out=9, band_hidden=33), [45, 16], 25),
    ("dct_41_noc0", "1BT_DCT", 15, 120, 45, 49, dict(coefs=6, add_c0=False, hamming=True), [40, 1, 44], 41),
    ("dct_12_even", "1BT_DCT", 23, 64, 30, 50, dict(coefs=5), [33, 20], 12),
]
GEOM_CLI_CASE = dict(nbanks=15, hidden=100, n_out=45, seed=51, coefs=5, add_c0=False, trap_len=21)
# offlinenorm/sent_max_norm and sent_chmax_norm (srec.cpp:1547-1587; no shipped config sets them): the reference CLI's
# posterior dumps on test.raw for a synthetic LCRC model of the usual geometry -> tests/golden/systems/lcrc_<name>.lop
NORM_CLI_MODEL = dict(nbanks=15, hidden=100, n_out=45, seed=52)
NORM_CLI_CASES = [("maxnorm", dict(sent_max_norm=True)), ("chmaxnorm", dict(sent_chmax_norm=True, sent_mean_norm=False)),
                  ("bothmax", dict(sent_max_norm=True, sent_chmax_norm=True))]


def write_geometry_model(path, system, nb, hid, nout, seed, kw, trap_len):
    if system == "LCRC":
        return modelgen.write_model_dir(path, nb, hid, nout, seed=seed, trap_len=trap_len, **kw)
    return modelgen.write_traps_dir(path, system, nb, hid, nout, seed=seed, trap_len=trap_len, **kw)


def main():
    out = {}
    for name, system, nb, hid, nout, seed, kw, lens in CASES:
        with tempfile.TemporaryDirectory() as td:
            modelgen.write_traps_dir(td, system, nb, hid, nout, seed=seed, **kw)
            t = ob.RefTraps(td, nb, bunch=5, system=system, add_c0=kw.get("add_c0", True),
                            hamming=kw.get("hamming", False))
            mels = [modelgen.synth_mel(n, nb, seed=1000 * seed + i) for i, n in enumerate(lens)]
            posts = [t.process_offline(m) for m in mels]
        out[name + "/off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        out[name + "/mel"] = np.concatenate(mels)
        out[name + "/post"] = np.concatenate(posts)
        print(name, system, lens, "ok")
    np.savez_compressed(os.path.join(GOLD, "systems.npz"), **out)

    out = {}
    for name, system, nb, hid, nout, seed, kw, lens, trap_len in GEOM_CASES:
        with tempfile.TemporaryDirectory() as td:
            write_geometry_model(td, system, nb, hid, nout, seed, kw, trap_len)
            t = ob.RefTraps(td, nb, bunch=5, system=system, add_c0=kw.get("add_c0", True),
                            hamming=kw.get("hamming", False), trap_len=trap_len)
            mels = [modelgen.synth_mel(n, nb, seed=1000 * seed + i) for i, n in enumerate(lens)]
            posts = [t.process_offline(m) for m in mels]
        out[name + "/off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        out[name + "/mel"] = np.concatenate(mels)
        out[name + "/post"] = np.concatenate(posts)
        print(name, system, trap_len, lens, "ok")
    np.savez_compressed(os.path.join(GOLD, "geometry.npz"), **out)

    cli = ob.ref_cli_path(False)
    sub = os.path.join(GOLD, "systems")
    os.makedirs(sub, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        c = dict(GEOM_CLI_CASE)
        modelgen.write_model_dir(td, c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"), seed=c.pop("seed"), **c)
        dst = os.path.join(td, "out.lop")
        subprocess.run([cli, "-c", td, "-i", os.path.join(GOLD, "test.raw"), "-t", "post", "-o", dst], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.copyfile(dst, os.path.join(sub, "lcrc_len21.lop"))
    for name, cfg in NORM_CLI_CASES:
        with tempfile.TemporaryDirectory() as td:
            c = dict(NORM_CLI_MODEL)
            modelgen.write_model_dir(td, c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"), seed=c.pop("seed"), **cfg)
            dst = os.path.join(td, "out.lop")
            subprocess.run([cli, "-c", td, "-i", os.path.join(GOLD, "test.raw"), "-t", "post", "-o", dst], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            shutil.copyfile(dst, os.path.join(sub, "lcrc_%s.lop" % name))
    with tempfile.TemporaryDirectory() as td:
        c = dict(CLI_CASE)
        modelgen.write_traps_dir(td, c.pop("system"), c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"),
                                 seed=c.pop("seed"), **c)
        raw = os.path.join(GOLD, "test.raw")
        for kind, suffix in (("post", "lop"), ("str", "rec")):
            dst = os.path.join(td, "out." + suffix)
            subprocess.run([cli, "-c", td, "-i", raw, "-t", kind, "-o", dst], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            shutil.copyfile(dst, os.path.join(sub, "1bt_dct." + suffix))
    print("cli goldens ok")


if __name__ == "__main__":
    main()
