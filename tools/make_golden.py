#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (run in the build container only).

Needs /root/reference and oracle/_ref (cd oracle && make).  Nothing here runs on
the GPU box; the fixtures it writes are data (inputs + expected outputs):

  test.raw                          the reference's bundled test input (test.sh:1)
  rec/*.rec                         the reference's golden label files for it
                                    (test.rec.org, test_en.rec, test_hu.rec, test_ru.rec)
  models/PHN_{CZ,EN,HU,RU}_*        the four model directories the tests need on the
                                    GPU box (weights .nbin, norms, windows, config,
                                    phoneme list, licence) -- research-licensed DATA
  <SYS>/test.mel  <SYS>/test.lop    `phnrec_ref -t par` / `-t post` HTK dumps for all
                                    four systems (mel is BEFORE sentence mean-norm)
  <SYS>/probe.npz                   intermediates (band inputs, band posteriors,
                                    merger input) for the first 64 + last 32 frames,
                                    read out of the reference's Traps object
  synth.npz                         posteriors of the reference on seeded synthetic
                                    model directories (phnrec_amd/modelgen.py) incl.
                                    short/ragged utterances
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob          # noqa: E402
from phnrec_amd import modelgen           # noqa: E402
from tests.util import read_htk           # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

REC = {"PHN_CZ_SPDAT_LCRC_N1500": "test.rec.org", "PHN_EN_TIMIT_LCRC_N500": "test_en.rec",
       "PHN_HU_SPDAT_LCRC_N1500": "test_hu.rec", "PHN_RU_SPDAT_LCRC_N1500": "test_ru.rec"}
SHIP_MODELS = tuple(REC)      # all four shipped systems (BASELINE configs[3] names HU, configs[4] all of them)

# (name, nbanks, hidden, n_out, seed, utterance lengths)
SYNTH_CASES = [
    ("tiny", 15, 64, 24, 11, [1, 2, 5, 14, 15, 16, 17, 31, 40]),
    ("odd", 23, 100, 30, 12, [33, 7]),
    ("hu_shape", 15, 1500, 186, 13, [48]),
    ("ru_shape", 15, 1400, 159, 14, [48]),
    ("cz_shape", 15, 1500, 138, 15, [70, 3, 26]),
    ("en_shape", 23, 500, 120, 16, [64]),
]


def run(cmd, cwd=None):
    subprocess.check_call(cmd, cwd=cwd, stdout=subprocess.DEVNULL)


def main():
    cli = ob.ref_cli_path()
    if cli is None or not os.path.isdir(REF):
        sys.exit("need /root/reference and oracle/_ref (cd oracle && make)")
    os.makedirs(os.path.join(GOLD, "rec"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "test.raw"), os.path.join(GOLD, "test.raw"))
    for sysname, rec in REC.items():
        with open(os.path.join(REF, rec), "rb") as f:
            txt = f.read().replace(b"\r\n", b"\n")
        with open(os.path.join(GOLD, "rec", sysname + ".rec"), "wb") as f:
            f.write(txt)

    for sysname in SHIP_MODELS:
        dst = os.path.join(GOLD, "models", sysname)
        for sub, names in (("weights", ["band0.nbin", "band1.nbin", "merger.nbin"]),
                           ("norms", ["band0.norms", "band1.norms", "merger.norms"]),
                           ("windows", ["band0.window", "band1.window"]),
                           ("dicts", ["phonemes"]), ("", ["config", "licence.txt"])):
            os.makedirs(os.path.join(dst, sub), exist_ok=True)
            for n in names:
                src = os.path.join(REF, sysname, sub, n)
                if os.path.exists(src):
                    shutil.copyfile(src, os.path.join(dst, sub, n))
                    os.chmod(os.path.join(dst, sub, n), 0o644)

    for sysname, spec in modelgen.SYSTEMS.items():
        out = os.path.join(GOLD, sysname)
        os.makedirs(out, exist_ok=True)
        mdir = os.path.join(REF, sysname)
        run([cli, "-c", mdir, "-i", os.path.join(REF, "test.raw"), "-t", "par",
             "-o", os.path.join(out, "test.mel")])
        run([cli, "-c", mdir, "-i", os.path.join(REF, "test.raw"), "-t", "post",
             "-o", os.path.join(out, "test.lop")])
        run([cli, "-c", mdir, "-i", os.path.join(REF, "test.raw"),
             "-o", os.path.join(out, "test.rec")])
        mel = read_htk(os.path.join(out, "test.mel"))
        if spec["sent_mean_norm"]:
            mel = ob.sentence_mean_norm(mel)
        # intermediates straight out of the reference's Traps object, bunch by bunch
        nb, n = spec["nbanks"], mel.shape[0]
        t = ob.RefTraps(mdir, nb, bunch=5)
        k, o = nb * 11, spec["n_out"]
        keep = list(range(0, min(64, n))) + list(range(max(64, n - 32), n))
        padded = np.concatenate([mel, np.repeat(mel[-1:], 15, axis=0)])  # flush frames
        t.reset()
        t.calc_bunched(padded[:15], needed=False)
        probes = {key: np.zeros((n, w), np.float32)
                  for key, w in (("in0", k), ("in1", k), ("p0", o), ("p1", o), ("g", 2 * o))}
        post = np.zeros((n, o), np.float32)
        for r0 in range(0, n, 5):
            m = min(5, n - r0)
            post[r0:r0 + m] = t.calc_bunched(padded[15 + r0:15 + r0 + m])
            for which, key in enumerate(("in0", "in1", "p0", "p1", "g")):
                probes[key][r0:r0 + m] = t.probe(which, m, probes[key].shape[1])
        lop = read_htk(os.path.join(out, "test.lop"))
        assert np.array_equal(post, lop), "shim-driven posteriors differ from the CLI dump"
        np.savez_compressed(os.path.join(out, "probe.npz"), rows=np.array(keep, np.int32),
                            **{key: v[keep] for key, v in probes.items()})
        # config + phoneme list (tiny text) so the CLI's decoder can be tested on all four systems
        shutil.copyfile(os.path.join(mdir, "config"), os.path.join(out, "config"))
        shutil.copyfile(os.path.join(mdir, "dicts", "phonemes"), os.path.join(out, "phonemes"))
        print(sysname, "frames", n, "ok")

    # -- CLI goldens (CZ): A-law front-end, list mode with an MLF, short files --
    cz = "PHN_CZ_SPDAT_LCRC_N1500"
    cli_dir = os.path.join(GOLD, "cli")
    os.makedirs(cli_dir, exist_ok=True)
    raw = open(os.path.join(REF, "test.raw"), "rb").read()
    with tempfile.TemporaryDirectory() as td:
        run([cli, "-c", os.path.join(REF, cz), "-w", "alaw", "-i", os.path.join(REF, "test.raw"), "-t", "par",
             "-o", os.path.join(cli_dir, "test_alaw.mel")])
        # three utterances cut from test.raw: full, 1.25 s, 0.19 s (17 frames: shorter than the context)
        pieces = {"utt_a.raw": raw, "utt_b.raw": raw[:20000], "utt_c.raw": raw[:3000]}
        sub = os.path.join(td, "data")
        os.makedirs(sub)
        for name, blob in pieces.items():
            open(os.path.join(sub, name), "wb").write(blob)
        lst = os.path.join(td, "list.txt")
        open(lst, "w").write("".join(os.path.join(sub, n) + "\n" for n in pieces))
        run([cli, "-c", os.path.join(REF, cz), "-l", lst, "-m", os.path.join(cli_dir, "list.mlf")])
        run([cli, "-c", os.path.join(REF, cz), "-l", lst])          # one .rec next to each input
        for name in pieces:
            shutil.copyfile(os.path.join(sub, name[:-4] + ".rec"), os.path.join(cli_dir, name[:-4] + ".rec"))
        run([cli, "-c", os.path.join(REF, cz), "-l", lst, "-t", "par"])   # one-column list, params/suffix
        shutil.copyfile(os.path.join(sub, "utt_c.mel"), os.path.join(cli_dir, "utt_c.mel"))
        for name in pieces:                                              # posteriors of each piece
            run([cli, "-c", os.path.join(REF, cz), "-i", os.path.join(sub, name), "-t", "post",
                 "-o", os.path.join(cli_dir, name[:-4] + ".lop")])
    print("cli goldens ok")

    synth = {}
    for name, nb, hid, nout, seed, lens in SYNTH_CASES:
        with tempfile.TemporaryDirectory() as td:
            nets = modelgen.write_model_dir(td, nb, hid, nout, seed=seed)
            synth[name + "/digest"] = np.frombuffer(bytes.fromhex(modelgen.nets_digest(nets)), np.uint8)
            t = ob.RefTraps(td, nb, bunch=5)
            tb = ob.RefTraps(td, nb, bunch=5, blas=True) if ob.ref_lib_path(True) else None
            mels, posts, posts_blas = [], [], []
            for i, n in enumerate(lens):
                mel = modelgen.synth_mel(n, nb, seed=1000 * seed + i)
                mels.append(mel)
                posts.append(t.process_offline(mel))
                if tb:
                    posts_blas.append(tb.process_offline(mel))
            synth[name + "/dims"] = np.array([nb, hid, nout, seed], np.int32)
            synth[name + "/off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            synth[name + "/mel"] = np.concatenate(mels)
            synth[name + "/post"] = np.concatenate(posts)
            if tb:
                synth[name + "/post_blas"] = np.concatenate(posts_blas)
        print("synth", name, lens, "ok")
    np.savez_compressed(os.path.join(GOLD, "synth.npz"), **synth)


if __name__ == "__main__":
    main()
