"""The oracle (oracle/lcrc_oracle.c) against the REAL reference's outputs.

CPU only.  Pins the checker itself: golden posteriors dumped by the reference
CLI compiled from /root/reference (tests/golden, tools/make_golden.py), the
reference's own Traps/NeuralNet objects in-process (oracle/_ref, when built),
and known-answer values of the FEXP bit trick (fexp.h:14-21).
"""
import os
import struct

import numpy as np
import pytest

from phnrec_amd import modelgen
from tests.util import GOLD, model_dir, read_htk

REF_ROOT = "/root/reference"


def _model(system):
    p = model_dir(system)
    if p is None and os.path.isdir(os.path.join(REF_ROOT, system)):
        p = os.path.join(REF_ROOT, system)
    return p


def _norm_mel(ob, system):
    mel = read_htk(os.path.join(GOLD, system, "test.mel"))
    if modelgen.SYSTEMS[system]["sent_mean_norm"]:
        mel = ob.sentence_mean_norm(mel)
    return mel


# ---- FEXP known answers -------------------------------------------------------

def _fexp_py(y):
    """Independent pure-Python statement of fexp.h:14-21 with lo word 0."""
    t = (1048576.0 / 0.69314718055994530942) * float(np.float32(y))
    i = int(t)  # trunc toward zero
    hi = (i + 1072693248 - 60801) & 0xFFFFFFFF
    d = struct.unpack("<d", struct.pack("<II", 0, hi))[0]
    return np.float32(d)


def test_fexp_known_answers(oracle_mod):
    ob = oracle_mod
    # FEXP(0) = hi word 0x3FEF1281 -> 0.9708...; exact powers of two land on hi + k*2^20
    assert ob.fexp(0.0) == _fexp_py(0.0)
    assert ob.fexp(0.0) == 0.9710078239440918   # 2^-1 * (1 + 0xF1281 / 2^20)
    for y in [0.0, -0.0, 1.0, -1.0, 0.5, -0.3, 10.0, -10.0, -50.0, -87.0, -100.0, 80.0, 1e-9, -1e-9]:
        assert ob.fexp(y) == _fexp_py(y), y
    rng = np.random.default_rng(0)
    for y in rng.uniform(-90, 20, 2000).astype(np.float32):
        got, want = ob.fexp(y), _fexp_py(y)
        assert got == want, (y, got, want)
        assert abs(got - np.exp(np.float64(y))) <= 0.07 * np.exp(np.float64(y)) + 1e-38


def test_fexp_sigmoid_and_softmax(oracle_mod):
    ob = oracle_mod
    for x in [-30.0, -5.0, -1.0, 0.0, 0.25, 3.0, 20.0]:
        d = float(np.float32(x))
        t = (1048576.0 / 0.69314718055994530942) * (-d)
        hi = (int(t) + 1072693248 - 60801) & 0xFFFFFFFF
        e = struct.unpack("<d", struct.pack("<II", 0, hi))[0]
        assert ob.fexp_sigmoid(x) == np.float32(1.0 / (1.0 + e))
    v = np.array([1.0, 2.0, -3.0, 0.5, 2.0], np.float32)
    p = ob.fexp_softmax(v)
    e = np.array([_fexp_py(np.float32(a) - np.float32(2.0)) for a in v], np.float32)
    s = np.float32(0)
    for a in e:
        s = np.float32(s + a)
    assert np.array_equal(p, e * (np.float32(1.0) / s))
    assert abs(p.sum() - 1) < 1e-6


# ---- golden posteriors of the four shipped systems ----------------------------

@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_oracle_matches_reference_posteriors(oracle_mod, system):
    ob = oracle_mod
    mdir = _model(system)
    if mdir is None:
        pytest.skip("weights for %s not in this checkout" % system)
    spec = modelgen.SYSTEMS[system]
    o = ob.Oracle(mdir, spec["nbanks"])
    assert o.n_out == spec["n_out"]
    mel = _norm_mel(ob, system)
    lop = read_htk(os.path.join(GOLD, system, "test.lop"))
    post = o.posteriors(mel)
    # the reference build used for the goldens zeroes FEXP's low word, as the
    # oracle does: the restatement is BIT-EXACT, not merely within 1e-4
    assert np.array_equal(post, lop)
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_oracle_intermediates(oracle_mod, system):
    ob = oracle_mod
    spec = modelgen.SYSTEMS[system]
    o = ob.Oracle(_model(system), spec["nbanks"])
    mel = _norm_mel(ob, system)
    g = np.load(os.path.join(GOLD, system, "probe.npz"))
    rows = g["rows"]
    pr = o.posteriors_probe(mel)
    for key in ("in0", "in1", "p0", "p1", "g"):
        assert np.array_equal(pr[key][rows], g[key]), key


@pytest.mark.parametrize("system", ["PHN_CZ_SPDAT_LCRC_N1500"])
def test_streaming_equals_stateless(oracle_mod, system):
    ob = oracle_mod
    spec = modelgen.SYSTEMS[system]
    o = ob.Oracle(_model(system), spec["nbanks"])
    mel = _norm_mel(ob, system)[:120]
    a = o.posteriors(mel)
    b = o.process_offline(mel, bunch=5)
    assert np.array_equal(a, b)
    # Traps::GetDelay semantics (traps.cpp:199,215-217)
    o.reset()
    o.push(mel[:1], needed=False)
    assert o.delay() == 0
    o.push(mel[1:8], needed=False)
    assert o.delay() == 7


def test_synthetic_goldens(oracle_mod, tmp_path):
    """Reference outputs on seeded synthetic models incl. 1..40-frame utterances."""
    ob = oracle_mod
    g = np.load(os.path.join(GOLD, "synth.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    assert "tiny" in names
    for name in names:
        nb, hid, nout, seed = [int(v) for v in g[name + "/dims"]]
        d = tmp_path / name
        nets = modelgen.write_model_dir(str(d), nb, hid, nout, seed=seed)
        assert bytes(g[name + "/digest"]).hex() == modelgen.nets_digest(nets), \
            "modelgen no longer reproduces the weights the goldens were made with"
        o = ob.Oracle(str(d), nb)
        off = g[name + "/off"]
        post = o.posteriors_batch(g[name + "/mel"], off)
        assert np.array_equal(post, g[name + "/post"]), name
        if name + "/post_blas" in g.files:   # MKL sgemv path of the reference
            assert np.abs(post - g[name + "/post_blas"]).max() < 1e-4
        # regenerated inputs are the committed ones
        for i in range(len(off) - 1):
            mel = modelgen.synth_mel(int(off[i + 1] - off[i]), nb, seed=1000 * seed + i)
            assert np.array_equal(mel, g[name + "/mel"][off[i]:off[i + 1]])


# ---- in-process against the reference classes (only where oracle/_ref exists) --

def _need_ref(ob, blas=False):
    if ob.ref_lib_path(blas) is None:
        pytest.skip("oracle/_ref not built")


def test_against_reference_traps_object(oracle_mod, tmp_path):
    ob = oracle_mod
    _need_ref(ob)
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 96, 30, seed=5)
    o = ob.Oracle(d, 15)
    t = ob.RefTraps(d, 15, bunch=5)
    for n in (1, 3, 15, 16, 29, 77):
        mel = modelgen.synth_mel(n, 15, seed=n)
        assert np.array_equal(o.posteriors(mel), t.process_offline(mel)), n
    # bunch size changes grouping only
    t7 = ob.RefTraps(d, 15, bunch=7)
    mel = modelgen.synth_mel(50, 15, seed=3)
    assert np.array_equal(t7.process_offline(mel), o.posteriors(mel))


def test_against_reference_neuralnet_and_loaders(oracle_mod, tmp_path):
    ob = oracle_mod
    _need_ref(ob)
    d = str(tmp_path / "m")
    nets = modelgen.write_model_dir(d, 15, 50, 21, seed=9, ascii_too=True)
    w = os.path.join(d, "weights", "band0.weights")
    nrm = os.path.join(d, "norms", "band0.norms")
    bin_path = os.path.join(d, "weights", "band0.nbin")
    a = ob.Net(weights=w, norms=nrm)          # ASCII parse
    b = ob.Net(nbin=bin_path)                 # binary written by modelgen
    for key in ("W1", "W2", "b1", "b2", "mean", "dev"):
        assert np.array_equal(a.array(key), b.array(key)), key
    assert a.dims == (165, 50, 21)
    assert np.array_equal(a.array("W1")[:50, :165], nets["band0"]["w1"])
    # oracle's writer == reference reader, reference forward == oracle forward
    out = str(tmp_path / "resaved.nbin")
    a.save_nbin(out)
    assert open(out, "rb").read() == open(bin_path, "rb").read()
    r = ob.RefNet(w, nrm, bunch=5)
    assert r.dims == a.dims
    x = np.random.default_rng(1).standard_normal((23, 165)).astype(np.float32) * 3
    assert np.array_equal(r.forward(x), a.forward(x))


def test_blas_path_within_tolerance(oracle_mod, tmp_path):
    """north_star: posteriors match the reference BLAS CPU path within 1e-4."""
    ob = oracle_mod
    _need_ref(ob, blas=True)
    mdir = _model("PHN_CZ_SPDAT_LCRC_N1500")
    o = ob.Oracle(mdir, 15)
    mel = _norm_mel(ob, "PHN_CZ_SPDAT_LCRC_N1500")[:100]
    for bunch in (5, 32):     # sgemv regime and sgemm regime (nn.cpp:723)
        t = ob.RefTraps(mdir, 15, bunch=bunch, blas=True)
        assert np.abs(t.process_offline(mel) - o.posteriors(mel)).max() < 1e-4


def test_sentence_mean_norm(oracle_mod):
    ob = oracle_mod
    mel = read_htk(os.path.join(GOLD, "PHN_CZ_SPDAT_LCRC_N1500", "test.mel"))
    got = ob.sentence_mean_norm(mel)
    want = mel.copy()
    for b in range(mel.shape[1]):
        s = np.float32(0)
        for v in mel[:, b]:
            s = np.float32(s + v)
        want[:, b] += -np.float32(s * (np.float32(1.0) / np.float32(mel.shape[0])))
    assert np.array_equal(got, want)


def test_other_posterior_systems_vs_reference_goldens(oracle_mod, tmp_path):
    """traps_oracle.c (1BT_DCT / 1BT / 3BT, with and without Hamming window and C0) against what the
    reference's own Traps class produced on the same seeded synthetic models (tests/golden/systems.npz,
    tools/make_golden_systems.py): bit-exact, and again in-process when oracle/_ref is present"""
    from tools.make_golden_systems import CASES
    gold = np.load(os.path.join(GOLD, "systems.npz"))
    for name, system, nb, hid, nout, seed, kw, lens in CASES:
        d = str(tmp_path / name)
        modelgen.write_traps_dir(d, system, nb, hid, nout, seed=seed, **kw)
        o = oracle_mod.TrapsOracle(d, system, nb, kw.get("add_c0", True), kw.get("hamming", False))
        assert o.n_out == nout
        assert o.n_band_nets == (0 if system == "1BT_DCT" else nb - 2 if system == "3BT" else nb)
        mel, off = gold[name + "/mel"], gold[name + "/off"]
        assert np.array_equal(mel, np.concatenate([modelgen.synth_mel(n, nb, seed=1000 * seed + i)
                                                   for i, n in enumerate(lens)]))
        got = o.posteriors_batch(mel, off)
        assert np.array_equal(got, gold[name + "/post"]), name
        if oracle_mod.ref_lib_path(False):
            t = oracle_mod.RefTraps(d, nb, bunch=3, system=system, add_c0=kw.get("add_c0", True),
                                    hamming=kw.get("hamming", False))
            a, b = int(off[0]), int(off[1])
            assert np.array_equal(t.process_offline(mel[a:b]), got[a:b])


def test_general_geometry_vs_reference_goldens(oracle_mod, tmp_path):
    """The geometries the reference's Traps accepts but no shipped model uses -- posteriors/length other than 31 (odd and
    even: for even lengths the reference's LCRC walk over be_mat drifts across the band rows, traps.cpp:296-306, and so
    does the restatement), LCRC without C0 or with another number of coefficients per band -- traps_oracle.c against
    what the reference's own class produced (tests/golden/geometry.npz): bit-exact, and in-process when oracle/_ref is
    present.  At the shipped geometry the run-time-geometry restatement equals lcrc_oracle.c bit for bit."""
    from tools.make_golden_systems import GEOM_CASES, write_geometry_model
    gold = np.load(os.path.join(GOLD, "geometry.npz"))
    for name, system, nb, hid, nout, seed, kw, lens, trap_len in GEOM_CASES:
        d = str(tmp_path / name)
        write_geometry_model(d, system, nb, hid, nout, seed, kw, trap_len)
        o = oracle_mod.TrapsOracle(d, system, nb, kw.get("add_c0", True), kw.get("hamming", False), trap_len=trap_len)
        assert o.n_out == nout
        mel, off = gold[name + "/mel"], gold[name + "/off"]
        assert np.array_equal(mel, np.concatenate([modelgen.synth_mel(n, nb, seed=1000 * seed + i)
                                                   for i, n in enumerate(lens)]))
        got = o.posteriors_batch(mel, off)
        assert np.array_equal(got, gold[name + "/post"]), name
        if oracle_mod.ref_lib_path(False):
            t = oracle_mod.RefTraps(d, nb, bunch=3, system=system, add_c0=kw.get("add_c0", True),
                                    hamming=kw.get("hamming", False), trap_len=trap_len)
            a, b = int(off[0]), int(off[1])
            assert np.array_equal(t.process_offline(mel[a:b]), got[a:b])
    d = model_dir("PHN_CZ_SPDAT_LCRC_N1500")
    mel = modelgen.synth_mel(70, 15, seed=9)
    assert np.array_equal(oracle_mod.TrapsOracle(d, "LCRC", 15).posteriors(mel), oracle_mod.Oracle(d, 15).posteriors(mel))


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_phndec_oracle_reproduces_reference_label_files(oracle_mod, system):
    """phndec_oracle.c on the logarithm of the reference's posterior dump == the reference's .rec
    (labels and times exact, scores equal at the printed precision)"""
    lop = read_htk(os.path.join(GOLD, system, "test.lop"))
    names = [l.rstrip("\r\n") for l in open(os.path.join(model_dir(system), "dicts", "phonemes"))]
    labs = oracle_mod.phndec(np.log(lop), len(names), 3, 40, modelgen.SYSTEMS[system]["wpenalty"])
    gold = [l.split() for l in open(os.path.join(GOLD, system, "test.rec"))]
    mine = [["%d00000" % s if s else "000000", "%d00000" % e, names[p], "%f" % sc] for s, e, p, sc in labs]
    assert [m[:3] for m in mine] == [g[:3] for g in gold]
    assert max(abs(float(m[3]) - float(g[3])) for m, g in zip(mine, gold)) < 1e-5
    assert oracle_mod.phndec(np.zeros((0, lop.shape[1]), np.float32), len(names)) == []
