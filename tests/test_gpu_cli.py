"""End-to-end through the drop-in CLI on the GPU: waveform -> labels / posteriors, against
what the reference CLI wrote for the same inputs (tests/golden)."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLD, ROOT, model_dir, read_htk

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
CZ, EN = "PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"
HU, RU = "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500"


def run(*args, env=None):
    """one CLI run that must succeed.  Runs on ONE GPU without -F / -E are what most tests use as "the host front-end": they
    get PHNREC_NO_AUTO_E=1 (since round 5 a list of ~100 files or more would take the GPU front-end by itself there as well
    -- the same bytes, but then those tests would compare the GPU front-end with itself); tests of the automatic choice
    pass their own environment (AUTO: "1" switches this default off)."""
    e = dict(os.environ)
    a = [str(x) for x in args]
    one_gpu = "-g" not in a or a[a.index("-g") + 1] == "1"
    if one_gpu and "-F" not in a and "-E" not in a and not (env or {}).get("AUTO"):
        e["PHNREC_NO_AUTO_E"] = "1"
    e.update({k: v for k, v in (env or {}).items() if k != "AUTO"})
    p = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, env=e)
    assert p.returncode == 0, p.stderr
    return p


def _labels_match(mine_path, gold_path):
    mine = [l.split() for l in open(mine_path) if len(l.split()) == 4]
    gold = [l.split() for l in open(gold_path) if len(l.split()) == 4]
    assert [m[:3] for m in mine] == [g[:3] for g in gold], "labels / times differ from the reference"
    assert max(abs(float(m[3]) - float(g[3])) for m, g in zip(mine, gold)) < 1e-2


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_bundled_utterance_end_to_end(system, tmp_path):
    """the reference's own smoke test (test.sh): phnrec -c DIR -i test.raw -o test.rec"""
    out = tmp_path / "t.rec"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-o", out)
    _labels_match(out, os.path.join(GOLD, "rec", system + ".rec"))       # the SHIPPED golden
    _labels_match(out, os.path.join(GOLD, system, "test.rec"))
    lop = tmp_path / "t.lop"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-t", "post", "-o", lop)
    got, want = read_htk(str(lop)), read_htk(os.path.join(GOLD, system, "test.lop"))
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_split_f16_flag(system, tmp_path):
    """-H (lcrc_set_arithmetic(LCRC_ARITH_SPLIT_F16)): the shipped label file and the reference's posterior dump at
    the same bar as without the flag; combined with the GPU front-end and decoder"""
    out = tmp_path / "t.rec"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-o", out, "-H")
    _labels_match(out, os.path.join(GOLD, "rec", system + ".rec"))
    lop = tmp_path / "t.lop"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-t", "post", "-o", lop, "-H")
    got, want = read_htk(str(lop)), read_htk(os.path.join(GOLD, system, "test.lop"))
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
    out2 = tmp_path / "t2.rec"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-o", out2, "-H", "-F", "-D")
    _labels_match(out2, os.path.join(GOLD, "rec", system + ".rec"))


@pytest.mark.parametrize("system", [CZ, EN])
def test_params_in_posteriors_out(system, tmp_path):
    """-s par: the hot path in its most direct CLI form (srec.cpp:1136-1145: the input is the reference's own
    HTK parameter dump, the front-end is skipped) -> -t post vs the reference's posterior dump, -t str vs its
    label file; and -s post -> -t str (decoder only) from the reference's posterior dump"""
    mel = os.path.join(GOLD, system, "test.mel")
    lop = tmp_path / "t.lop"
    run("-c", model_dir(system), "-s", "par", "-i", mel, "-t", "post", "-o", lop)
    got, want = read_htk(str(lop)), read_htk(os.path.join(GOLD, system, "test.lop"))
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
    rec = tmp_path / "t.rec"
    run("-c", model_dir(system), "-s", "par", "-i", mel, "-t", "str", "-o", rec)
    _labels_match(rec, os.path.join(GOLD, system, "test.rec"))
    rec2 = tmp_path / "t2.rec"
    run("-c", model_dir(system), "-s", "post", "-i", os.path.join(GOLD, system, "test.lop"), "-o", rec2)
    _labels_match(rec2, os.path.join(GOLD, system, "test.rec"))
    # a list of parameter files into an MLF
    lst = tmp_path / "l.txt"
    lst.write_text("%s\n%s\n" % (mel, mel))
    mlf = tmp_path / "o.mlf"
    run("-c", model_dir(system), "-s", "par", "-l", lst, "-m", mlf)
    lines = mlf.read_text().splitlines()
    gold = [l.split() for l in open(os.path.join(GOLD, system, "test.rec")) if len(l.split()) == 4]
    body = [l.split() for l in lines if len(l.split()) == 4]
    assert lines[0] == "#!MLF!#" and len(body) == 2 * len(gold)
    # (an MLF prints a zero time as "0", a label file as "000000": srec.cpp:137-161 vs phndec.cpp:230)
    key = lambda rows: [(int(r[0]), int(r[1]), r[2]) for r in rows]
    assert key(body[:len(gold)]) == key(gold) and key(body[len(gold):]) == key(gold)


def test_posterior_softening_through_the_cli(tmp_path):
    """posteriors/softening_func = log / igor / gmm_bypass (srec.cpp:164-176,1062-1070): `-t post` dumps against
    the REFERENCE CLI's dumps for the same configs (tools/make_golden_softening.py); the functions run in the
    kernel's epilogue, the dump is the kernel's big-endian output written as is"""
    import shutil
    from tools.make_golden_softening import CASES
    raw = tmp_path / "x.raw"
    raw.write_bytes(open(os.path.join(GOLD, "test.raw"), "rb").read()[:20000])
    for name, value in CASES.items():
        d = tmp_path / name
        shutil.copytree(model_dir(CZ), d)
        cfg, section = (d / "config").read_text().splitlines(True), None
        for i, line in enumerate(cfg):
            if line.startswith("["):
                section = line.strip()
            if section == "[posteriors]" and line.startswith("softening_func="):
                cfg[i] = "softening_func=%s\n" % value
        (d / "config").write_text("".join(cfg))
        want = read_htk(os.path.join(GOLD, "cli", "soft_%s.lop" % name))
        for extra in ([], ["-F"]):
            lop = tmp_path / (name + ".lop")
            run("-c", d, "-i", raw, "-t", "post", "-o", lop, *extra)
            got = read_htk(str(lop))
            assert got.shape == want.shape
            # log-type functions turn the RELATIVE error of a posterior (up to ~5e-5 at p ~ 1e-10, far inside
            # the 1e-4 absolute bar) into an absolute one
            assert np.abs(got - want).max() < 1e-3, (name, extra, np.abs(got - want).max())


def test_file_list_batched_over_the_gpu(tmp_path):
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    pieces = {"utt_a": raw, "utt_b": raw[:20000], "utt_c": raw[:3000]}
    data = tmp_path / "data"
    data.mkdir()
    for n, blob in pieces.items():
        (data / (n + ".raw")).write_bytes(blob)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s\n" % (data / (n + ".raw")) for n in pieces))
    mlf = tmp_path / "out.mlf"
    p = run("-c", model_dir(CZ), "-l", lst, "-m", mlf, "-g", 1, env={"PHNREC_STATS": "1"})
    assert "frames=%d" % (747 + 123 + 17) in p.stderr
    gold = open(os.path.join(GOLD, "cli", "list.mlf")).read().splitlines()
    mine = mlf.read_text().splitlines()
    assert len(mine) == len(gold)
    for a, b in zip(mine, gold):
        pa, pb = a.split(), b.split()
        if len(pb) == 4:
            assert pa[:3] == pb[:3] and abs(float(pa[3]) - float(pb[3])) < 1e-2
        else:
            assert a == b                       # "#!MLF!#", "*/utt_a.rec", "."
    # posteriors of every piece (one launch for the three utterances vs one launch each)
    run("-c", model_dir(CZ), "-l", lst, "-t", "post")
    for n in pieces:
        got = read_htk(str(data / (n + ".lop")))
        want = read_htk(os.path.join(GOLD, "cli", n + ".lop"))
        assert np.abs(got - want).max() < 1e-4, n
    run("-c", model_dir(CZ), "-l", lst, "-t", "post", "-b", 1)    # one utterance per launch
    for n in pieces:
        got = read_htk(str(data / (n + ".lop")))
        want = read_htk(os.path.join(GOLD, "cli", n + ".lop"))
        assert np.abs(got - want).max() < 1e-4, n


def test_gpu_front_end_flag(tmp_path):
    """-F: waveform -> posteriors entirely on the device; labels still equal the reference's"""
    out = tmp_path / "t.rec"
    run("-c", model_dir(CZ), "-F", "-i", os.path.join(GOLD, "test.raw"), "-o", out)
    _labels_match(out, os.path.join(GOLD, "rec", CZ + ".rec"))
    lop = tmp_path / "t.lop"
    run("-c", model_dir(EN), "-F", "-i", os.path.join(GOLD, "test.raw"), "-t", "post", "-o", lop)
    assert np.abs(read_htk(str(lop)) - read_htk(os.path.join(GOLD, EN, "test.lop"))).max() < 1e-4
    # -t par keeps the host front-end (its dump is bit-identical to the reference's)
    mel = tmp_path / "t.mel"
    run("-c", model_dir(CZ), "-F", "-i", os.path.join(GOLD, "test.raw"), "-t", "par", "-o", mel)
    assert open(mel, "rb").read() == open(os.path.join(GOLD, CZ, "test.mel"), "rb").read()


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_gpu_energies_flag_is_the_host_front_end_bit_for_bit(system, tmp_path):
    """-E: decode, window, FFT, power spectrum and bank sums on the GPU, ln() (this host's libm) and the normalisations on
    the host.  The GPU part repeats the reference's operations in the reference's order, so the features ARE the host
    front-end's and the posterior dump must equal the default mode's BYTE FOR BYTE (the CLI pins the fused kernels, whose
    bits do not depend on how frames are batched); A-law too; labels equal the shipped ones; `-t par` keeps the host
    front-end."""
    raw = os.path.join(GOLD, "test.raw")
    a, b = tmp_path / "host.lop", tmp_path / "e.lop"
    run("-c", model_dir(system), "-i", raw, "-t", "post", "-o", a)
    run("-c", model_dir(system), "-i", raw, "-t", "post", "-o", b, "-E")
    assert open(a, "rb").read() == open(b, "rb").read()
    assert np.abs(read_htk(str(b)) - read_htk(os.path.join(GOLD, system, "test.lop"))).max() < 1e-4
    if system == CZ:
        run("-c", model_dir(system), "-w", "alaw", "-i", raw, "-t", "post", "-o", a)
        run("-c", model_dir(system), "-w", "alaw", "-i", raw, "-t", "post", "-o", b, "-E")
        assert open(a, "rb").read() == open(b, "rb").read()
        mel = tmp_path / "t.mel"
        run("-c", model_dir(CZ), "-E", "-i", raw, "-t", "par", "-o", mel)
        assert open(mel, "rb").read() == open(os.path.join(GOLD, CZ, "test.mel"), "rb").read()
    out = tmp_path / "t.rec"
    run("-c", model_dir(system), "-i", raw, "-o", out, "-E")
    _labels_match(out, os.path.join(GOLD, "rec", system + ".rec"))
    out2 = tmp_path / "t2.rec"
    run("-c", model_dir(system), "-i", raw, "-o", out2, "-E", "-D")
    _labels_match(out2, os.path.join(GOLD, "rec", system + ".rec"))


def test_two_gpus_called_as_the_reference_take_the_gpu_front_end(tmp_path):
    """`phnrec -g 2 -l ... -m ...` without -F / -E switches the GPU front-end on by itself (the host front-end would feed 1.4
    GPUs) -- -F where its ln() is this host's libm's own sequence (glibc: here), else -E --: its stage-1 CPU time falls to a
    fraction, the MLF stays the one `-g 1` (host front-end) writes, byte for byte; PHNREC_NO_AUTO_E=1 keeps the host front-end"""
    lst = _make_list(tmp_path, "hu", 150, seed=23)
    one, two, two_host = tmp_path / "g1.mlf", tmp_path / "g2.mlf", tmp_path / "g2h.mlf"
    a = run("-c", model_dir(HU), "-l", lst, "-m", one, env={"PHNREC_STATS": "1"})
    b = run("-c", model_dir(HU), "-l", lst, "-m", two, "-g", 2, env={"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0"})
    c = run("-c", model_dir(HU), "-l", lst, "-m", two_host, "-g", 2,
            env={"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0", "PHNREC_NO_AUTO_E": "1"})
    assert one.read_text() == two.read_text() == two_host.read_text()

    def stage1(p):
        line = [l for l in p.stderr.splitlines() if l.startswith("phnrec: files=")][-1]
        return float(line.split("stage1=")[1].split()[0])
    assert stage1(b) < 0.5 * stage1(a) and stage1(c) > 0.5 * stage1(a)
    assert " mode=F,auto " in b.stderr and " mode=host " in c.stderr and " mode=host " in a.stderr
    # ... and so does a list of this length on ONE GPU when nothing holds it back (the helper's default does, above)
    auto1 = tmp_path / "g1auto.mlf"
    d = run("-c", model_dir(HU), "-l", lst, "-m", auto1, env={"PHNREC_STATS": "1", "AUTO": "1"})
    assert " mode=F,auto " in d.stderr and auto1.read_text() == one.read_text() and stage1(d) < 0.5 * stage1(a)
    # a handful of files keep the host front-end
    few = tmp_path / "few.scp"
    few.write_text("".join(l + "\n" for l in lst.read_text().split()[:5]))
    f = run("-c", model_dir(HU), "-l", few, "-m", tmp_path / "few.mlf", env={"PHNREC_STATS": "1", "AUTO": "1"})
    assert " mode=host " in f.stderr


def test_host_whose_libm_is_not_glibcs_takes_the_energies_road(tmp_path):
    """PHNREC_LN_FORM=0 stands in for a host whose logf matches neither of glibc's sequences: a list over two GPUs then selects
    -E by itself (its ln() is the host's own libm whatever that is) -- mode E,auto, and E+D,auto from four --, the MLF stays
    the host front-end's byte for byte; -F given explicitly still runs, with log() in double (labels equal, features within
    one ulp: the posterior dump within 1e-4 of the host front-end's)"""
    lst = _make_list(tmp_path, "hu", 120, seed=37)
    one, two, four = tmp_path / "g1.mlf", tmp_path / "g2.mlf", tmp_path / "g4.mlf"
    run("-c", model_dir(HU), "-l", lst, "-m", one)
    b = run("-c", model_dir(HU), "-l", lst, "-m", two, "-g", 2,
            env={"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0", "PHNREC_LN_FORM": "0"})
    c = run("-c", model_dir(HU), "-l", lst, "-m", four, "-g", 4,
            env={"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0,0,0", "PHNREC_LN_FORM": "0"})
    assert " mode=E,auto " in b.stderr and " mode=E+D,auto " in c.stderr
    assert one.read_text() == two.read_text() == four.read_text()
    raw = os.path.join(GOLD, "test.raw")
    a, f = tmp_path / "host.lop", tmp_path / "f.lop"
    run("-c", model_dir(HU), "-i", raw, "-t", "post", "-o", a)
    run("-c", model_dir(HU), "-i", raw, "-t", "post", "-o", f, "-F", env={"PHNREC_LN_FORM": "0"})
    assert np.abs(read_htk(str(a)) - read_htk(str(f))).max() < 1e-4


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_gpu_front_end_flag_is_the_host_front_end_bit_for_bit(system, tmp_path):
    """-F with ln() taken as THIS host's libm takes it (lcrc_frontend_set_ln: glibc's logf sequence, in the build the CLI found
    the process's logf to match): the whole front-end on the GPU -- decode, window, FFT, bank sums, ln(), sentence mean in
    the reference's order -- gives the host front-end's features, so the posterior dump equals the default mode's BYTE FOR
    BYTE, as -E's does; A-law too; over a list (several launches, two logical GPUs, device decoder) the MLF as well."""
    raw = os.path.join(GOLD, "test.raw")
    a, b = tmp_path / "host.lop", tmp_path / "f.lop"
    run("-c", model_dir(system), "-i", raw, "-t", "post", "-o", a)
    run("-c", model_dir(system), "-i", raw, "-t", "post", "-o", b, "-F")
    assert open(a, "rb").read() == open(b, "rb").read()
    if system == CZ:
        run("-c", model_dir(system), "-w", "alaw", "-i", raw, "-t", "post", "-o", a)
        run("-c", model_dir(system), "-w", "alaw", "-i", raw, "-t", "post", "-o", b, "-F")
        assert open(a, "rb").read() == open(b, "rb").read()
    lst = _make_list(tmp_path, system[4:6].lower(), 60, seed=31, rate=16000 if system == EN else 8000)
    host, fe, fed = tmp_path / "host.mlf", tmp_path / "f.mlf", tmp_path / "fd.mlf"
    run("-c", model_dir(system), "-l", lst, "-m", host, "-b", 3000)
    run("-c", model_dir(system), "-l", lst, "-m", fe, "-b", 3000, "-F", "-g", 2, env={"PHNREC_DEVICE_MAP": "0,0"})
    run("-c", model_dir(system), "-l", lst, "-m", fed, "-F", "-D")
    assert host.read_text() == fe.read_text() == fed.read_text()


def test_gpu_ln_on_this_boxs_host_and_gpu():
    """the GPU front-end's ln() in the form this host's libm matches, against that libm's logf over every positive float and a
    sixteenth of the other bit patterns (phnrec --selftest-gpu-ln)"""
    p = subprocess.run([BIN, "--selftest-gpu-ln"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and ": 0 of " in p.stdout, p.stdout + p.stderr


def test_vector_ln_on_this_boxs_host():
    """the host's ln() (host/veclog.cpp) against libm's logf over every non-negative float, on the GPU box's own CPU and libm
    (the CPU suite runs the same check in the build container)"""
    p = subprocess.run([BIN, "--selftest-ln"], capture_output=True, text=True)
    assert p.returncode == 0 and " 0 of 2^31" in p.stdout, p.stdout + p.stderr


def test_four_gpus_decode_on_the_device_by_themselves(tmp_path):
    """`phnrec -g 4 -l ... -m ...` (four logical GPUs on this box's one) switches the device decoder on by itself: no
    Viterbi time on the host, the MLF the one `-g 1` writes with the host decoder, byte for byte; PHNREC_NO_AUTO_D=1 keeps
    the host decoder; -b still decides the launch size"""
    lst = _make_list(tmp_path, "hu", 150, seed=29)
    one, four, four_host, four_b = tmp_path / "g1.mlf", tmp_path / "g4.mlf", tmp_path / "g4h.mlf", tmp_path / "g4b.mlf"
    env4 = {"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0,0,0"}
    a = run("-c", model_dir(HU), "-l", lst, "-m", one, "-F", env={"PHNREC_STATS": "1"})
    b = run("-c", model_dir(HU), "-l", lst, "-m", four, "-F", "-g", 4, env=env4)
    c = run("-c", model_dir(HU), "-l", lst, "-m", four_host, "-F", "-g", 4, env=dict(env4, PHNREC_NO_AUTO_D="1"))
    d = run("-c", model_dir(HU), "-l", lst, "-m", four_b, "-g", 4, "-b", 5000, env=env4)       # (no -F: the -E road + decoder)
    assert one.read_text() == four.read_text() == four_host.read_text()
    host = tmp_path / "host.mlf"
    run("-c", model_dir(HU), "-l", lst, "-m", host)
    assert host.read_text() == four_b.read_text()

    def viterbi(p):
        line = [l for l in p.stderr.splitlines() if l.startswith("phnrec: files=")][-1]
        return float(line.split("viterbi=")[1].split(")")[0])
    assert viterbi(a) > 0 and viterbi(c) > 0 and viterbi(b) == 0 and viterbi(d) == 0


def test_gpu_energies_flag_on_a_list(tmp_path):
    """-E over a list (150 synthetic files, several launches, two logical GPUs): the MLF equals the default mode's byte
    for byte, and so does every posterior dump"""
    lst = _make_list(tmp_path, "hu", 150, seed=21)
    host, en = tmp_path / "host.mlf", tmp_path / "e.mlf"
    run("-c", model_dir(HU), "-l", lst, "-m", host, "-b", 3000)
    run("-c", model_dir(HU), "-l", lst, "-m", en, "-b", 3000, "-E", "-g", 2, env={"PHNREC_DEVICE_MAP": "0,0"})
    assert host.read_text() == en.read_text()
    names = lst.read_text().split()[:12]
    sub = tmp_path / "sub.scp"
    sub.write_text("".join("%s %s.h.lop\n" % (n, n) for n in names))
    run("-c", model_dir(HU), "-l", sub, "-t", "post")
    sub.write_text("".join("%s %s.e.lop\n" % (n, n) for n in names))
    run("-c", model_dir(HU), "-l", sub, "-t", "post", "-E")
    for n in names:
        assert open(n + ".h.lop", "rb").read() == open(n + ".e.lop", "rb").read(), n


def test_default_system_1bt_dct_end_to_end(tmp_path):
    """posteriors/system=1BT_DCT (the schema default, srec.cpp:69) through the CLI on the bundled utterance,
    against what the reference CLI wrote for the same synthetic model (tools/make_golden_systems.py)"""
    from phnrec_amd import modelgen
    from tools.make_golden_systems import CLI_CASE
    c = dict(CLI_CASE)
    d = str(tmp_path / "model")
    modelgen.write_traps_dir(d, c.pop("system"), c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"),
                             seed=c.pop("seed"), **c)
    raw = os.path.join(GOLD, "test.raw")
    lop = tmp_path / "t.lop"
    run("-c", d, "-i", raw, "-t", "post", "-o", lop)
    got, want = read_htk(str(lop)), read_htk(os.path.join(GOLD, "systems", "1bt_dct.lop"))
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
    rec = tmp_path / "t.rec"
    run("-c", d, "-i", raw, "-o", rec)
    _labels_match(rec, os.path.join(GOLD, "systems", "1bt_dct.rec"))
    rec2 = tmp_path / "t2.rec"
    run("-c", d, "-i", raw, "-o", rec2, "-F")           # GPU front-end feeds these systems as well
    _labels_match(rec2, os.path.join(GOLD, "systems", "1bt_dct.rec"))


def test_sentence_maximum_normalisations(tmp_path):
    """offlinenorm/sent_max_norm and sent_chmax_norm (srec.cpp:1547-1587; no shipped config sets them) against the
    posterior dumps the reference CLI wrote with them -- sent_max_norm as the reference computes it: its loop ends up
    subtracting column 0's maximum everywhere.  With -F they are refused (host front-end only)."""
    from phnrec_amd import modelgen
    from tools.make_golden_systems import NORM_CLI_MODEL, NORM_CLI_CASES
    raw = os.path.join(GOLD, "test.raw")
    dumps = {}
    for name, cfg in NORM_CLI_CASES:
        c = dict(NORM_CLI_MODEL)
        d = str(tmp_path / name)
        modelgen.write_model_dir(d, c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"), seed=c.pop("seed"), **cfg)
        lop = tmp_path / (name + ".lop")
        run("-c", d, "-i", raw, "-t", "post", "-o", lop)
        got, want = read_htk(str(lop)), read_htk(os.path.join(GOLD, "systems", "lcrc_%s.lop" % name))
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4, name
        dumps[name] = got
        p = subprocess.run([BIN, "-c", d, "-i", raw, "-t", "post", "-o", str(lop), "-F"], capture_output=True, text=True)
        assert p.returncode != 0 and "host front-end" in p.stderr
        lop_e = tmp_path / (name + "_E.lop")                # -E: the normalisations run on the host: accepted, same bytes
        run("-c", d, "-i", raw, "-t", "post", "-o", lop_e, "-E")
        assert open(lop_e, "rb").read() == open(lop, "rb").read(), name
    assert np.abs(dumps["maxnorm"] - dumps["chmaxnorm"]).max() > 1e-3        # (they do differ)
    assert np.array_equal(dumps["maxnorm"], dumps["bothmax"])                # the global form overrides the per-channel one


def test_fuzzed_lists_and_pipeline_settings():
    """tools/fuzz_cli.py with a fixed seed: random lists (empty files, files shorter than a frame, exactly one frame, up to
    20 s, sometimes an unreadable name) through random batch sizes, logical GPU counts, host thread counts, the five modes
    (host front-end, -E, -E -D, -F, -F -D), contexts per GPU, launch order and decoder overlap on / off -- every configuration
    writes the MLF (or fails where) the plain host-front-end run does, byte for byte, within the time limit"""
    from tools import fuzz_cli
    assert fuzz_cli.fuzz(seed=20261004, n_lists=3, log=lambda *a: None) == 45


def test_lcrc_at_another_length_end_to_end(tmp_path):
    """posteriors/length=21, add_c0=false (a geometry the reference accepts but no shipped model uses) through the CLI on
    the bundled utterance, against the posterior dump the reference CLI wrote for the same synthetic model; the list
    form and the GPU front-end feed the general kernels as well"""
    from phnrec_amd import modelgen
    from tools.make_golden_systems import GEOM_CLI_CASE
    c = dict(GEOM_CLI_CASE)
    d = str(tmp_path / "model")
    modelgen.write_model_dir(d, c.pop("nbanks"), c.pop("hidden"), c.pop("n_out"), seed=c.pop("seed"), **c)
    raw = os.path.join(GOLD, "test.raw")
    want = read_htk(os.path.join(GOLD, "systems", "lcrc_len21.lop"))
    for k, extra in enumerate(((), ("-F",))):
        lop = tmp_path / ("t%d.lop" % k)
        run("-c", d, "-i", raw, "-t", "post", "-o", lop, *extra)
        got = read_htk(str(lop))
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
    data = tmp_path / "data"
    data.mkdir()
    blob = open(raw, "rb").read()
    for n, b in (("a", blob), ("b", blob[:9000])):
        (data / (n + ".raw")).write_bytes(b)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s\n" % (data / (n + ".raw")) for n in ("a", "b")))
    run("-c", d, "-l", lst, "-t", "post")
    assert np.abs(read_htk(str(data / "a.lop")) - want).max() < 1e-4
    assert read_htk(str(data / "b.lop")).shape == ((9000 // 2 - 200) // 80 + 1, want.shape[1])


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_decoder_on_the_gpu_flag(system, tmp_path):
    """-D: PhnDec runs on the device behind the posterior kernel; same label files as the reference's, for
    the single-file form, for the -F form and for a batched list into an MLF"""
    raw_path = os.path.join(GOLD, "test.raw")
    out = tmp_path / "t.rec"
    run("-c", model_dir(system), "-i", raw_path, "-o", out, "-D")
    _labels_match(out, os.path.join(GOLD, system, "test.rec"))
    out2 = tmp_path / "t2.rec"
    run("-c", model_dir(system), "-i", raw_path, "-o", out2, "-D", "-F")
    _labels_match(out2, os.path.join(GOLD, system, "test.rec"))
    if system != CZ:
        return
    raw = open(raw_path, "rb").read()
    pieces = {"utt_a": raw, "utt_b": raw[:20000], "utt_c": raw[:3000]}
    data = tmp_path / "data"
    data.mkdir()
    for n, blob in pieces.items():
        (data / (n + ".raw")).write_bytes(blob)
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s\n" % (data / (n + ".raw")) for n in pieces))
    host, dev = tmp_path / "host.mlf", tmp_path / "dev.mlf"
    run("-c", model_dir(CZ), "-l", lst, "-m", host)
    run("-c", model_dir(CZ), "-l", lst, "-m", dev, "-D", "-b", 800)      # several launches
    assert dev.read_text() == host.read_text()                            # same posteriors, same f32 additions


# ---- the reference's own remaining goldens (tests/golden/ref: data copied from /root/reference) ---------------------
REF = os.path.join(GOLD, "ref")


def _es_model(tmp_path):
    """test/PHN_ES of the reference = the HU weights, norms and windows under another phoneme list and config
    (source/format=lin16): assembled here from the committed HU model and the two committed ES text files"""
    import shutil
    d = tmp_path / "PHN_ES"
    shutil.copytree(model_dir(HU), d)
    shutil.copy(os.path.join(REF, "PHN_ES", "config"), d / "config")
    shutil.copy(os.path.join(REF, "PHN_ES", "dicts", "phonemes"), d / "dicts" / "phonemes")
    return d


@pytest.mark.parametrize("flags", [(), ("-F",), ("-F", "-D"), ("-E",)])
def test_reference_list_golden_8580(flags, tmp_path):
    """/root/reference/test: `phnrec -c PHN_ES -l lsit.txt -m test` run inside that directory.  lsit.txt:1 is a
    one-column line with a bare file name, so the MLF entry is "8580.rec" (no "*/": ChangeFilePath leaves a name
    without separator alone, srec.cpp:1424-1436); 8580.wav's 44-byte RIFF header is consumed as 22 samples
    (srec.cpp:1384-1422 reads the whole file).  Names, labels and times exact, scores within 1e-2;
    also the single-file form against test/8580.rec."""
    import shutil
    es = _es_model(tmp_path)
    work = tmp_path / "work"
    work.mkdir()
    shutil.copy(os.path.join(REF, "8580.wav"), work / "8580.wav")
    shutil.copy(os.path.join(REF, "lsit.txt"), work / "lsit.txt")
    e = dict(os.environ)
    p = subprocess.run([BIN, "-c", str(es), "-l", "lsit.txt", "-m", "out.mlf"] + list(flags), cwd=work,
                       capture_output=True, text=True, env=e)
    assert p.returncode == 0, p.stderr
    mine = (work / "out.mlf").read_text().splitlines()
    gold = open(os.path.join(REF, "8580.mlf")).read().splitlines()
    assert len(mine) == len(gold)
    for a, b in zip(mine, gold):
        pa, pb = a.split(), b.split()
        if len(pb) == 4:
            assert pa[:3] == pb[:3] and abs(float(pa[3]) - float(pb[3])) < 1e-2, (a, b)
        else:
            assert a == b, (a, b)                # "#!MLF!#", "8580.rec", "."
    out = tmp_path / "8580.rec"
    run("-c", es, "-i", work / "8580.wav", "-o", out, *flags)
    _labels_match(out, os.path.join(REF, "8580.rec"))
    if "-D" not in flags:                  # posteriors against the reference CLI's dump of the same file
        lop = tmp_path / "8580.lop"
        run("-c", es, "-i", work / "8580.wav", "-t", "post", "-o", lop, *flags)
        got, want = read_htk(str(lop)), read_htk(os.path.join(REF, "8580.lop"))
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
    assert open(out).readline().split()[0] == "000000"      # a label file prints time zero as the reference does


@pytest.mark.parametrize("flags", [(), ("-F",), ("-F", "-D")])
@pytest.mark.parametrize("which", ["es", "hu"])
def test_reference_golden_es_wav(which, flags, tmp_path):
    """/root/reference/es.wav -> es.rec (19 s, lin16 with its header taken as samples): no phoneme the ES and HU lists
    name differently occurs in it, so both model directories must reproduce it"""
    d = _es_model(tmp_path) if which == "es" else model_dir(HU)
    out = tmp_path / "es.rec"
    run("-c", d, "-i", os.path.join(REF, "es.wav"), "-o", out, *flags)
    _labels_match(out, os.path.join(REF, "es.rec"))


# ---- the multi-GPU split (BASELINE configs[3] / configs[4]) on a 1-GPU box -----------------------
def _make_list(tmp_path, name, n_files, seed, fmt="lin16", rate=8000):
    """`n_files` synthetic waveform files of 0.2-2.5 s (5 sines + noise, as SURVEY 8d prescribes) + the list"""
    rng = np.random.default_rng(seed)
    d = tmp_path / name
    d.mkdir()
    lines = []
    for i in range(n_files):
        n = int(rng.integers(rate // 5, int(rate * 2.5)))
        t = np.arange(n) / rate
        x = sum(np.sin(2 * np.pi * f * t + p) for f, p in zip(rng.uniform(200, 3400, 5), rng.uniform(0, 6.28, 5)))
        x = 0.3 * 32767 / 5 * x + rng.normal(0, 1000, n)
        p = d / ("f%03d.raw" % i)
        np.clip(x, -32768, 32767).astype("<i2").tofile(p)
        lines.append(str(p))
    lst = tmp_path / (name + ".scp")
    lst.write_text("\n".join(lines) + "\n")
    return lst


@pytest.mark.parametrize("flags", [(), ("-F",), ("-F", "-D"), ("-H",), ("-H", "-F", "-D")])
def test_two_logical_gpus_same_mlf_as_one(flags, tmp_path):
    """`-g 2` (four contexts pulling launches from one queue; both logical GPUs mapped onto this box's GPU
    with PHNREC_DEVICE_MAP) writes byte for byte the MLF of `-g 1`: utterances never interact, outputs are
    gathered in list order (srec.cpp:1246-1290, MLF format srec.cpp:1273,1156,1180)."""
    lst = _make_list(tmp_path, "hu", 60, seed=4)
    one, two = tmp_path / "one.mlf", tmp_path / "two.mlf"
    run("-c", model_dir(HU), "-l", lst, "-m", one, "-g", 1, "-b", 600, *flags)
    p = run("-c", model_dir(HU), "-l", lst, "-m", two, "-g", 2, "-b", 600, *flags,
            env={"PHNREC_DEVICE_MAP": "0,0", "PHNREC_STATS": "1"})
    assert "files=60" in p.stderr
    a, b = one.read_text(), two.read_text()
    assert a == b and a.startswith("#!MLF!#\n") and a.count("\n.\n") == 60
    # posterior dumps as well
    run("-c", model_dir(HU), "-l", lst, "-t", "post", "-g", 2, "-b", 600, *flags[:1], env={"PHNREC_DEVICE_MAP": "0,0"})
    two_lop = {f: open(f, "rb").read() for f in sorted(str(x) for x in (tmp_path / "hu").glob("*.lop"))}
    run("-c", model_dir(HU), "-l", lst, "-t", "post", "-g", 1, "-b", 600, *flags[:1])
    assert len(two_lop) == 60
    for f, blob in two_lop.items():
        assert open(f, "rb").read() == blob, f


@pytest.mark.parametrize("n_gpus,flags", [(4, ()), (8, ()), (8, ("-F",)), (4, ("-F", "-D"))])
def test_four_and_eight_logical_gpus_same_mlf_as_one(n_gpus, flags, tmp_path):
    """`-g 4` / `-g 8` (8 / 16 contexts on ONE launch queue that runs from the first line of the list to the last --
    nothing joins the contexts in between), every logical GPU mapped onto this box's GPU: byte for byte the MLF
    of `-g 1`, and the run reports every file (srec.cpp:1246-1290 is a sequential loop; the order of its output is
    the list's)."""
    lst = _make_list(tmp_path, "hu", 150, seed=5)
    one, many = tmp_path / "one.mlf", tmp_path / "many.mlf"
    run("-c", model_dir(HU), "-l", lst, "-m", one, "-g", 1, "-b", 700, *flags)
    p = run("-c", model_dir(HU), "-l", lst, "-m", many, "-g", n_gpus, "-b", 700, *flags,
            env={"PHNREC_DEVICE_MAP": ",".join(["0"] * n_gpus), "PHNREC_STATS": "1"})
    assert "files=150" in p.stderr
    a, b = one.read_text(), many.read_text()
    assert a == b and a.startswith("#!MLF!#\n") and a.count("\n.\n") == 150
    heads = [l for l in a.splitlines() if l.startswith('"')]
    assert heads == sorted(heads) and len(heads) == 150


def test_unreadable_file_in_a_gpu_list_stops_there(tmp_path):
    """the reference exit(1)s at the first file it cannot open (srec.cpp:1280-1284): with the GPU pipeline too, every
    file BEFORE it has its MLF entry, none behind it, whatever -g is"""
    lst = _make_list(tmp_path, "cz", 40, seed=6)
    names = lst.read_text().split()
    os.remove(names[25])
    for g in (1, 4):
        mlf = tmp_path / ("g%d.mlf" % g)
        e = dict(os.environ, PHNREC_DEVICE_MAP=",".join(["0"] * g))
        p = subprocess.run([BIN, "-c", model_dir(CZ), "-l", str(lst), "-m", str(mlf), "-g", str(g), "-b", "400"],
                           capture_output=True, text=True, env=e)
        assert p.returncode == 1 and "Can not open waveform file: %s" % names[25] in p.stderr
        heads = [l for l in mlf.read_text().splitlines() if l.startswith('"')]
        assert heads == ['"*/f%03d.rec"' % i for i in range(25)], (g, heads[-3:])


def test_file_that_turns_unreadable_behind_the_stat_stops_the_list_there(tmp_path):
    """-F: stage 1 only stat()s a file; it is read later, straight into a context's pinned buffer.  A file that can be
    stat()ed but not read (mode 000; skipped when the tests run as root, who may read anything) must stop the list AT
    that file like any other unreadable one: every entry before it is written -- also those of launches other contexts
    were still computing -- none behind it (srec.cpp:1246-1290)."""
    if os.geteuid() == 0:
        pytest.skip("root reads files of mode 000")
    lst = _make_list(tmp_path, "cz", 40, seed=8)
    names = lst.read_text().split()
    os.chmod(names[25], 0)
    try:
        for g, flags in ((1, ("-F",)), (4, ("-F",)), (2, ("-F", "-D"))):
            mlf = tmp_path / ("g%d%s.mlf" % (g, "d" if "-D" in flags else ""))
            e = dict(os.environ, PHNREC_DEVICE_MAP=",".join(["0"] * g))
            p = subprocess.run([BIN, "-c", model_dir(CZ), "-l", str(lst), "-m", str(mlf), "-g", str(g), "-b", "400"]
                               + list(flags), capture_output=True, text=True, env=e)
            assert p.returncode == 1 and "Can not open waveform file: %s" % names[25] in p.stderr, p.stderr
            heads = [l for l in mlf.read_text().splitlines() if l.startswith('"')]
            assert heads == ['"*/f%03d.rec"' % i for i in range(25)], (g, flags, heads[-3:])
    finally:
        os.chmod(names[25], 0o644)


def test_configs3_list_at_scale(tmp_path):
    """BASELINE configs[3] at its stated size on the one GPU of this box: the shipped HU weights, 10 000 files of
    3-15 s (slices of one synthetic 8 kHz signal, ~9 M frames), host Viterbi, one MLF -- once with `-g 1`, once as
    `-g 2` (two logical GPUs on the physical one, PHNREC_DEVICE_MAP): the two MLFs are byte-identical, every file
    has its entry in list order, and utterances decoded alone give the lines the list run wrote for them"""
    rng = np.random.default_rng(1236)
    n_base = 16 * 8000
    t = np.arange(n_base) / 8000.0
    base = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, n_base)
    base = np.clip(base, -32768, 32767).astype("<i2")
    data = tmp_path / "d"
    data.mkdir()
    names, frames = [], 0
    for i in range(10000):
        n = int(rng.uniform(3.0, 15.0) * 8000)
        o = int(rng.integers(0, n_base - n))
        f = data / ("u%05d.raw" % i)
        base[o:o + n].tofile(f)
        names.append(str(f))
        frames += (n - 200) // 80 + 1
    lst = tmp_path / "list.scp"
    lst.write_text("".join(n + "\n" for n in names))
    mlf1, mlf2 = tmp_path / "g1.mlf", tmp_path / "g2.mlf"
    p1 = run("-c", model_dir(HU), "-l", lst, "-m", mlf1, "-g", 1, env={"PHNREC_STATS": "1"})
    p2 = run("-c", model_dir(HU), "-l", lst, "-m", mlf2, "-g", 2, env={"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": "0,0"})
    assert "files=10000 frames=%d" % frames in p1.stderr and "files=10000 frames=%d" % frames in p2.stderr
    a, b = mlf1.read_bytes(), mlf2.read_bytes()
    assert a == b, "-g 2 must write the bytes -g 1 writes"
    # `-g 8` as eight logical GPUs on this one device, what it picks by itself (-F -D): contexts come up beside the running
    # list, and only those the list lives to see and the device can use -- three per PHYSICAL device at most; with
    # PHNREC_ALL_CONTEXTS=1 every planned one (16 here: one process, 16 worker threads, one launch queue).  Same bytes.
    e8 = {"PHNREC_STATS": "1", "PHNREC_DEVICE_MAP": ",".join(["0"] * 8)}
    for extra, lo, hi in (({}, 1, 3), ({"PHNREC_ALL_CONTEXTS": "1"}, 16, 24)):
        mlf8 = tmp_path / "g8.mlf"
        p8 = run("-c", model_dir(HU), "-l", lst, "-m", mlf8, "-g", 8, env=dict(e8, AUTO="1", **extra))
        st = [l for l in p8.stderr.splitlines() if l.startswith("phnrec: files=")][-1]
        assert "mode=F+D,auto" in st and "files=10000 frames=%d" % frames in st
        n_ctx = int(st.split("contexts=")[1].split()[0])
        assert lo <= n_ctx <= hi, st
        assert mlf8.read_bytes() == a, "-g 8 %s must write the bytes -g 1 writes" % extra
    lines = a.decode().splitlines()
    heads = [l for l in lines if l.startswith('"')]
    assert lines[0] == "#!MLF!#" and len(heads) == 10000 and lines.count(".") == 10000
    assert heads[0] == '"*/u00000.rec"' and heads[-1] == '"*/u09999.rec"'
    assert heads == sorted(heads), "entries in list order"
    for i in (0, 4711, 9999):                             # an utterance alone == its entry in the list run
        rec = tmp_path / "one.rec"
        run("-c", model_dir(HU), "-i", names[i], "-o", rec)
        one = [l.split() for l in open(rec) if len(l.split()) == 4]
        k = lines.index('"*/u%05d.rec"' % i)
        entry = []
        for l in lines[k + 1:]:
            if l == ".":
                break
            entry.append(l.split())
        assert [(int(x[0]), int(x[1]), x[2]) for x in entry] == [(int(x[0]), int(x[1]), x[2]) for x in one], i
        assert max(abs(float(x[3]) - float(y[3])) for x, y in zip(entry, one)) < 1e-3


def test_bench_line_carries_every_leg(tmp_path):
    """bench.py in the driver's form (shortened): ONE JSON line on stdout, at most 6 KB (the driver parses a bounded
    line: round 5's 21.7 KB one was cut, BENCH_r05.parsed = null), with the contract's keys, the roofline object with
    its cold figure / windows / traffic, the CPU baseline, and NUMBERS ONLY for the side legs; the full record --
    every leg's break-down -- in the file the line names, and leg by leg on stderr"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PHNREC_DEVICE_MAP")}
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--preheat", "10", "--cpu-seconds", "1", "--list-files", "120", "--detail-out", str(detail)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    assert len(lines[0]) + 1 <= 6144, "the stdout line is %d bytes" % (len(lines[0]) + 1)
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "ranks", "roofline", "cpu_baseline", "preheat_launches"):
        assert k in c, k
    assert "dropped" not in c, "every side leg fits: %s" % c.get("dropped")
    assert not any(k == "what" for k in _all_keys(c)), "no prose on stdout"
    assert c["roofline"]["bound"] == "mfma" and 0.3 < c["roofline"]["frac"] < 1.0 and "cold" in c["roofline"]
    assert "windows" in c["roofline"] and c["roofline"]["traffic"] > 0
    assert c["cpu_baseline"]["kind"] in ("reference", "port") and c["cpu_baseline"]["cores"] == 1
    assert c["cpu_baseline"]["parity_max_abs_vs_gpu"] < 1e-4 and "cpu_model" in c["cpu_baseline"]["host"]
    for leg in ("sharded_list", "four_systems", "systems", "small_launches", "single_file", "split_f16", "push_bunch5"):
        assert leg in c, leg
    assert c["sharded_list"]["F"]["value"] > 50000 and c["sharded_list"]["F"]["process_frames_per_s"] > 0
    assert c["sharded_list"]["mlf_all_modes_equal"] is True, c["sharded_list"]
    assert c["four_systems"].get("mlf_equal") is True, (c["four_systems"], p.stderr[-1500:])
    assert "bench detail [roofline]" in p.stderr and "bench detail [sharded_list]" in p.stderr
    d = json.loads(detail.read_text())                    # the full record
    assert d["value"] == c["value"] and d["ms_per_step"] == c["ms_per_step"] and d["roofline"]["frac"] == c["roofline"]["frac"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "preheat_launches"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["preheat_launches"] == 10
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.3 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["cold"]["launches"] == "1-20" and r["traffic_source"].startswith("profiles/hbm_traffic.json")
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] == 1
    assert c["parity_max_abs_vs_gpu"] < 1e-4
    assert c["parity_max_abs_vs_gpu_split_f16"] < 1e-4
    sf = d["split_f16"]              # the opt-in arithmetic beside the headline, never as it
    assert sf["value"] > d["value"] and sf["max_abs_vs_f32_kernels"] < 2e-5 and sf["rows_sum_to_one"] is True
    assert d["push_bunch5"]["value"] > 50000, "the 50 k frames/s floor at the shipped bunch of 5"
    assert d["push_bunch512"]["value"] > d["push_bunch5"]["value"]
    assert set(d["small_launches"]) >= {"cz_2048", "cz_4096", "en_4096"}
    assert d["wave_path"]["frames"] == 8192 and d["wave_path"]["rows_sum_to_one"] is True
    # every shipped system at the headline's launch size, in the driver-run record
    assert set(d["systems"]) >= {"hu_8192", "ru_8192", "en_8192"}
    assert all(0.2 < d["systems"][k]["frac"] < 1.0 for k in ("hu_8192", "ru_8192", "en_8192"))
    if "dropin_reference_cli" in d:
        assert d["dropin_reference_cli"]["value"] > 50000
    # configs[1]'s own input (EN, 16 kHz lin16, seed 1234, 4096 frames) through the waveform entry
    assert d["wave_path_en"]["frames"] == 4096 and d["wave_path_en"]["rows_sum_to_one"] is True
    # the reference's smoke test as a process, beside the reference's CPU build on the same file
    sf1 = d["single_file"]
    assert sf1["str"]["process_wall_s"] > 0 and sf1["post"]["process_wall_s"] > 0 and "create_trace_ms" in sf1["str"]
    if "reference_cpu_mkl" in sf1:
        assert sf1["reference_cpu_mkl"]["process_wall_s"] > 0
    # the sharded list (configs[3]'s system and list recipe), with what the host alone can do beside it
    sl = d["sharded_list"]
    assert sl["files"] == 120 and sl["gpus"] == 1 and sl["frames_per_s"] > 50000 and sl["mlf_F_equals_F_D"] is True
    assert sl["gpu_energies_E"]["value"] > 50000 and sl["mlf_E_equals_host_frontend"] is True
    assert sl["host_ceiling"]["frames_per_s"] > 0 and sl["host_ceiling"]["gpu_frontend_F"]["host_cpu_s"] > 0
    assert sl["cz_same_list"]["host_frontend"]["value"] > 50000 and sl["cz_same_list"]["gpu_frontend_F"]["value"] > 50000
    assert sl["host_ceiling"]["per_file_serial"]["files_per_s"] > 20000      # (round 3's pipeline: 80 k on these hosts; now ~400 k)
    assert sl["gpu_energies_decoder_E_D"]["value"] > 50000 and sl["mlf_E_D_equals_host_frontend"] is True
    # the fixed-work-per-GPU form (every file listed 8 times), every mode, and what `-g 8` picks by itself on this one device
    wl = sl["weak_list"]
    assert wl["files"] == 8 * 120 and all(wl[k]["value"] > 50000 and wl[k]["ceiling_over_8_gpus"] > 0 for k in (
        "host_frontend", "gpu_energies_E", "gpu_energies_decoder_E_D", "gpu_frontend_F", "gpu_frontend_decoder_F_D"))
    assert wl["as_g8_default"]["mode"] == "F+D,auto" and wl["as_g8_default"]["value"] > 50000
    assert wl["as_g8_default"]["contexts"] <= 3 < wl["as_g8_all_contexts"]["contexts"] and wl["as_g8_all_contexts"]["value"] > 50000
    # configs[4]: the four systems at once (here: all on this box's one GPU), MLFs those of each system run alone
    fs = d["four_systems"]
    assert fs["oversubscribed"] is True and set(fs["default_flags"]["per_system"]) == set(fs["systems"])
    assert fs["default_flags"]["value"] > 50000 and fs["gpu_frontend_decoder_F_D"]["value"] > 50000
    assert all(fs["mlf_equals_single_system_run"].values()) and len(fs["mlf_equals_single_system_run"]) == 4


def _all_keys(o):
    if isinstance(o, dict):
        for k, v in o.items():
            yield k
            yield from _all_keys(v)
    elif isinstance(o, list):
        for v in o:
            yield from _all_keys(v)


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 started plainly (no launcher): the parent starts two ranks itself before touching the GPU;
    PHNREC_DEVICE_MAP=0,0 puts both on the one GPU of this box (functional run: rendezvous over gloo, labelled
    oversubscribed); rank 0 prints one line whose rank count is what the process group saw"""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["PHNREC_DEVICE_MAP"] = "0,0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--preheat", "0", "--no-cpu", "--no-extras", "--list-files", "150",
                        "--detail-out", str(tmp_path / "detail.json")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    short = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    assert len(short) + 1 <= 6144 and json.loads(short)["ranks"]["world"] == 2
    line = json.loads((tmp_path / "detail.json").read_text())
    assert line["ranks"]["world"] == 2 and line["ranks"]["oversubscribed"] is True
    assert line["ranks"]["device_map"] == [0, 0] and line["n_gpus"] == 1
    assert line["value"] > 0 and line["steps"] == 5
    # the N-rank line carries the sharded-list figure: `phnrec -g 2` over the ranks' GPUs, and the host's ceiling
    sl = line["sharded_list"]
    assert sl["gpus"] == 2 and sl["device_map"] == [0, 0] and sl["files"] == 150
    # ... and the fixed-work-per-GPU form of the list (every file listed N times)
    wl = sl["weak_list"]
    assert wl["files"] == 16 * 150 and wl["frames"] == 16 * sl["frames"]
    assert wl["gpu_frontend_F"]["value"] > 50000 and wl["gpu_frontend_decoder_F_D"]["value"] > 50000
    assert wl["host_frontend"]["mode"] == "F,auto" and wl["gpu_energies_decoder_E_D"]["value"] > 50000
    fs = line["four_systems"]
    assert fs["gpu_pairs"] == ["0,0"] * 4 and fs["oversubscribed"] is True and fs["default_flags"]["value"] > 50000
    assert sl["frames_per_s"] > 50000 and sl["host_ceiling"]["frames_per_s"] > 0 and sl["cores_usable"] >= 1


def test_bench_one_rank_over_rccl(tmp_path):
    """The driver's launcher form on this box's one GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1
    bench.py --gpus 1 ...` (launcher first; nothing touches the GPU before it).  Under a launcher Ranks.init takes the
    process-group branch even at world size 1, so this executes on real RCCL what the 8-GPU scaling run adds to an
    N = 1 run: init_process_group("nccl"), the gloo side group, dist.barrier() on the device and the device-side
    all_reduce(MAX) of timed_steps (phnrec_amd/distrun.py)."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PHNREC_DEVICE_MAP")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    detail = tmp_path / "detail.json"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--kernel-only", "--steps", "5", "--warmup", "2",
                        "--preheat", "10", "--detail-out", str(detail)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) + 1 <= 6144, p.stdout[:500]
    c = json.loads(lines[0])
    assert c["ranks"]["backend"] == "nccl" and c["ranks"]["world"] == 1
    assert c["ranks"]["launcher"] == "torch.distributed.run" and c["ranks"]["oversubscribed"] is False
    assert c["n_gpus"] == 1 and c["steps"] == 5 and c["warmup"] == 2
    # 8192 CZ frames per step: 0.19-0.25 ms of kernel; the bracket (barrier + synchronise on both sides) adds the collective
    assert 0.15 < c["ms_per_step_before_closing_barrier"] <= c["ms_per_step"] < 2.0, c
    assert 0.3 < c["roofline"]["frac"] < 1.0 and c["roofline"]["kernel_ms"] <= c["ms_per_step"]
    assert "device_ids" not in p.stderr or "using GPU" not in p.stderr, "the barrier names its device"


def test_verbose_says_which_contexts_came_up(tmp_path):
    """`phnrec -v` on a list: the planned road (front-end, decoder, GPUs x contexts, ordered or shared) before the list, and
    how many of the planned contexts came up behind it -- a short list over two logical GPUs on one device plans six and
    gets by with fewer"""
    lst = _make_list(tmp_path, "hu", 40, seed=3)
    p = run("-v", "-c", model_dir(HU), "-l", lst, "-m", tmp_path / "o.mlf", "-g", 2, env={"PHNREC_DEVICE_MAP": "0,0", "AUTO": "1"})
    lines = [l for l in p.stdout.splitlines() if l.startswith("Device path:")]
    assert len(lines) == 2, p.stdout[-600:]
    assert "front-end GPU (-F, chosen by itself" in lines[0] and "2 GPU(s) x 3 context(s) planned" in lines[0]
    n = int(lines[1].split("Device path: ")[1].split(" of ")[0])
    assert 1 <= n <= 3 and "of 6 planned contexts came up" in lines[1]
    p = run("-v", "-c", model_dir(HU), "-l", lst, "-m", tmp_path / "o2.mlf", "-g", 2,
            env={"PHNREC_DEVICE_MAP": "0,0", "AUTO": "1", "PHNREC_ALL_CONTEXTS": "1"})
    assert "Device path: 6 of 6 planned contexts came up" in p.stdout


def test_more_gpus_than_the_box_has_fails_loudly(tmp_path):
    lst = _make_list(tmp_path, "cz", 3, seed=1)
    e = dict(os.environ)
    e.pop("PHNREC_DEVICE_MAP", None)
    p = subprocess.run([BIN, "-c", model_dir(CZ), "-l", str(lst), "-m", str(tmp_path / "x.mlf"), "-g", "64"],
                       capture_output=True, text=True, env=e)
    assert p.returncode != 0 and "device_id out of range" in p.stderr


def test_four_systems_at_once(tmp_path):
    """BASELINE configs[4]: tools/run_four_systems.sh -- four `phnrec -g 2` processes, one per system (CZ, HU,
    RU at 8 kHz, EN at 16 kHz), here with every GPU pair mapped onto this box's one GPU; each system's MLF equals
    the one a plain single-GPU run of that system writes."""
    script = os.path.join(ROOT, "tools", "run_four_systems.sh")
    args, lists = [], {}
    for k, system in enumerate((CZ, HU, RU, EN)):
        lst = _make_list(tmp_path, system[4:6].lower(), 12, seed=10 + k, rate=16000 if system == EN else 8000)
        with open(lst, "a") as f:                          # + the reference's bundled utterance as a 13th entry
            f.write(os.path.join(GOLD, "test.raw") + "\n")
        lists[system] = lst
        args += [model_dir(system), str(lst)]
    e = dict(os.environ, PHNREC_GPU_PAIRS="0,0 0,0 0,0 0,0")
    p = subprocess.run(["bash", script] + args + ["-b", "500"], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr
    assert p.stderr.count("phnrec: files=13") == 4
    for system, lst in lists.items():
        mlf = str(lst)[:-4] + ".mlf"
        four = open(mlf).read()
        ref = tmp_path / (system + ".ref.mlf")
        run("-c", model_dir(system), "-l", lst, "-m", ref)
        assert four == ref.read_text(), system          # (parity of each system on its own: the tests above)
        assert four.count("\n.\n") == 13
        # ... and against the reference itself: the bundled utterance's entry vs the label file SHIPPED with the reference
        lines = four.splitlines()
        k = lines.index('"*/test.rec"')
        entry = [l.split() for l in lines[k + 1:lines.index(".", k)]]
        gold = [l.split() for l in open(os.path.join(GOLD, "rec", system + ".rec")) if len(l.split()) == 4]
        assert [(int(x[0]), int(x[1]), x[2]) for x in entry] == [(int(x[0]), int(x[1]), x[2]) for x in gold], system
        assert max(abs(float(x[3]) - float(y[3])) for x, y in zip(entry, gold)) < 1e-2


def _cli_peak_rss_mb(args):
    """one CLI run; its own peak resident set (PHNREC_STATS: getrusage of the process, pinned buffers included)"""
    p = run(*args, env={"PHNREC_STATS": "1"})
    line = [ln for ln in p.stderr.splitlines() if ln.startswith("phnrec: files=")][-1]
    return float(line.split("max_rss_mb=")[1].split()[0])


def test_long_file_runs_as_row_ranges_with_bounded_pinned_memory(tmp_path):
    """SURVEY 5 'long-context': a file longer than -b frames is computed as consecutive row ranges with 15-frame halos
    (lcrc_posteriors_rows; srec.cpp:1035-1059's prime / main / flush is the halo rule) and decoded range by range.  A
    1.2 M-frame file (3.3 hours of 8 kHz audio, as a parameter file: 72 MB) with -b 32768 gives the MLF entry of the
    same file run as ONE launch (-b 2000000) byte for byte, and without the 660 MB of pinned posteriors of that launch:
    peak resident set stated and compared."""
    from tests.util import write_htk
    from phnrec_amd import modelgen
    n = 1200000
    base = modelgen.synth_mel(40000, 15, seed=21, mean_norm=True)
    mel = np.tile(base, (n // len(base), 1))
    src = tmp_path / "long.mel"
    write_htk(str(src), mel)
    lst = tmp_path / "l.txt"
    lst.write_text("%s\n" % src)
    a, b = tmp_path / "chunked.mlf", tmp_path / "one.mlf"
    rss_chunked = _cli_peak_rss_mb(["-c", model_dir(CZ), "-s", "par", "-l", lst, "-m", a, "-b", 32768])
    rss_one = _cli_peak_rss_mb(["-c", model_dir(CZ), "-s", "par", "-l", lst, "-m", b, "-b", 2000000])
    ta, tb = a.read_text(), b.read_text()
    assert ta == tb and ta.count("\n") > 1000
    # the floor of any run of this CLI on this box (HIP runtime, code objects, one context -- a one-line list gets no more,
    # whatever the file's length): a 1000-frame file, same flags
    small = tmp_path / "small.mel"
    write_htk(str(small), base[:1000])
    lst2 = tmp_path / "l2.txt"
    lst2.write_text("%s\n" % small)
    rss_floor = _cli_peak_rss_mb(["-c", model_dir(CZ), "-s", "par", "-l", lst2, "-m", tmp_path / "small.mlf", "-b", 32768])
    print("peak RSS: %.0f MB as row ranges of 32768 frames, %.0f MB as one launch, %.0f MB for a 1000-frame file"
          % (rss_chunked, rss_one, rss_floor))
    # above that floor: the features of the file (72 MB, up to three times while the parameter file is unpacked and
    # copied) + one launch's buffers (32768 x (60 + 552) B pinned, 18 MB of pageable posteriors): bound 450 MB, of which
    # only the features grow with the file
    assert rss_chunked < rss_floor + 450.0
    assert rss_one > rss_chunked + 400.0                 # 1.2 M x 552 B of posteriors, pinned


@pytest.mark.parametrize("system,fmt", [(CZ, "lin16"), (EN, "lin16")])
def test_long_waveform_row_ranges_every_mode_and_the_dump(system, fmt, tmp_path):
    """the same through the front-end, on a 100 000-frame waveform with -b 8192: labels and the posterior dump equal the
    one-launch form byte for byte; with -F / -E the long file's features come from the host front-end (the -E road's
    bits = the default mode's), so those modes write the default mode's bytes; -D decodes such a file on the host"""
    rate = 16000 if system == EN else 8000
    rng = np.random.default_rng(9)
    n = 99999 * (rate // 100) + rate // 40
    t = np.arange(16 * rate) / float(rate)
    sig = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, len(t))
    raw = np.tile(np.clip(sig, -32768, 32767).astype("<i2"), n // len(t) + 1)[:n]
    src = tmp_path / "long.raw"
    raw.tofile(str(src))
    one = tmp_path / "one.rec"
    run("-c", model_dir(system), "-i", src, "-o", one, "-b", 200000)
    for k, flags in enumerate([(), ("-E",), ("-F",), ("-F", "-D"), ("-g", "2")]):
        rec = tmp_path / ("c%d.rec" % k)
        run("-c", model_dir(system), "-i", src, "-o", rec, "-b", 8192, *flags,
            env={"PHNREC_DEVICE_MAP": "0,0"} if "-g" in flags else None)
        assert rec.read_text() == one.read_text(), flags
    lop1, lop2 = tmp_path / "one.lop", tmp_path / "c.lop"
    run("-c", model_dir(system), "-i", src, "-t", "post", "-o", lop1, "-b", 200000)
    run("-c", model_dir(system), "-i", src, "-t", "post", "-o", lop2, "-b", 8192)
    assert lop1.read_bytes() == lop2.read_bytes()
    assert read_htk(str(lop2)).shape[0] == 100000
