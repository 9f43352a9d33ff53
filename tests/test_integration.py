"""The binding INTEGRATION.md describes, compiled and run (tests/integration/): class Traps forwarding to the C
ABI, (a) under a small driver that repeats SpeechRec::ProcessOffline's call sequence and (b) under the
REFERENCE's own command line, built from its sources where they lie with traps.cpp / nn.cpp left out."""
import os
import subprocess

import numpy as np
import pytest

from phnrec_amd import modelgen
from tests.util import GOLD, ROOT, model_dir, read_htk

HERE = os.path.join(ROOT, "tests", "integration")
DEMO = os.path.join(HERE, "_build", "binding_demo")
REFCLI = os.path.join(HERE, "_build", "phnrec_ref_lcrc")
CZ, EN = "PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"


def test_binding_compiles_and_links():
    """traps_lcrc.cpp against include/lcrc.h, linked with libphnrec_lcrc.so; with /root/reference present also
    the reference's CLI over it.  Without a GPU Traps::Init must fail the way the reference's does: message on
    stderr, exit(1) -- there is no CPU path to fall back to."""
    if os.path.isdir("/root/reference") or not os.path.exists(DEMO):
        subprocess.check_call(["make", "-s", "-C", HERE])
    assert os.path.exists(DEMO)
    needed = subprocess.run(["readelf", "-d", DEMO], capture_output=True, text=True).stdout
    assert "libphnrec_lcrc.so" in needed
    if os.path.isdir("/root/reference"):
        assert os.path.exists(REFCLI)
    try:
        import torch
        if torch.cuda.is_available():
            return
    except ImportError:
        pass
    for exe, args in ((DEMO, [model_dir(CZ), "15", "5", os.path.join(GOLD, "test.raw"), "/dev/null"]),
                      (REFCLI, ["-c", model_dir(CZ), "-i", os.path.join(GOLD, "test.raw"), "-o", "/dev/null"])):
        if not os.path.exists(exe):
            continue
        p = subprocess.run([exe] + args, capture_output=True, text=True)
        assert p.returncode == 1 and "no HIP device" in p.stderr, (exe, p.returncode, p.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("system,bunch", [(CZ, 5), (EN, 5), (CZ, 1000)])
def test_binding_demo_reproduces_the_reference(system, bunch, tmp_path, oracle_mod):
    """setters / Init / Reset / prime / main / flush through class Traps == the reference CLI's -t post dump"""
    spec = modelgen.SYSTEMS[system]
    mel = read_htk(os.path.join(GOLD, system, "test.mel"))
    if spec["sent_mean_norm"]:
        mel = oracle_mod.sentence_mean_norm(mel)
    mel.astype(np.float32).tofile(tmp_path / "mel.f32")
    p = subprocess.run([DEMO, model_dir(system), str(spec["nbanks"]), str(bunch), str(tmp_path / "mel.f32"),
                        str(tmp_path / "post.f32")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    want = read_htk(os.path.join(GOLD, system, "test.lop"))
    got = np.fromfile(tmp_path / "post.f32", np.float32).reshape(want.shape)
    assert np.abs(got - want).max() < 1e-4
    assert "delay %d" % min(9999, mel.shape[0] + 15 - 1) in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("split_f16", [False, True])
@pytest.mark.parametrize("system", [CZ, EN])
def test_reference_cli_over_the_library(system, split_f16, tmp_path, monkeypatch):
    """the reference's own phnrec.cpp / srec.cpp / melbanks.cpp / phndec.cpp with the MI355X library behind
    Traps: its smoke test (test.sh) gives the shipped label file, -t post its own posterior dump; also with the
    binding's PHNREC_SPLIT_F16 switch (lcrc_set_arithmetic)"""
    if not os.path.exists(REFCLI):
        pytest.skip("tests/integration/_build/phnrec_ref_lcrc is built where /root/reference exists")
    if split_f16:
        monkeypatch.setenv("PHNREC_SPLIT_F16", "1")
    raw = os.path.join(GOLD, "test.raw")
    rec = tmp_path / "t.rec"
    p = subprocess.run([REFCLI, "-c", model_dir(system), "-i", raw, "-o", str(rec)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = [l.split() for l in open(rec) if len(l.split()) == 4]
    gold = [l.split() for l in open(os.path.join(GOLD, "rec", system + ".rec")) if len(l.split()) == 4]
    assert [m[:3] for m in mine] == [g[:3] for g in gold]
    assert max(abs(float(m[3]) - float(g[3])) for m, g in zip(mine, gold)) < 1e-2
    lop = tmp_path / "t.lop"
    p = subprocess.run([REFCLI, "-c", model_dir(system), "-i", raw, "-t", "post", "-o", str(lop)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert np.abs(read_htk(str(lop)) - read_htk(os.path.join(GOLD, system, "test.lop"))).max() < 1e-4
