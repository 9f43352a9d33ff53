"""A device failure in the middle of a list (the reference's rule: the loop stops at the first failing line,
srec.cpp:1279-1290; everything in front of it has been written).

The list pipeline runs 3 x N contexts, each with its own worker thread, which now also CREATE their contexts beside the
running list.  Whatever fails on the device side -- a staging allocation (a context's reserve, a launch's buffers), a
posterior launch, the creation of a context on a device that does not exist -- must end the run at once: exit code != 0
within the time-out (no worker left waiting for a launch that never comes), the error named on stderr, and an MLF that
holds COMPLETE entries, in list order, up to some file, and nothing behind it: a byte prefix of the clean run's MLF that
ends at an entry's end.  The failures come from the library's test hooks (lcrc_debug_fail_alloc / lcrc_debug_fail_launch),
armed by the CLI from the environment (LCRC_FAULT_INJECTION=1 PHNREC_FAIL_ALLOC_NTH=k / PHNREC_FAIL_LAUNCH_NTH=k).
"""
import os
import subprocess

import numpy as np
import pytest

from tests.util import ROOT, model_dir

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
HU = "PHN_HU_SPDAT_LCRC_N1500"
N_FILES = 240


@pytest.fixture(scope="module")
def hu_list(tmp_path_factory):
    """240 synthetic 8 kHz files of 0.2-2.5 s and their list (12 KB: a "long" list, whose contexts reserve their buffers)"""
    d = tmp_path_factory.mktemp("faults")
    rng = np.random.default_rng(77)
    rate = 8000
    lines = []
    for i in range(N_FILES):
        n = int(rng.integers(rate // 5, int(rate * 2.5)))
        t = np.arange(n) / rate
        x = sum(np.sin(2 * np.pi * f * t + p) for f, p in zip(rng.uniform(200, 3400, 5), rng.uniform(0, 6.28, 5)))
        x = 0.3 * 32767 / 5 * x + rng.normal(0, 1000, n)
        p = d / ("f%03d.raw" % i)
        np.clip(x, -32768, 32767).astype("<i2").tofile(p)
        lines.append(str(p))
    lst = d / "hu.scp"
    lst.write_text("\n".join(lines) + "\n")
    assert lst.stat().st_size >= 4096
    return d, lst


def _cli(lst, mlf, g, dmap, flags, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PHNREC_") and k != "LCRC_FAULT_INJECTION"}
    env.update(PHNREC_DEVICE_MAP=dmap, **env_extra)
    if g == 1 and "-F" not in flags:
        env["PHNREC_NO_AUTO_E"] = "1"
    # -b 600: ~4 files per launch, some 60 launches -- a launch count the faults below can land in the middle of
    return subprocess.run([BIN, "-c", model_dir(HU), "-l", str(lst), "-m", str(mlf), "-g", str(g), "-b", "600"] + list(flags),
                          capture_output=True, text=True, env=env, timeout=timeout)


def _check_prefix(clean, got, what):
    """`got` = the header + whole entries of `clean`, in order, from the first on"""
    assert clean.startswith(got), "%s: the MLF is not a prefix of the clean run's" % what
    assert got.startswith("#!MLF!#\n"), what
    body = got[len("#!MLF!#\n"):]
    assert body == "" or body.endswith("\n.\n"), "%s: the MLF ends inside an entry" % what
    heads = [l for l in body.splitlines() if l.startswith('"')]
    assert heads == ['"*/f%03d.rec"' % i for i in range(len(heads))], "%s: entries out of list order" % what
    assert body.count("\n.\n") == len(heads), what
    return len(heads)


MODES = [(), ("-F",), ("-F", "-D")]
ARRANGEMENTS = [(1, "0"), (4, "0,0,0,0")]


@pytest.mark.parametrize("g,dmap", ARRANGEMENTS)
@pytest.mark.parametrize("flags", MODES)
def test_a_failing_launch_ends_the_list_there(g, dmap, flags, hu_list, tmp_path):
    """the n-th posterior launch of the process fails (first, early, mid-list, late): exit != 0 at once, the injected
    error on stderr, the MLF a whole-entry prefix of the clean run's"""
    d, lst = hu_list
    clean_mlf = tmp_path / "clean.mlf"
    p = _cli(lst, clean_mlf, g, dmap, flags, {})
    assert p.returncode == 0, p.stderr
    clean = clean_mlf.read_text()
    assert clean.count("\n.\n") == N_FILES
    written = []
    for nth in (0, 3, 20, 45):
        mlf = tmp_path / ("launch%d.mlf" % nth)
        p = _cli(lst, mlf, g, dmap, flags, {"LCRC_FAULT_INJECTION": "1", "PHNREC_FAIL_LAUNCH_NTH": str(nth)})
        what = "-g %d %s launch %d" % (g, " ".join(flags), nth)
        assert p.returncode not in (0, -9, -15), "%s: rc %d\n%s" % (what, p.returncode, p.stderr[-600:])
        assert "injected launch failure" in p.stderr, "%s: %s" % (what, p.stderr[-600:])
        n = _check_prefix(clean, mlf.read_text(), what)
        assert n < N_FILES, what
        written.append(n)
    # (launches hold ~4 files: a failure at launch 45 leaves more of the list written than one at launch 0 or 3)
    assert written[0] <= 4 * 3 * g and written[-1] >= written[0], written
    if g == 1:
        assert written[-1] >= 100, written


@pytest.mark.parametrize("g,dmap", ARRANGEMENTS)
@pytest.mark.parametrize("flags", MODES)
def test_a_failing_allocation_ends_the_list_there(g, dmap, flags, hu_list, tmp_path):
    """the n-th staging-buffer allocation of the process fails -- the first context's reserve, a clone's reserve, a
    launch's offsets --: either the run ends with the out-of-memory error named and a whole-entry prefix on disk, or (an n
    beyond the allocations this run makes) it completes with the clean run's bytes.  Never a hang, never a torn entry."""
    d, lst = hu_list
    clean_mlf = tmp_path / "clean.mlf"
    p = _cli(lst, clean_mlf, g, dmap, flags, {})
    assert p.returncode == 0, p.stderr
    clean = clean_mlf.read_text()
    failed = 0
    for nth in (0, 2, 5, 9, 14, 23):
        mlf = tmp_path / ("alloc%d.mlf" % nth)
        p = _cli(lst, mlf, g, dmap, flags, {"LCRC_FAULT_INJECTION": "1", "PHNREC_FAIL_ALLOC_NTH": str(nth)})
        what = "-g %d %s alloc %d" % (g, " ".join(flags), nth)
        assert p.returncode not in (-9, -15), "%s: killed (%d)\n%s" % (what, p.returncode, p.stderr[-600:])
        if p.returncode == 0:
            assert mlf.read_text() == clean, what
            continue
        failed += 1
        assert "allocate" in p.stderr.lower() or "memory" in p.stderr.lower(), "%s: %s" % (what, p.stderr[-600:])
        _check_prefix(clean, mlf.read_text(), what)
    assert failed >= 2, "the hook never fired"


def test_a_context_on_a_missing_device_ends_the_list(hu_list, tmp_path):
    """PHNREC_DEVICE_MAP=0,99 -g 2: the contexts of the second logical GPU cannot be created (its worker threads find out
    beside the running list): the run ends with the error named; what the first GPU's contexts had finished by then is
    on disk as whole entries in list order"""
    d, lst = hu_list
    clean_mlf = tmp_path / "clean.mlf"
    assert _cli(lst, clean_mlf, 1, "0", ("-F",), {}).returncode == 0
    for flags in MODES:
        mlf = tmp_path / "missing.mlf"
        p = _cli(lst, mlf, 2, "0,99", flags, {})
        assert p.returncode not in (0, -9, -15), p.stderr[-600:]
        assert "device_id out of range" in p.stderr
        _check_prefix(clean_mlf.read_text(), mlf.read_text(), "0,99 " + " ".join(flags))


def test_fault_variables_without_the_arming_switch_are_refused(hu_list, tmp_path):
    d, lst = hu_list
    p = _cli(lst, tmp_path / "x.mlf", 1, "0", (), {"PHNREC_FAIL_LAUNCH_NTH": "3"})
    assert p.returncode != 0 and "LCRC_FAULT_INJECTION" in p.stderr


def test_list_pipeline_under_thread_sanitizer_on_the_gpu(hu_list, tmp_path):
    """The GPU list modes' host code with -fsanitize=thread (`make tsan`), ON the GPU: workers that build their contexts
    beside the running list, clones that wait for their base, contexts that are left out, the fault path.  The HIP / HSA
    runtimes are not instrumented: the hand-offs between their own threads show up as reports whose two sides both lie
    inside libhsa-runtime64 / libamdhip64 -- suppressed by library name; what is left must be nothing.  Runs with the
    address-space randomisation off (`setarch -R`: the sanitizer's shadow layout does not survive this kernel's mmap
    entropy otherwise); skipped where that cannot be had."""
    import shutil
    csrc = os.path.join(ROOT, "phnrec_amd", "csrc")
    if subprocess.run(["make", "-s", "-C", csrc, "tsan"], capture_output=True, text=True).returncode != 0 or not shutil.which("setarch"):
        pytest.skip("no thread-sanitizer build / no setarch here")
    tsan = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec_tsan")
    supp = tmp_path / "tsan.supp"
    supp.write_text("race:libhsa-runtime64\nrace:libamdhip64\ncalled_from_lib:libhsa-runtime64\ncalled_from_lib:libamdhip64\n")
    d, lst = hu_list
    clean = tmp_path / "clean.mlf"
    assert _cli(lst, clean, 1, "0", ("-F",), {}).returncode == 0

    def run_tsan(mlf, g, dmap, flags, extra=None):
        env = {k: v for k, v in os.environ.items() if not k.startswith("PHNREC_") and k != "LCRC_FAULT_INJECTION"}
        env.update(PHNREC_DEVICE_MAP=dmap, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_thread_leaks=0 suppressions=%s" % supp,
                   **(extra or {}))
        if g == 1 and "-F" not in flags:
            env["PHNREC_NO_AUTO_E"] = "1"
        return subprocess.run(["setarch", "x86_64", "-R", tsan, "-c", model_dir(HU), "-l", str(lst), "-m", str(mlf), "-g", str(g),
                               "-b", "600"] + list(flags), capture_output=True, text=True, env=env, timeout=300)

    probe = run_tsan(tmp_path / "p.mlf", 1, "0", ("-F",))
    if "unexpected memory mapping" in probe.stderr or probe.returncode in (-11, 139):
        pytest.skip("the sanitizer does not start on this kernel: %s" % probe.stderr[-200:])
    for g, dmap in ARRANGEMENTS:
        for flags in MODES:
            mlf = tmp_path / "t.mlf"
            p = run_tsan(mlf, g, dmap, flags)
            what = "-g %d %s" % (g, " ".join(flags))
            assert "WARNING: ThreadSanitizer" not in p.stderr, "%s\n%s" % (what, p.stderr[:4000])
            assert p.returncode == 0, "%s: rc %d %s" % (what, p.returncode, p.stderr[-500:])
            assert mlf.read_text() == clean.read_text(), what
    # the fault path: a launch that fails while other workers are mid-launch and others still build their contexts
    mlf = tmp_path / "f.mlf"
    p = run_tsan(mlf, 4, "0,0,0,0", ("-F", "-D"), {"LCRC_FAULT_INJECTION": "1", "PHNREC_FAIL_LAUNCH_NTH": "7"})
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:4000]
    assert p.returncode not in (0, 66) and "injected launch failure" in p.stderr
    _check_prefix(clean.read_text(), mlf.read_text(), "tsan, launch 7")
