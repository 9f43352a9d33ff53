"""The drop-in `phnrec` CLI, host-only legs (no GPU): front-end (-t par), decoder (-s post),
list / MLF modes, configuration and command-line errors -- against files produced by the
reference CLI itself (tests/golden, tools/make_golden.py)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from phnrec_amd import modelgen
from tests.util import GOLD, ROOT, model_dir, read_htk, read_htk_header

BIN = os.environ.get("PHNREC_BIN", os.path.join(ROOT, "phnrec_amd", "bin", "phnrec"))
CZ, EN = "PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"


def run(*args, ok=True):
    p = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True)
    if ok:
        assert p.returncode == 0, p.stderr
    return p


def test_binary_exists():
    assert os.path.exists(BIN), "build with make -C phnrec_amd/csrc"


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_front_end_dump_is_bit_identical(system, tmp_path):
    out = tmp_path / "t.mel"
    run("-c", model_dir(system), "-i", os.path.join(GOLD, "test.raw"), "-t", "par", "-o", out)
    gold = os.path.join(GOLD, system, "test.mel")
    assert open(out, "rb").read() == open(gold, "rb").read()
    n, period, size, kind = read_htk_header(str(out))
    assert (period, kind) == (100000, 6) and size == 4 * modelgen.SYSTEMS[system]["nbanks"]


def test_alaw_front_end(tmp_path):
    out = tmp_path / "a.mel"
    run("-c", model_dir(CZ), "-w", "alaw", "-i", os.path.join(GOLD, "test.raw"), "-t", "par", "-o", out)
    assert open(out, "rb").read() == open(os.path.join(GOLD, "cli", "test_alaw.mel"), "rb").read()


def _system_dir(tmp_path, system):
    """Model dir for the decoder leg: the shipped one, or synthetic nets + the real config/phonemes."""
    d = model_dir(system)
    if d:
        return d
    d = str(tmp_path / system)
    modelgen.write_system(d, system, seed=1)
    shutil.copyfile(os.path.join(GOLD, system, "config"), os.path.join(d, "config"))
    shutil.copyfile(os.path.join(GOLD, system, "phonemes"), os.path.join(d, "dicts", "phonemes"))
    return d


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_decoder_reproduces_reference_labels(system, tmp_path):
    """post -> str on the reference's own posteriors: byte-identical .rec, and the same labels
    and times as the label file SHIPPED with the reference (scores within 1e-2)."""
    out = tmp_path / "t.rec"
    run("-c", _system_dir(tmp_path, system), "-s", "post", "-i", os.path.join(GOLD, system, "test.lop"), "-o", out)
    assert open(out).read() == open(os.path.join(GOLD, system, "test.rec")).read()
    mine = [l.split() for l in open(out)]
    shipped = [l.split() for l in open(os.path.join(GOLD, "rec", system + ".rec"))]
    assert [m[:3] for m in mine] == [s[:3] for s in shipped]
    assert max(abs(float(m[3]) - float(s[3])) for m, s in zip(mine, shipped)) < 1e-2
    assert mine[0][0] == "000000"            # "%d00000" of frame 0, as the reference prints it


def _es_model(tmp_path):
    """the reference's test/PHN_ES = the HU model under another config (source/format=lin16) and phoneme list"""
    d = tmp_path / "PHN_ES"
    shutil.copytree(model_dir("PHN_HU_SPDAT_LCRC_N1500"), d)
    shutil.copy(os.path.join(GOLD, "ref", "PHN_ES", "config"), d / "config")
    shutil.copy(os.path.join(GOLD, "ref", "PHN_ES", "dicts", "phonemes"), d / "dicts" / "phonemes")
    return d


def test_reference_fixture_8580_host_legs(tmp_path):
    """/root/reference/test (8580.wav, lsit.txt, the MLF `test`): the host legs of that fixture.  The .wav's 44-byte
    RIFF header is read as 22 samples like the reference does (srec.cpp:1384-1422 loads the whole file): `-t par`
    must equal the reference CLI's dump bit for bit; decoding the reference's posterior dump through the list mode
    (`-s post -l`, one-column line with a bare name -> MLF entry "8580.rec", srec.cpp:1424-1436) must give the
    reference's MLF: names, labels, times exact, scores within 1e-2 of the SHIPPED file."""
    ref = os.path.join(GOLD, "ref")
    es = _es_model(tmp_path)
    mel = tmp_path / "8580.mel"
    run("-c", es, "-i", os.path.join(ref, "8580.wav"), "-t", "par", "-o", mel)
    assert open(mel, "rb").read() == open(os.path.join(ref, "8580.mel"), "rb").read()
    work = tmp_path / "work"
    work.mkdir()
    shutil.copy(os.path.join(ref, "8580.lop"), work / "8580.lop")
    (work / "l.txt").write_text("8580.lop\n")
    p = subprocess.run([BIN, "-c", str(es), "-s", "post", "-l", "l.txt", "-m", "out.mlf"], cwd=work,
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    mine = (work / "out.mlf").read_text().splitlines()
    gold = open(os.path.join(ref, "8580.mlf")).read().splitlines()
    assert len(mine) == len(gold)
    for a, b in zip(mine, gold):
        pa, pb = a.split(), b.split()
        if len(pb) == 4:
            assert pa[:3] == pb[:3] and abs(float(pa[3]) - float(pb[3])) < 1e-2, (a, b)
        else:
            assert a == b, (a, b)
    out = tmp_path / "8580.rec"
    run("-c", es, "-s", "post", "-i", work / "8580.lop", "-o", out)
    mine = [l.split() for l in open(out)]
    shipped = [l.split() for l in open(os.path.join(ref, "8580.rec"))]
    assert [m[:3] for m in mine] == [s[:3] for s in shipped]
    assert max(abs(float(m[3]) - float(s[3])) for m, s in zip(mine, shipped)) < 1e-2


def test_list_mode_mlf_and_derived_names(tmp_path):
    data = tmp_path / "data"
    data.mkdir()
    for n in ("utt_a", "utt_b", "utt_c"):
        shutil.copyfile(os.path.join(GOLD, "cli", n + ".lop"), data / (n + ".lop"))
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s\n" % (data / (n + ".lop")) for n in ("utt_a", "utt_b", "utt_c")))
    mlf = tmp_path / "out.mlf"
    run("-c", model_dir(CZ), "-s", "post", "-l", lst, "-m", mlf)
    assert mlf.read_text() == open(os.path.join(GOLD, "cli", "list.mlf")).read()
    # without -m: one label file next to each input, suffix labels/suffix
    run("-c", model_dir(CZ), "-s", "post", "-l", lst)
    for n in ("utt_a", "utt_b", "utt_c"):
        assert (data / (n + ".rec")).read_text() == open(os.path.join(GOLD, "cli", n + ".rec")).read()
    # two-column lines name the target explicitly
    lst2 = tmp_path / "list2.txt"
    lst2.write_text("%s \t %s\n" % (data / "utt_b.lop", tmp_path / "x.lab"))
    run("-c", model_dir(CZ), "-s", "post", "-l", lst2, "-j", 2)
    assert (tmp_path / "x.lab").read_text() == open(os.path.join(GOLD, "cli", "utt_b.rec")).read()


def test_short_file_front_end_and_param_suffix(tmp_path):
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    (tmp_path / "utt_c.raw").write_bytes(raw[:3000])
    lst = tmp_path / "l.txt"
    lst.write_text("%s\n" % (tmp_path / "utt_c.raw"))
    run("-c", model_dir(CZ), "-l", lst, "-t", "par")          # target derived with params/suffix = mel
    assert (tmp_path / "utt_c.mel").read_bytes() == open(os.path.join(GOLD, "cli", "utt_c.mel"), "rb").read()
    assert read_htk(str(tmp_path / "utt_c.mel")).shape == (17, 15)
    # a signal shorter than one frame still yields one frame (srec.cpp:731,945)
    (tmp_path / "tiny.raw").write_bytes(raw[:100])
    run("-c", model_dir(CZ), "-i", tmp_path / "tiny.raw", "-t", "par", "-o", tmp_path / "tiny.mel")
    assert read_htk(str(tmp_path / "tiny.mel")).shape == (1, 15)


def test_insertion_penalty_option(tmp_path):
    lop = os.path.join(GOLD, CZ, "test.lop")
    run("-c", model_dir(CZ), "-s", "post", "-i", lop, "-o", tmp_path / "a.rec", "-p", "-4.6875")
    run("-c", model_dir(CZ), "-s", "post", "-i", lop, "-o", tmp_path / "b.rec", "-p-20")
    assert (tmp_path / "a.rec").read_text() == open(os.path.join(GOLD, CZ, "test.rec")).read()
    assert len((tmp_path / "b.rec").read_text().splitlines()) < len((tmp_path / "a.rec").read_text().splitlines())


def test_errors_match_the_reference_texts(tmp_path):
    p = run("-i", "x", ok=False)
    assert p.returncode == 1 and "ERROR: Configuration directory is not set (-c)" in p.stderr
    p = run("-c", tmp_path / "none", "-i", "x", ok=False)
    assert p.returncode == 1 and "Can not open configuration file" in p.stderr
    d = tmp_path / "m"
    modelgen.write_model_dir(str(d), 15, 32, 12, seed=1)
    p = run("-c", d, "-s", "str", "-t", "par", "-i", "x", ok=False)
    assert "Unsupported data conversion (-s, -t)" in p.stderr
    p = run("-c", d, "-o", "y", ok=False)
    assert "The input file is not specified (-i)" in p.stderr
    p = run("-c", d, "-s", "foo", "-i", "x", ok=False)
    assert "Invalid data format 'foo'" in p.stderr
    p = run("-c", d, "-i", tmp_path / "missing.raw", "-t", "par", "-o", tmp_path / "o.mel", ok=False)
    assert "Can not open waveform file" in p.stderr
    p = run("-c", d, "-q", ok=False)
    assert "Error during command line parsing" in p.stderr
    # unknown configuration variable is fatal, with the line number (srec.cpp:242,251)
    cfg = open(os.path.join(d, "config")).read().replace("[melbanks]\n", "[melbanks]\nbogus=1\n")
    open(os.path.join(d, "config"), "w").write(cfg)
    p = run("-c", d, "-i", "x", ok=False)
    assert "Unknown variable in configuration file" in p.stderr and "line" in p.stderr
    cfg = cfg.replace("bogus=1\n", "nbanks=abc\n")
    open(os.path.join(d, "config"), "w").write(cfg)
    p = run("-c", d, "-i", "x", ok=False)
    assert "Invalid argument for a vatiable" in p.stderr
    # a damaged model is reported at start-up, before any GPU is touched
    d2 = tmp_path / "m2"
    modelgen.write_model_dir(str(d2), 15, 32, 12, seed=1)
    os.remove(os.path.join(d2, "weights", "band1.nbin"))
    p = run("-c", d2, "-i", "x", ok=False)
    assert "ERROR: Loading neural network" in p.stderr


def test_verbose_banner(tmp_path):
    p = run("-v", "-c", model_dir(CZ), "-s", "post", "-i", os.path.join(GOLD, CZ, "test.lop"), "-o", tmp_path / "v.rec")
    for needle in ("System initialization", "- mel-banks ...", "- posteriors (loading NNs) ...",
                   "------------------- SUMMARY -------------------", "Word penalty: -4.687500", "test.lop -> "):
        assert needle in p.stdout
    q = run("-c", model_dir(CZ), "-s", "post", "-i", os.path.join(GOLD, CZ, "test.lop"), "-o", tmp_path / "q.rec")
    assert q.stdout == ""


def _device_path_line(*args, env=None):
    e = dict(os.environ, **(env or {}))
    p = subprocess.run([BIN, "-v"] + [str(a) for a in args], capture_output=True, text=True, env=e)
    lines = [l for l in p.stdout.splitlines() if l.startswith("Device path: front-end")]      # (a list run adds "... contexts came up")
    assert len(lines) == 1, p.stdout[-800:] + p.stderr[-300:]
    return lines[0], p


def test_verbose_names_the_device_path(tmp_path):
    """`phnrec -v` prints ONE line that names the road a run takes through the device: front-end (host / -E / -F), decoder
    (host / device), GPUs x contexts planned, ordered or shared posterior kernels, frames per launch -- and why, when the
    run chose by itself.  The line is written when the run is planned, before any context exists: without a GPU (here) the
    run then fails with the library's "no HIP device" error, never with a CPU fallback."""
    raw = os.path.join(GOLD, "test.raw")
    short = tmp_path / "short.scp"
    short.write_text((raw + "\n") * 3)
    long_ = tmp_path / "long.scp"
    long_.write_text((raw + "\n") * 200)
    assert long_.stat().st_size >= 4096
    mlf = tmp_path / "o.mlf"
    line, p = _device_path_line("-c", model_dir(CZ), "-i", raw, "-o", tmp_path / "a.rec")
    assert "front-end host (one file)" in line and "decoder host" in line and "1 GPU(s) x 1 context(s)" in line
    assert "share the device" in line and "32768 frames per launch" in line
    if p.returncode != 0:                    # (here: no GPU.  On a GPU box the run simply succeeds)
        assert "no HIP device" in p.stderr
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", short, "-m", mlf)
    assert "front-end host (a short list on one GPU)" in line and "1 GPU(s) x 3 context(s)" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-m", mlf)
    assert "front-end GPU (-F, chosen by itself" in line and "decoder host" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-m", mlf, env={"PHNREC_LN_FORM": "0"})
    assert "front-end host (PHNREC_NO_AUTO_E" in line          # (a libm whose logf the device cannot reproduce, one GPU)
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-m", mlf, "-g", "2", env={"PHNREC_LN_FORM": "0"})
    assert "mel-bank energies" in line and "-E, chosen by itself" in line and "2 GPU(s) x 3 context(s)" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-m", mlf, "-g", "8")
    assert "front-end GPU (-F, chosen by itself" in line and "decoder GPU (-D, chosen by itself" in line
    assert "8 GPU(s) x 2 context(s)" in line and "one after the other in queueing order" in line and "65536 frames per launch" in line
    assert "decoder runs beside the context's next launch" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-m", mlf, "-F", "-D", "-b", "5000")
    assert "front-end GPU (-F);" in line and "decoder GPU (-D);" in line and "5000 frames per launch" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-l", long_, "-t", "post", "-E")
    assert "mel-bank energies (-E)" in line and "decoder none (posterior dump)" in line
    line, _ = _device_path_line("-c", model_dir(CZ), "-s", "par", "-i", os.path.join(GOLD, CZ, "test.mel"), "-o", tmp_path / "b.rec")
    assert "front-end none (parameter files in)" in line
    # conversions that never touch the GPU print no such line
    p = run("-v", "-c", model_dir(CZ), "-s", "post", "-i", os.path.join(GOLD, CZ, "test.lop"), "-o", tmp_path / "v.rec")
    assert "Device path" not in p.stdout


def test_a_missing_file_in_the_middle_of_a_list(tmp_path):
    """The reference works through a list line by line and exit(1)s at the first file it cannot open
    (srec.cpp:1280-1284, MError srec.cpp:118-122): everything BEFORE that line has been written (label files,
    MLF entries), nothing after it.  The chunked pipeline here must leave the same outputs behind."""
    data = tmp_path / "data"
    data.mkdir()
    names = ["utt_a", "utt_b", "gone", "utt_c"]
    for n in names:
        if n != "gone":
            shutil.copyfile(os.path.join(GOLD, "cli", n + ".lop"), data / (n + ".lop"))
    lst = tmp_path / "list.txt"
    lst.write_text("".join("%s\n" % (data / (n + ".lop")) for n in names))
    mlf = tmp_path / "out.mlf"
    p = run("-c", model_dir(CZ), "-s", "post", "-l", lst, "-m", mlf, ok=False)
    assert p.returncode == 1 and "ERROR: Can not open file: %s" % (data / "gone.lop") in p.stderr
    gold = open(os.path.join(GOLD, "cli", "list.mlf")).read()
    want = gold[:gold.index('"*/utt_c.rec"')]                  # header + utt_a + utt_b
    assert mlf.read_text() == want
    p = run("-c", model_dir(CZ), "-s", "post", "-l", lst, ok=False)
    assert p.returncode == 1
    assert (data / "utt_a.rec").exists() and (data / "utt_b.rec").exists() and not (data / "utt_c.rec").exists()
    # an unparsable line stops the list the same way: the lines before it are processed
    lst.write_text("%s\n%s\n \n%s\n" % (data / "utt_a.lop", data / "utt_b.lop", data / "utt_c.lop"))
    for n in ("utt_a", "utt_b"):
        os.remove(data / (n + ".rec"))
    p = run("-c", model_dir(CZ), "-s", "post", "-l", lst, ok=False)
    assert p.returncode == 1 and "Invalid line in file list" in p.stderr
    assert (data / "utt_a.rec").exists() and (data / "utt_b.rec").exists() and not (data / "utt_c.rec").exists()


@pytest.mark.parametrize("threads,batch", [(1, 100), (4, 300), (8, 100000)])
def test_long_list_leaves_in_list_order(tmp_path, threads, batch):
    """The list pipeline (read-ahead on the pool, runs of consecutive staged files, in-order writer) on a list of
    240 entries with small and large launch sizes and 1 / 4 / 8 host threads: the MLF is the concatenation of the
    reference CLI's entries in list order (srec.cpp:1246-1290 is a sequential loop), whatever the interleaving."""
    gold = open(os.path.join(GOLD, "cli", "list.mlf")).read()
    entries = {}
    for n in ("utt_a", "utt_b", "utt_c"):
        i = gold.index('"*/%s.rec"' % n)
        entries[n] = gold[i:gold.index("\n.\n", i) + 3]
    rng = np.random.default_rng(threads)
    order = [("utt_a", "utt_b", "utt_c")[int(k)] for k in rng.integers(0, 3, 240)]
    data = tmp_path / "d"
    data.mkdir()
    lines = []
    for i, n in enumerate(order):
        sub = data / ("%03d" % i)
        sub.mkdir()
        shutil.copyfile(os.path.join(GOLD, "cli", n + ".lop"), sub / (n + ".lop"))
        lines.append("%s\n" % (sub / (n + ".lop")))
    lst = tmp_path / "list.txt"
    lst.write_text("".join(lines))
    mlf = tmp_path / "out.mlf"
    run("-c", model_dir(CZ), "-s", "post", "-l", lst, "-m", mlf, "-j", threads, "-b", batch)
    assert mlf.read_text() == "#!MLF!#\n" + "".join(entries[n] for n in order)


def test_host_pipeline_under_thread_sanitizer(tmp_path):
    """`make tsan` (csrc/Makefile): the host code with -fsanitize=thread.  The CPU-only list modes -- `-t par -l` (read
    + front-end on the pool, dumps written by the host workers) and `-s post -l -m` (decode + in-order MLF writer) -- run
    through the same RunPipeline as the GPU modes with 8 pool threads: zero ThreadSanitizer reports, MLF in list order,
    every dump bit-identical to the plain build's."""
    csrc = os.path.join(ROOT, "phnrec_amd", "csrc")
    b = subprocess.run(["make", "-s", "-C", csrc, "tsan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    tsan = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec_tsan")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_thread_leaks=0")

    def run_tsan(*args):
        p = subprocess.run([tsan] + [str(a) for a in args], capture_output=True, text=True, env=env)
        assert "ThreadSanitizer" not in p.stderr and p.returncode == 0, p.stderr[-3000:]

    # -t par over 60 waveform files of varying length
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    rng = np.random.default_rng(11)
    wav = tmp_path / "wav"
    wav.mkdir()
    lines = []
    for i in range(60):
        n = int(rng.integers(400, len(raw) // 2)) * 2
        (wav / ("w%02d.raw" % i)).write_bytes(raw[:n])
        lines.append("%s %s\n" % (wav / ("w%02d.raw" % i), wav / ("w%02d.tsan.mel" % i)))
    lst = tmp_path / "par.txt"
    lst.write_text("".join(lines))
    run_tsan("-c", model_dir(CZ), "-t", "par", "-l", lst, "-j", 8, "-b", 700)
    lst2 = tmp_path / "par2.txt"
    lst2.write_text("".join(l.replace(".tsan.mel", ".mel") for l in lines))
    run("-c", model_dir(CZ), "-t", "par", "-l", lst2, "-j", 8)
    for i in range(60):
        assert (wav / ("w%02d.tsan.mel" % i)).read_bytes() == (wav / ("w%02d.mel" % i)).read_bytes()

    # -s post -m over 40 posterior dumps, in list order
    gold = open(os.path.join(GOLD, "cli", "list.mlf")).read()
    entries = {}
    for n in ("utt_a", "utt_b", "utt_c"):
        i = gold.index('"*/%s.rec"' % n)
        entries[n] = gold[i:gold.index("\n.\n", i) + 3]
    order = [("utt_a", "utt_b", "utt_c")[int(k)] for k in rng.integers(0, 3, 40)]
    lines = []
    for i, n in enumerate(order):
        sub = tmp_path / ("p%02d" % i)
        sub.mkdir()
        shutil.copyfile(os.path.join(GOLD, "cli", n + ".lop"), sub / (n + ".lop"))
        lines.append("%s\n" % (sub / (n + ".lop")))
    lst3 = tmp_path / "post.txt"
    lst3.write_text("".join(lines))
    mlf = tmp_path / "tsan.mlf"
    run_tsan("-c", model_dir(CZ), "-s", "post", "-l", lst3, "-m", mlf, "-j", 8, "-b", 900)
    assert mlf.read_text() == "#!MLF!#\n" + "".join(entries[n] for n in order)


def test_list_pipeline_many_settings_no_hang(tmp_path):
    """The pipeline (feeder with chunked stage 1, LIFO pool whose callers work along, workers, in-order writer) under 24
    combinations of pool size and launch size, each run under a timeout: every run ends, with the same MLF.  (A lost
    wake-up shows as a hang, not as a wrong answer: hence the timeout.)"""
    data = tmp_path / "d"
    data.mkdir()
    rng = np.random.default_rng(3)
    lines = []
    for i in range(120):
        n = ("utt_a", "utt_b", "utt_c")[int(rng.integers(0, 3))]
        sub = data / ("%03d" % i)
        sub.mkdir()
        shutil.copyfile(os.path.join(GOLD, "cli", n + ".lop"), sub / (n + ".lop"))
        lines.append("%s\n" % (sub / (n + ".lop")))
    lst = tmp_path / "list.txt"
    lst.write_text("".join(lines))
    first = None
    for k in range(24):
        mlf = tmp_path / ("o%d.mlf" % (k % 2))
        p = subprocess.run([BIN, "-c", model_dir(CZ), "-s", "post", "-l", str(lst), "-m", str(mlf), "-j", str(k % 8 + 1),
                            "-b", str((k % 5) * 700 + 100)], capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        text = mlf.read_text()
        first = first or text
        assert text == first


@pytest.mark.parametrize("P,S,prune", [(5, 1, 40), (17, 2, 12), (40, 3, 40), (62, 3, 40), (33, 4, 25), (64, 3, 7), (8, 3, 40),
                                       (16, 3, 40), (47, 3, 3), (61, 3, 40),
                                       # around the packed AVX-512 form's limit ((winner + 1) << 24 in a signed word: <= 127 phonemes)
                                       (127, 3, 40), (128, 3, 40), (130, 3, 12), (200, 3, 40)])
def test_host_decoder_vector_and_plain_forms_vs_the_decoder_oracle(tmp_path, P, S, prune):
    """The host Viterbi runs state-major: sixteen phonemes at a time on AVX-512 where the CPU has it (three states per
    phoneme: the whole frame in one pass, packed tokens, the entry row as scalars), eight on AVX2 (PHNREC_NO_AVX512=1),
    as plain C++ with PHNREC_NO_AVX2=1: all
    must write exactly what the decoder oracle (the restatement of phndec.cpp:96-303 that reproduces the reference's
    .rec files) gives -- on posteriors drawn from FOUR values, so that ties between tokens (first strict maximum,
    phoneme-major order) occur all the time -- for phoneme counts around the vector width, 1-4 states, short horizons"""
    from oracle import binding as ob
    from tests.util import write_htk
    rng = np.random.default_rng(P * 100 + S)
    d = tmp_path / "m"
    shutil.copytree(model_dir(CZ), d)
    cfg, section = (d / "config").read_text().splitlines(True), None
    for i, line in enumerate(cfg):
        if line.startswith("["):
            section = line.strip()
        if section == "[decoder]" and line.startswith("num_states_per_phn="):
            cfg[i] = "num_states_per_phn=%d\n" % S
        if section == "[decoder]" and line.startswith("time_pruning="):
            cfg[i] = "time_pruning=%d\n" % prune
    (d / "config").write_text("".join(cfg))
    (d / "dicts" / "phonemes").write_text("".join("q%d\n" % i for i in range(P)))
    wpen = -4.6875                                                  # PHN_CZ's decoder/wpenalty
    for trial, T in enumerate((1, 3, prune, prune + 1, 300)):
        post = rng.choice(np.array([0.5, 0.25, 0.125, 1e-3], np.float32), size=(T, P * S))
        hold = rng.integers(0, P * S, size=T // 6 + 1)
        for t in range(T):                                          # a token that stays a few frames, like speech
            post[t, hold[t // 6]] = 0.5
        lop = tmp_path / ("t%d.lop" % trial)
        write_htk(str(lop), post)
        want = ob.phndec(np.log(post), P, S, prune, wpen)
        text = "".join("%d00000 %d00000 q%d %f\n" % (a, b, p, s) for a, b, p, s in want)
        for env in ({}, {"PHNREC_NO_AVX512": "1"}, {"PHNREC_NO_AVX2": "1"}):
            rec = tmp_path / "o.rec"
            p = subprocess.run([BIN, "-c", str(d), "-s", "post", "-i", str(lop), "-o", str(rec)], capture_output=True,
                               text=True, env=dict(os.environ, **env))
            assert p.returncode == 0, p.stderr
            assert rec.read_text() == text, (T, env)


@pytest.mark.parametrize("system,options", [(CZ, {}), (EN, {}), (CZ, {"preem_coef": "0.97", "z_mean_source": "true"}),
                                           (CZ, {"nbanks_full": "19"})])
def test_front_end_eight_frames_in_lockstep_equals_the_plain_form(tmp_path, system, options):
    """The host front-end runs sixteen (AVX-512) or eight (AVX2) frames side by side -- one frame per vector lane, the
    scalar code's IEEE operations in the scalar code's order -- and the tail one frame at a time; PHNREC_NO_AVX512=1
    keeps to groups of eight, PHNREC_NO_AVX2=1 runs every frame the plain way.  The `-t par` dumps must be the same
    bytes -- also with pre-emphasis, source mean removal and a wider filter bank switched on, for lin16 and A-law, and
    for lengths around the group sizes (1 ... 33 frames, and a long one)."""
    d = tmp_path / "m"
    shutil.copytree(model_dir(system), d)
    if options:
        cfg, section = (d / "config").read_text().splitlines(True), None
        seen = set()
        for i, line in enumerate(cfg):
            if line.startswith("["):
                section = line.strip()
            for k, v in options.items():
                if section == "[melbanks]" and line.startswith(k + "="):
                    cfg[i] = "%s=%s\n" % (k, v)
                    seen.add(k)
        at = [i for i, l in enumerate(cfg) if l.strip() == "[melbanks]"][0]
        for k, v in options.items():
            if k not in seen:
                cfg.insert(at + 1, "%s=%s\n" % (k, v))
        (d / "config").write_text("".join(cfg))
    rng = np.random.default_rng(5)
    rate = modelgen.SYSTEMS[system]["sample_freq"]
    vs, step = rate // 40, rate // 100
    for frames in (1, 7, 8, 9, 15, 16, 17, 25, 33, 1203):
        n = (frames - 1) * step + vs + int(rng.integers(0, step))
        sig = (rng.normal(0, 3000, n) + 8000 * np.sin(np.arange(n) * 0.05)).clip(-32768, 32767).astype("<i2")
        raw = tmp_path / "x.raw"
        sig.tofile(raw)
        for fmt in ("lin16", "alaw"):
            outs = []
            for env in ({}, {"PHNREC_NO_AVX512": "1"}, {"PHNREC_NO_AVX2": "1"}):
                out = tmp_path / ("o%d.mel" % len(outs))
                p = subprocess.run([BIN, "-c", str(d), "-w", fmt, "-i", str(raw), "-t", "par", "-o", str(out)],
                                   capture_output=True, text=True, env=dict(os.environ, **env))
                assert p.returncode == 0, p.stderr
                outs.append(out.read_bytes())
            assert outs[0] == outs[1] == outs[2], (frames, fmt)
            assert read_htk_header(str(tmp_path / "o0.mel"))[0] == (frames if fmt == "lin16" else (2 * n - vs) // step + 1)


def test_vector_ln_is_this_hosts_logf_everywhere():
    """The host front-end's ln() (host/veclog.cpp: glibc's logf restated on AVX-512 registers, used only after a check
    against this process's libm) equals logf() on EVERY non-negative float and a stride of the negative bit patterns:
    `phnrec --selftest-ln` counts the differences.  On a host without AVX-512, or with a libm whose logf is another
    algorithm, the vector form switches itself off and the count is trivially zero -- the line says which form ran."""
    p = subprocess.run([BIN, "--selftest-ln"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 of 2^31" in p.stdout, p.stdout
    q = subprocess.run([BIN, "--selftest-ln"], capture_output=True, text=True, env=dict(os.environ, PHNREC_NO_VECTOR_LN="1"))
    assert q.returncode == 0 and "libm logf per value" in q.stdout, q.stdout
