"""GPU mel-bank front-end ("next" row f1): raw samples -> log mel energies -> posteriors, all on the
device, against the reference CLI's own `-t par` / `-t post` dumps (tests/golden)."""
import os

import numpy as np
import pytest

from phnrec_amd import modelgen
from tests.util import GOLD, model_dir, read_htk

pytestmark = pytest.mark.gpu
CZ, EN = "PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"
HU, RU = "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500"
# the FFT, window, power and mel sums follow the reference operation by operation; ln() is computed in
# double and rounded once, glibc's logf is correctly rounded except in rare cases: <= 1 ulp (1.9e-6 at ~20)
TOL_MEL = 2e-6


@pytest.fixture(scope="module")
def capi():
    from phnrec_amd import capi
    capi.load()
    return capi


def _ctx(capi, system, **over):
    spec = modelgen.SYSTEMS[system]
    ctx = capi.Lcrc(model_dir(system), spec["nbanks"])
    cfg = dict(wave_format="lin16", sample_freq=spec["sample_freq"], vector_size=spec["vector_size"],
               vector_step=spec["vector_step"], lower_freq=float(spec["lower"]), higher_freq=float(spec["higher"]),
               sent_mean_norm=spec["sent_mean_norm"])
    cfg.update(over)
    ctx.configure_frontend(**cfg)
    return ctx


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_mel_matches_reference_dump(capi, system):
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ctx = _ctx(capi, system)
    mel, foff = ctx.wave_to_mel([raw])
    want = read_htk(os.path.join(GOLD, system, "test.mel"))
    assert mel.shape == want.shape and list(foff) == [0, want.shape[0]]
    assert np.abs(mel - want).max() <= TOL_MEL
    assert (mel == want).mean() > 0.999, "identical up to the rare last-bit difference of logf"


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_energies_plus_libm_logf_equal_the_reference_dump_bit_for_bit(capi, system):
    """lcrc_wave_stage_energies: the GPU front-end stopped in front of ln().  Everything up to there is IEEE arithmetic in
    the reference's order, so ln() taken by THIS host's libm (the reference's `x > 0 ? logf(x) : 0`, dspc.h:155-160) must
    give the reference CLI's `-t par` dump bit for bit -- where the all-GPU features are allowed one ulp."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype, libm.logf.argtypes = ctypes.c_float, [ctypes.c_float]
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ctx = _ctx(capi, system)
    en, foff = ctx.wave_energies_staged([raw])
    want = read_htk(os.path.join(GOLD, system, "test.mel"))
    assert en.shape == want.shape and list(foff) == [0, want.shape[0]]
    got = np.array([libm.logf(float(v)) if v > 0 else 0.0 for v in en.ravel()], np.float32).reshape(en.shape)
    assert np.array_equal(got, want)
    # ... batched, the same energies per utterance
    en2, foff2 = ctx.wave_energies_staged([raw[:20000], raw, raw[:3000]])
    assert np.array_equal(en2[foff2[1]:foff2[2]], en)
    ctx.close()


def test_alaw_and_short_files(capi):
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ctx = _ctx(capi, CZ, wave_format="alaw")
    mel, _ = ctx.wave_to_mel([raw])
    want = read_htk(os.path.join(GOLD, "cli", "test_alaw.mel"))
    assert mel.shape == want.shape and np.abs(mel - want).max() <= TOL_MEL
    ctx = _ctx(capi, CZ)
    # several utterances in one call, incl. a 17-frame one, an odd byte count and one shorter than a frame
    blobs = [raw[:3000], raw[:20001], raw[:100], raw]
    mel, foff = ctx.wave_to_mel(blobs)
    assert list(foff) == [0, 17, 17 + 123, 17 + 123 + 1, 17 + 123 + 1 + 747]
    assert np.abs(mel[:17] - read_htk(os.path.join(GOLD, "cli", "utt_c.mel"))).max() <= TOL_MEL
    assert np.abs(mel[-747:] - read_htk(os.path.join(GOLD, CZ, "test.mel"))).max() <= TOL_MEL
    one, _ = ctx.wave_to_mel([raw[:20001]])
    assert np.array_equal(one, mel[17:17 + 123])
    assert np.isfinite(mel).all()


@pytest.mark.parametrize("system", [CZ, EN, HU, RU])
def test_wave_to_posteriors(capi, system):
    """waveform -> posteriors without leaving the GPU vs the reference's -t post dump"""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ctx = _ctx(capi, system)
    post, foff = ctx.wave_to_posteriors([raw])
    want = read_htk(os.path.join(GOLD, system, "test.lop"))
    assert post.shape == want.shape
    assert np.abs(post - want).max() < 1e-4
    if system == CZ:                                   # a list: the three pieces of the CLI goldens
        blobs = [raw, raw[:20000], raw[:3000]]
        post, foff = ctx.wave_to_posteriors(blobs)
        for k, name in enumerate(("utt_a", "utt_b", "utt_c")):
            w = read_htk(os.path.join(GOLD, "cli", name + ".lop"))
            assert np.abs(post[foff[k]:foff[k + 1]] - w).max() < 1e-4, name


def test_sentence_mean_orders(capi, oracle_mod):
    """lcrc_set_mean_order: the reference's sequential sums (srec.cpp:1500-1511, matrix.h:2101-2116; the default)
    vs the opt-in fixed-shape tree sums.  Both stay within the bar of the reference's dumps; the tree is batch-invariant (an
    utterance's posteriors do not depend on what else is in the call); on a long utterance (many 256-row blocks)
    both agree with the oracle fed with numpy's own mean-normalised features"""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    want = read_htk(os.path.join(GOLD, CZ, "test.lop"))
    ctx = _ctx(capi, CZ)
    ctx.set_hidden_split(1)
    got = {}
    for seq in (False, True):
        ctx.set_mean_order(seq)
        got[seq], _ = ctx.wave_to_posteriors([raw])
        assert np.abs(got[seq] - want).max() < 1e-4
    assert np.abs(got[False] - got[True]).max() < 5e-5
    ctx.set_mean_order(False)
    blobs = [raw[:30000], raw, raw[:100], raw[:50001]]
    post, foff = ctx.wave_to_posteriors(blobs)
    assert np.array_equal(post[foff[1]:foff[2]], got[False]), "batching must not change an utterance's mean"
    alone, _ = ctx.wave_to_posteriors([raw[:50001]])
    assert np.array_equal(post[foff[3]:foff[4]], alone)
    # a long utterance: 12 copies of the file = 8990 frames = 36 blocks
    long_raw = raw * 12
    mel, _ = ctx.wave_to_mel([long_raw])
    norm = (mel.astype(np.float64) - mel.astype(np.float64).mean(axis=0)).astype(np.float32)
    o = oracle_mod.Oracle(model_dir(CZ), 15)
    rows = slice(4000, 4064)
    ref = o.posteriors(norm[4000 - 15:4064 + 15])[15:15 + 64]
    for seq in (False, True):
        ctx.set_mean_order(seq)
        p, _ = ctx.wave_to_posteriors([long_raw])
        assert p.shape[0] == mel.shape[0]
        assert np.abs(p[rows] - ref).max() < 1e-4, seq


@pytest.mark.parametrize("system", [CZ, EN])
def test_sequential_sentence_mean_bit_for_bit(capi, system):
    """The default sentence mean IS the reference's: column sums added frame by frame in f32 (matrix.h:2101-2116), mean =
    sum * (1.0f / rows), x += -mean (srec.cpp:1500-1511).  Checked bit for bit through the posterior kernel: the waveform
    entry's posteriors must EQUAL those of the host-pointer entry fed with the front-end's own features normalised on the
    host in exactly that order (the posterior kernel is deterministic and, with the fused kernels pinned, batch-invariant,
    so equal posteriors on every row mean equal normalised features).  Short utterances (the means' workgroups subtract
    themselves), a long one (separate subtract kernel), a batch; 15 banks and -- the EN system with the normalisation
    switched on -- 23."""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ctx = _ctx(capi, system, sent_mean_norm=True)
    ctx.set_hidden_split(1)
    blobs = [raw, raw[:20000], raw * 6, raw[:1000]]
    mel, foff = ctx.wave_to_mel(blobs)                        # un-normalised features, as `-t par` dumps them
    assert foff[3] - foff[2] > 2048
    norm = np.empty_like(mel)
    for u in range(len(blobs)):
        x = mel[foff[u]:foff[u + 1]]
        total = np.add.accumulate(x, axis=0, dtype=np.float32)[-1]          # sequential f32 sums, frame by frame
        mean = total * (np.float32(1.0) / np.float32(len(x)))
        norm[foff[u]:foff[u + 1]] = x + (-mean)
    want = ctx.posteriors_batch(norm, np.asarray(foff, np.int32))
    got, foff2 = ctx.wave_to_posteriors(blobs)
    assert list(foff2) == list(foff)
    assert np.array_equal(got, want)
    for u in (0, 2):                                          # ... and alone
        one, _ = ctx.wave_to_posteriors([blobs[u]])
        assert np.array_equal(one, want[foff[u]:foff[u + 1]])
    ctx.close()


def test_default_mean_order_is_the_references(capi):
    """a fresh context sums the sentence mean in the reference's sequential order (ABI 2): identical bits to an
    explicit lcrc_set_mean_order(1), and batch-invariant"""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    a = _ctx(capi, CZ)
    a.set_hidden_split(1)
    dflt, _ = a.wave_to_posteriors([raw])
    a.set_mean_order(True)
    seq, _ = a.wave_to_posteriors([raw])
    assert np.array_equal(dflt, seq)
    post, foff = a.wave_to_posteriors([raw[:30000], raw, raw[:100]])
    assert np.array_equal(post[foff[1]:foff[2]], dflt)
    # launches of short utterances subtract the mean inside the means' own workgroups, launches with a long utterance
    # (> 2048 frames) keep the separate subtract kernel: the same bits either way
    long_one = raw * 4
    post2, foff2 = a.wave_to_posteriors([raw, long_one, raw[:30000]])
    assert foff2[2] - foff2[1] > 2048
    assert np.array_equal(post2[foff2[0]:foff2[1]], dflt)
    alone, _ = a.wave_to_posteriors([long_one])
    assert np.array_equal(post2[foff2[1]:foff2[2]], alone)
    a.close()


def test_config1_signal_through_the_waveform_entry(capi, oracle_mod):
    """BASELINE configs[1]'s input as SURVEY 8(d) cfg2 defines it (EN, 16 kHz lin16, 5 sines + noise, seed 1234,
    4096 frames, posterior-only) through lcrc_wave_to_posteriors -- bench.py's `wave_path_en` leg: every row against
    the oracle evaluated on the front-end's own features (which test_mel_matches_reference_dump ties to the
    reference's `-t par` dumps), and the two waveform entries agree bit for bit"""
    import bench
    raw = bench.config1_lin16_signal()
    assert len(raw) == 2 * 655600
    ctx = _ctx(capi, EN)
    mel, foff = ctx.wave_to_mel([raw])
    assert mel.shape == (4096, 23) and list(foff) == [0, 4096]
    assert 6.0 < float(mel.mean()) < 30.0 and np.isfinite(mel).all()    # log-mel energies of a 0.3 full-scale signal
    post, _ = ctx.wave_to_posteriors([raw])
    want = oracle_mod.Oracle(model_dir(EN), 23).posteriors(mel, threads=bench.usable_cpus())   # EN: no sentence mean norm
    err = np.abs(post - want).max(axis=1)
    assert post.shape == (4096, 120) and err.max() < 1e-4, (int(err.argmax()), float(err.max()))
    staged, _ = ctx.wave_to_posteriors_staged([raw])
    assert np.array_equal(staged, post)
    ctx.close()


def test_front_end_options_vs_host_front_end(capi, tmp_path):
    """pre-emphasis, z_mean_source, dc_shift, scale: the host CLI's front-end (itself bit-identical to the
    reference's) is the comparison"""
    import subprocess
    from tests.util import ROOT
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()[:16000]
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 32, 12, seed=1)
    cfg = open(os.path.join(d, "config")).read()
    cfg = cfg.replace("preem_coef=0.0", "preem_coef=0.97\nz_mean_source=true")
    cfg = cfg.replace("sample_freq=8000", "sample_freq=8000\nscale=0.5\ndc_shift=3.0")
    open(os.path.join(d, "config"), "w").write(cfg)
    (tmp_path / "x.raw").write_bytes(raw)
    subprocess.check_call([os.path.join(ROOT, "phnrec_amd", "bin", "phnrec"), "-c", d, "-i", str(tmp_path / "x.raw"),
                           "-t", "par", "-o", str(tmp_path / "x.mel")])
    want = read_htk(str(tmp_path / "x.mel"))
    ctx = capi.Lcrc(d, 15)
    ctx.configure_frontend(preem_coef=0.97, z_mean_source=True, scale=0.5, dc_shift=3.0)
    mel, _ = ctx.wave_to_mel([raw])
    assert mel.shape == want.shape and np.abs(mel - want).max() <= TOL_MEL


def test_frontend_errors(capi):
    ctx = capi.Lcrc(model_dir(CZ), 15)
    with pytest.raises(capi.LcrcError):
        ctx.wave_to_mel([b"\0" * 1000])                # not configured
    with pytest.raises(capi.LcrcError):
        ctx.configure_frontend(vector_size=1000)
    ctx.configure_frontend()
    assert ctx.frontend_frames(119846) == 747 and ctx.frontend_frames(10) == 1


def test_zero_copy_waveform_staging(capi):
    capi_mod = capi
    """lcrc_wave_stage_buffer / lcrc_wave_stage_run == lcrc_wave_to_posteriors (bit for bit), incl. odd byte
    counts, an empty file and buffer regrowth; misuse is rejected"""
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    ctx = capi_mod.Lcrc(model_dir(system), 15)
    ctx.configure_frontend(wave_format="lin16", sample_freq=8000, vector_size=200, vector_step=80,
                           lower_freq=64.0, higher_freq=4000.0, sent_mean_norm=True)
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    for blobs in ([raw[:4001], raw[:20000], b"", raw[:777]], [raw, raw[:30001]]):
        a, fa = ctx.wave_to_posteriors(blobs)
        b, fb = ctx.wave_to_posteriors_staged(blobs)
        assert np.array_equal(fa, fb) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    import ctypes as C
    buf = C.POINTER(C.c_ubyte)()
    assert ctx.L.lcrc_wave_stage_buffer(ctx.h, 1000, C.byref(buf)) == 0
    post = np.zeros((10, ctx.n_out), np.float32)
    foff = np.zeros(3, np.int32)
    bad = ctx.L.lcrc_wave_stage_run(ctx.h, np.array([0, 401], np.int64), np.array([400, 400], np.int64), 2, post, foff)
    assert bad == capi_mod.LCRC_E_ARG                          # odd start of a lin16 utterance
    bad = ctx.L.lcrc_wave_stage_run(ctx.h, np.array([0, 200], np.int64), np.array([400, 400], np.int64), 2, post, foff)
    assert bad == capi_mod.LCRC_E_ARG                          # overlap
    bad = ctx.L.lcrc_wave_stage_run(ctx.h, np.array([0], np.int64), np.array([1 << 30], np.int64), 1, post, foff)
    assert bad == capi_mod.LCRC_E_ARG                          # beyond the reserved capacity


def test_reserve_allocates_ahead_and_changes_nothing(capi):
    """lcrc_reserve: the buffers of later calls exist afterwards (the staging pointers handed out before a call of the
    reserved size are the ones handed out after it), results are the same bits as without it, misuse is rejected"""
    import ctypes as C
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    blobs = [raw, raw[:30001], raw[:4001]]
    plain = capi.Lcrc(model_dir(system), 15)
    with pytest.raises(capi.LcrcError):
        plain.reserve(100, 4, 1000)                            # waveform bytes before lcrc_frontend_configure
    with pytest.raises(capi.LcrcError):
        plain.reserve(-1, 4)
    fe = dict(wave_format="lin16", sample_freq=8000, vector_size=200, vector_step=80, lower_freq=64.0, higher_freq=4000.0,
              sent_mean_norm=True)
    plain.configure_frontend(**fe)
    want, fw = plain.wave_to_posteriors_staged(blobs)
    ctx = capi.Lcrc(model_dir(system), 15)
    ctx.configure_frontend(**fe)
    ctx.reserve(4096, 8, 400000)
    buf0, buf1 = C.POINTER(C.c_ubyte)(), C.POINTER(C.c_ubyte)()
    pm0, pp0, pm1, pp1 = (C.POINTER(C.c_float)() for _ in range(4))
    assert ctx.L.lcrc_wave_stage_buffer(ctx.h, 1000, C.byref(buf0)) == 0
    assert ctx.L.lcrc_stage_buffers(ctx.h, 16, C.byref(pm0), C.byref(pp0)) == 0
    got, fg = ctx.wave_to_posteriors_staged(blobs)             # 1000+ frames, 150 kB: within what was reserved
    assert np.array_equal(fw, fg) and np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert ctx.L.lcrc_wave_stage_buffer(ctx.h, 300000, C.byref(buf1)) == 0
    assert ctx.L.lcrc_stage_buffers(ctx.h, 4096, C.byref(pm1), C.byref(pp1)) == 0
    addr = lambda p: C.cast(p, C.c_void_p).value
    assert addr(buf0) == addr(buf1) and addr(pm0) == addr(pm1) and addr(pp0) == addr(pp1)
    ctx.reserve(0, 0)                                          # nothing to do
    # the frame entry and the decoder's buffers
    rng = np.random.default_rng(5)
    mel = rng.standard_normal((500, 15)).astype(np.float32)
    a = plain.posteriors(mel)
    ctx.configure_decoder(45, 3, 21, -3.8)
    ctx.reserve(4096, 8)
    b = ctx.posteriors(mel)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_decoder_overlapped_with_the_next_call_gives_the_same_labels(capi):
    """lcrc_set_decoder_overlap: staged calls return when their posterior kernels are done, the decoder of call k runs
    beside the kernels of call k + 1 on the context's second stream and second set of buffers; prev_labels() after call
    k + 1 (last_labels() after the last call) returns what last_labels() returns right after call k without the overlap --
    for the waveform entry (lcrc_wave_stage_run) and the frame entry (lcrc_stage_run), across calls of growing and
    shrinking size (buffers reallocated between the two sets), empty calls, and a switch back to synchronous decoding."""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()       # 7.5 s of speech (lin16, 8 kHz): real label sequences
    rng = np.random.default_rng(11)

    def blobs_of(sizes):                                          # slices of the utterance, `sizes` in samples
        out = []
        for n in sizes:
            o = 2 * int(rng.integers(0, len(raw) // 2 - n + 1))
            out.append(raw[o:o + 2 * n])
        return out

    calls = [blobs_of([4000, 9000, 200]), blobs_of([30000]), blobs_of([100, 100]), blobs_of([]), blobs_of([59000, 8000, 24000, 500]),
             blobs_of([8000]), blobs_of([59900, 4000] + [16000] * 6), blobs_of([2000, 2000, 2000])]

    def make():
        c = _ctx(capi, CZ)
        c.configure_output(("log",))
        c.configure_decoder(45, 3, 40, -4.6875)
        c.set_posterior_readback(False)
        return c

    ref = make()
    want = []
    for b in calls:
        ref.wave_decode_staged(b)
        want.append(ref.last_labels())
    assert sum(len(u) for w in want for u in w) > 200           # (real label sequences, not empty lists)
    ctx = make()
    ctx.set_decoder_overlap(True)
    got = []
    for k, b in enumerate(calls):
        ctx.wave_decode_staged(b)
        if k > 0:
            got.append(ctx.prev_labels())
    got.append(ctx.last_labels())
    assert got == want
    # once more on the same context (the sets keep alternating), fetching nothing in between except at the end of each pair
    for k in range(0, len(calls) - 1, 2):
        ctx.wave_decode_staged(calls[k])
        ctx.wave_decode_staged(calls[k + 1])
        assert ctx.prev_labels() == want[k] and ctx.last_labels() == want[k + 1]
    # the frame entry (lcrc_stage_run): features in, labels out
    mels, offs, want2 = [], [], []
    for k, lens in enumerate(([300, 0, 41, 900], [5000], [40, 40, 40], [2500, 1])):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mels.append(np.tile(read_htk(os.path.join(GOLD, CZ, "test.mel")), (8, 1))[:int(off[-1])] - 11.0)
        offs.append(off)
        ref.posteriors_staged(mels[-1], off)
        want2.append(ref.last_labels())
    got2 = []
    for k in range(len(mels)):
        ctx.posteriors_staged(mels[k], offs[k])
        if k > 0:
            got2.append(ctx.prev_labels())
    got2.append(ctx.last_labels())
    assert got2 == want2
    # back to synchronous decoding on the same context
    ctx.set_decoder_overlap(False)
    ctx.wave_decode_staged(calls[0])
    assert ctx.last_labels() == want[0]
    ref.close()
    ctx.close()


def test_device_ln_forms_against_this_hosts_logf(capi):
    """lcrc_frontend_set_ln's three forms on raw values (lcrc_device_ln): one of glibc's two logf sequences (with / without
    fused multiply-adds) must give THIS host's logf bit for bit on every probe -- all exponents, the ends of every table
    interval, values next to 1, subnormals, infinity, NaN, zero and negative values (sLn: 0) --, and log() in double
    rounded once stays within one ulp of it."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype, libm.logf.argtypes = ctypes.c_float, [ctypes.c_float]
    bits = []
    for k in range(-127, 129):
        for iv in range(16):
            for m in (0, 1, 0x3ffff, 0x40000, 0x7fffe, 0x7ffff):
                bits.append((0x3f330000 + (((k & 0x1ff) << 23) | (iv << 19) | m)) & 0xffffffff)
    bits += [0x3f800000 + d for d in range(-64, 65)]
    bits += [0, 0x80000000, 0xbf800000, 1, 0x000fffff, 0x00800000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000]
    rng = np.random.default_rng(3)
    bits = np.concatenate([np.array(bits, np.uint64), rng.integers(0, 1 << 32, 60000, dtype=np.uint64)]).astype(np.uint32)
    x = bits.view(np.float32)
    want = np.array([libm.logf(float(v)) if v > 0 else 0.0 for v in x], np.float32)
    got = [capi.device_ln(x, f) for f in (0, 1, 2)]
    same = [np.array_equal(g.view(np.uint32), want.view(np.uint32)) for g in got]
    assert same[1] or same[2], "neither of glibc's logf sequences gives this host's logf"
    finite = np.isfinite(want) & (x > 0)
    ulp = np.abs(got[0].view(np.int32)[finite].astype(np.int64) - want.view(np.int32)[finite].astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() > 0.999
    assert np.array_equal(got[0][~finite].view(np.uint32), want[~finite].view(np.uint32))      # zero, negative, NaN: 0; inf: inf


def test_decoder_overlap_is_inert_while_posteriors_are_read_back(capi):
    """lcrc_set_decoder_overlap applies to contexts that decode WITHOUT reading posteriors back: with read-back on, a staged
    call decodes before it returns, as always -- last_labels() right behind it is that call's, prev_labels() has nothing --,
    and the posteriors it hands out are the ones of a context that never heard of the overlap"""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    ref, ctx = _ctx(capi, CZ), _ctx(capi, CZ)
    for c in (ref, ctx):
        c.configure_output(("log",))
        c.configure_decoder(45, 3, 40, -4.6875)
    ctx.set_decoder_overlap(True)                               # read-back stays on (the default)
    mel = read_htk(os.path.join(GOLD, CZ, "test.mel")) - 11.0
    off = np.array([0, 300, 747], np.int32)
    a = ref.posteriors_staged(mel, off)
    b = ctx.posteriors_staged(mel, off)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert ctx.last_labels() == ref.last_labels() and len(ctx.last_labels()[1]) > 10
    assert ctx.prev_labels() == []
    pa, _ = ref.wave_to_posteriors_staged([raw[:40000], raw[40000:]])
    pb, _ = ctx.wave_to_posteriors_staged([raw[:40000], raw[40000:]])
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)) and ctx.last_labels() == ref.last_labels()
    ref.close()
    ctx.close()


def test_overlapped_decoder_is_settled_before_calls_of_other_entry_points(capi):
    """A decoder left running by an overlapped staged call is ordered only against LATER overlapped calls.  Mixing entry
    points is allowed (lcrc.h): a synchronous lcrc_posteriors_batch, a staged call with a caller's posterior buffer, or a
    staged call after lcrc_set_posterior_readback(1), right behind an overlapped call on the same context, first waits
    for that decoder -- the overlapped call's labels (fetched AFTER the other call) are the ones a synchronous context
    gives, the other call's own labels and posteriors too, and the context goes on overlapping afterwards.  Long
    utterances (59 900 samples = 747 frames, many per call) keep the decoder busy for milliseconds behind the call."""
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()
    big = [raw[:119800]] * 24                                   # 24 x 747 frames: a decoder launch of a few milliseconds
    small = [raw[2000:42000], raw[50000:70000]]
    mel = read_htk(os.path.join(GOLD, CZ, "test.mel")) - 11.0
    off = np.array([0, 300, 747], np.int32)

    def make():
        c = _ctx(capi, CZ)
        c.configure_output(("log",))
        c.configure_decoder(45, 3, 40, -4.6875)
        c.set_posterior_readback(False)
        return c

    ref = make()
    ref.wave_decode_staged(big)
    want_big = ref.last_labels()
    ref.wave_decode_staged(small)
    want_small = ref.last_labels()
    ref.set_posterior_readback(True)
    want_post = ref.posteriors_batch(mel, off)
    want_batch = ref.last_labels()
    want_wave_post, _ = ref.wave_to_posteriors_staged(small)
    ref.set_posterior_readback(False)
    assert sum(len(u) for u in want_big) > 1000 and len(want_batch[1]) > 10

    ctx = make()
    ctx.set_decoder_overlap(True)
    for _ in range(3):
        # (1) overlapped call, then the synchronous batch entry (decodes on the launch stream into the same label buffers)
        # (read-back is off on this context: the call returns labels only)
        ctx.wave_decode_staged(big)
        ctx.posteriors_batch(mel, off)
        assert ctx.last_labels() == want_batch
        # ... and with read-back on for the one call: its posteriors too
        ctx.wave_decode_staged(big)
        ctx.set_posterior_readback(True)
        got_post = ctx.posteriors_batch(mel, off)
        assert ctx.last_labels() == want_batch
        assert np.array_equal(got_post.view(np.uint32), want_post.view(np.uint32))
        ctx.set_posterior_readback(False)
        # (2) overlapped call, then read-back switched on: the staged call takes the synchronous road
        ctx.wave_decode_staged(big)
        ctx.set_posterior_readback(True)
        p, _ = ctx.wave_to_posteriors_staged(small)
        assert np.array_equal(p.view(np.uint32), want_wave_post.view(np.uint32)) and ctx.last_labels() == want_small
        ctx.set_posterior_readback(False)
        # (3) and overlapping again: two overlapped calls, both sets' labels
        ctx.wave_decode_staged(big)
        ctx.wave_decode_staged(small)
        assert ctx.prev_labels() == want_big and ctx.last_labels() == want_small
    ref.close()
    ctx.close()
