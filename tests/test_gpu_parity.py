"""HIP path (through the C ABI) vs the oracle and vs the reference's goldens.

Bar (BASELINE.json north_star): posteriors within 1e-4 max-abs per frame of the
reference CPU path.  TOL below is that bar; the measured differences are about
1e-6 (summation order of the f32 products).
"""
import os

import numpy as np
import pytest

from phnrec_amd import modelgen
from tests.util import GOLD, model_dir, read_htk

pytestmark = pytest.mark.gpu

TOL = 1e-4          # max-abs per frame, north_star
TOL_STAGE = 2e-5    # per-stage probes (tighter: each stage is compared at its own scale)


@pytest.fixture(scope="module")
def capi():
    from phnrec_amd import capi
    capi.load()
    return capi


def _norm_mel(ob, system):
    mel = read_htk(os.path.join(GOLD, system, "test.mel"))
    if modelgen.SYSTEMS[system]["sent_mean_norm"]:
        mel = ob.sentence_mean_norm(mel)
    return mel


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_real_system_vs_reference_golden(capi, oracle_mod, system):
    """bundled test.raw: GPU posteriors vs the reference CLI's own -t post dump"""
    spec = modelgen.SYSTEMS[system]
    mel = _norm_mel(oracle_mod, system)
    lop = read_htk(os.path.join(GOLD, system, "test.lop"))
    g = capi.Lcrc(model_dir(system), spec["nbanks"])
    assert g.n_out == spec["n_out"]
    assert g.kernel_name.startswith(("cz_", "en_", "hu_", "ru_"))      # a shape-specialised variant, never "generic"
    post = g.posteriors(mel)
    err = np.abs(post - lop).max(axis=1)
    assert err.max() < TOL, err.max()
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5
    assert np.array_equal(post.argmax(axis=1), lop.argmax(axis=1)) or \
        (post.argmax(axis=1) != lop.argmax(axis=1)).mean() < 0.005


@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_stage_probes_vs_reference(capi, oracle_mod, system):
    """each fused stage against the reference's own intermediates (Traps members)"""
    spec = modelgen.SYSTEMS[system]
    mel = _norm_mel(oracle_mod, system)
    g = capi.Lcrc(model_dir(system), spec["nbanks"])
    pr = g.posteriors_probe(mel)
    gold = np.load(os.path.join(GOLD, system, "probe.npz"))
    rows = gold["rows"]
    for key in ("in0", "in1", "p0", "p1"):
        err = np.abs(pr[key][rows] - gold[key]).max()
        scale = max(1.0, np.abs(gold[key]).max())
        assert err < TOL_STAGE * scale, (key, err)
    # ln() of tiny posteriors amplifies relative error: compare where p > 1e-30
    err = np.abs(pr["g"][rows] - gold["g"])
    assert err.max() < 1e-3, err.max()


def test_synthetic_goldens_all_shapes(capi, tmp_path):
    """reference outputs on seeded synthetic models: every shipped shape, the generic
    kernel on odd shapes, and utterances of 1..40 frames (edge replication)"""
    g = np.load(os.path.join(GOLD, "synth.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    seen = set()
    for name in names:
        nb, hid, nout, seed = [int(v) for v in g[name + "/dims"]]
        d = tmp_path / name
        nets = modelgen.write_model_dir(str(d), nb, hid, nout, seed=seed)
        assert bytes(g[name + "/digest"]).hex() == modelgen.nets_digest(nets)
        ctx = capi.Lcrc(str(d), nb)
        ctx.set_hidden_split(1)            # fused kernel only: batching must not change a single bit
        seen.add(ctx.kernel_name)
        off = g[name + "/off"]
        mel, want = g[name + "/mel"], g[name + "/post"]
        got = ctx.posteriors_batch(mel, off)
        assert np.abs(got - want).max() < TOL, (name, np.abs(got - want).max())
        # utterance by utterance through the single-utterance entry point
        for i in range(len(off) - 1):
            a, b = int(off[i]), int(off[i + 1])
            one = ctx.posteriors(mel[a:b])
            assert np.abs(one - want[a:b]).max() < TOL, (name, i)
            assert np.array_equal(one, got[a:b]), "batched and single launches must agree bit for bit"
        ctx.close()
    assert {"cz_42_69_9", "hu_42_93_12", "ru_42_80_10", "en_64_60_8"} <= seen and any(k.startswith("generic_") for k in seen), seen


def test_vs_oracle_random_batches(capi, oracle_mod, tmp_path):
    """seeded ragged batches incl. empty utterances, tile-straddling boundaries"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 200, 45, seed=21)
    o = oracle_mod.Oracle(d, 15)
    ctx = capi.Lcrc(d, 15)
    rng = np.random.default_rng(5)
    for trial in range(3):
        lens = [int(v) for v in rng.integers(0, 70, size=9)]
        lens[rng.integers(0, 9)] = 0
        lens[rng.integers(0, 9)] = 1
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mel = modelgen.synth_mel(int(off[-1]), 15, seed=100 + trial, mean_norm=False)
        want = o.posteriors_batch(mel, off)
        got = ctx.posteriors_batch(mel, off)
        assert np.abs(got - want).max() < TOL


def test_streaming_push_matches_traps_semantics(capi, oracle_mod, tmp_path):
    """lcrc_reset/lcrc_push/lcrc_delay == Traps::Reset/CalcFeaturesBunched/GetDelay"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 64, 24, seed=3)
    o = oracle_mod.Oracle(d, 15)
    ctx = capi.Lcrc(d, 15)
    mel = modelgen.synth_mel(90, 15, seed=8)
    # the ProcessOffline sequence (srec.cpp:1035-1059) through the streaming entry points
    ctx.reset()
    assert ctx.push(mel[:15], needed=False) is None
    assert ctx.delay() == 14
    main = ctx.push(mel[15:])
    tail = ctx.push(np.repeat(mel[-1:], 15, axis=0))
    got = np.concatenate([main, tail])
    want = o.posteriors(mel)
    assert np.abs(got - want).max() < TOL
    # arbitrary chunking gives the same frames as the oracle's ring buffer
    ctx.reset()
    o.reset()
    pos = 0
    for n in (1, 4, 5, 7, 30, 2):
        a = ctx.push(mel[pos:pos + n])
        b = o.push(mel[pos:pos + n], needed=True)
        assert np.abs(a - b).max() < TOL
        pos += n
        assert ctx.delay() == o.delay()


def test_streaming_bunches_of_five_over_a_long_stream(capi, oracle_mod, tmp_path):
    """the shipped bunch_size=5 (PHN_*/config:11, traps.cpp:518-535) over a stream long enough to wrap and to
    regrow the pinned strip: every bunch equals the whole-utterance rows it corresponds to (window ending at
    pushed frame i is centred at i - 15) and the oracle's ring buffer on spot checks"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 64, 24, seed=13)
    o = oracle_mod.Oracle(d, 15)
    ctx = capi.Lcrc(d, 15)
    n = 4700                                            # > the strip's 4096 rows: one wrap
    mel = modelgen.synth_mel(n, 15, seed=5)
    whole = ctx.posteriors(mel)
    ctx.reset()
    o.reset()
    got = []
    for pos in range(0, n, 5):
        a = ctx.push(mel[pos:pos + 5])
        got.append(a)
        if pos < 100 or pos % 1000 == 0:
            b = o.push(mel[pos:pos + 5], needed=True)
            assert np.abs(a - b).max() < TOL, pos
        else:
            o.push(mel[pos:pos + 5], needed=False)
    got = np.concatenate(got)
    assert np.abs(got[15:] - whole[:n - 15]).max() < 1e-5
    assert ctx.delay() == min(9999, n - 1)
    # a bunch larger than the strip makes it grow, history kept
    big = modelgen.synth_mel(9000, 15, seed=6)
    a = ctx.push(big)
    joined = np.concatenate([mel, big])
    want = ctx.posteriors(joined[n - 30:])[15:15 + 9000]
    assert np.abs(a - want).max() < 1e-5
    # pushes that are not needed only advance the history
    assert ctx.push(big[:7], needed=False) is None
    b = ctx.push(big[7:12])
    want = ctx.posteriors(np.concatenate([big[-30:], big[:12]]))[15 + 7:15 + 12]
    assert np.abs(b - want).max() < 1e-5


def test_split_hidden_path(capi, oracle_mod, tmp_path):
    """small launches spread every frame tile's hidden dimension over several workgroups (two launches, partial
    output tiles added by the last arriver in slice order): against the oracle and against the fused kernel for
    the shipped shapes and the generic kernel, every launch size class, forced and automatic splits; each
    setting is deterministic, the fused setting is bit-identical however the frames are batched"""
    cases = [(model_dir("PHN_CZ_SPDAT_LCRC_N1500"), 15, "cz_42_69_9"), (model_dir("PHN_EN_TIMIT_LCRC_N500"), 23, "en_64_60_8"),
             (model_dir("PHN_HU_SPDAT_LCRC_N1500"), 15, "hu_42_93_12"), (model_dir("PHN_RU_SPDAT_LCRC_N1500"), 15, "ru_42_80_10")]
    for idx, (nb, hid, nout, hm) in enumerate([(11, 70, 33, 70), (13, 40, 100, 61), (7, 1, 3, 5), (23, 300, 200, 320)]):
        d = str(tmp_path / ("g%d" % idx))
        modelgen.write_model_dir(d, nb, hid, nout, seed=40 + idx, hidden_merger=hm)
        cases.append((d, nb, "generic"))
    for d, nb, name in cases:
        ctx = capi.Lcrc(d, nb)
        assert ctx.kernel_name.startswith(name)
        o = oracle_mod.Oracle(d, nb)
        mel = modelgen.synth_mel(2100, nb, seed=3)
        for n in (1, 5, 16, 17, 100, 700, 2048, 2100):
            ctx.set_hidden_split(1)
            fused = ctx.posteriors(mel[:n])
            if n <= 100:
                assert np.abs(fused - o.posteriors(mel[:n])).max() < TOL, (name, n)
            for split in (0, 2, 5, 12, 64):
                ctx.set_hidden_split(split)
                got = ctx.posteriors(mel[:n])
                assert np.abs(got - fused).max() < 2e-5, (name, n, split, np.abs(got - fused).max())
                assert np.abs(got.sum(axis=1) - 1).max() < 1e-5
                assert np.array_equal(ctx.posteriors(mel[:n]), got), "deterministic for a given setting"
        # ragged batches incl. empty utterances; the writer path runs in the split path's epilogue too
        lens = [0, 3, 40, 0, 17, 1, 64]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        bm = mel[:int(off[-1])]
        want = o.posteriors_batch(bm, off)
        for split in (0, 1, 7):
            ctx.set_hidden_split(split)
            assert np.abs(ctx.posteriors_batch(bm, off) - want).max() < TOL, (name, split)
        ctx.set_hidden_split(0)
        ctx.configure_output(("log",), big_endian=True)
        with np.errstate(divide="ignore"):
            lw = np.log(want)
        got = ctx.posteriors_batch(bm, off).view(">f4").astype(np.float32)
        ok = np.isfinite(lw) & (want > 1e-30)
        assert np.abs(got[ok] - lw[ok]).max() < 1e-3
        ctx.configure_output(())
        with pytest.raises(capi.LcrcError):
            ctx.set_hidden_split(-1)
        ctx.close()


def test_row_ranges(capi, oracle_mod, tmp_path):
    """lcrc_posteriors_rows: only a row range of a strip is computed, the rest is context -- bit-identical to
    the same rows of the whole strip (fused kernel), so a long file cut into chunks with 15-frame halos gives
    the whole file's posteriors; edge replication applies at the strip's ends only"""
    for d, nb in ((model_dir("PHN_CZ_SPDAT_LCRC_N1500"), 15), (None, 9)):
        if d is None:
            d = str(tmp_path / "m")
            modelgen.write_model_dir(d, nb, 50, 21, seed=2)
        ctx = capi.Lcrc(d, nb)
        mel = modelgen.synth_mel(600, nb, seed=12)
        ctx.set_hidden_split(1)
        whole = ctx.posteriors(mel)
        for first, count in ((0, 600), (0, 1), (599, 1), (15, 5), (16, 31), (100, 333), (37, 0)):
            got = ctx.posteriors_rows(mel, first, count)
            assert got.shape == (count, ctx.n_out)
            assert np.array_equal(got, whole[first:first + count]), (first, count)
        # chunks of 128 rows with halos == the whole utterance
        parts = []
        for a in range(0, 600, 128):
            b = min(600, a + 128)
            lo, hi = max(0, a - 15), min(600, b + 15)
            parts.append(ctx.posteriors_rows(mel[lo:hi], a - lo, b - a))
        assert np.array_equal(np.concatenate(parts), whole)
        ctx.set_hidden_split(0)                          # small ranges take the split path: same values to the last bits
        for first, count in ((15, 5), (100, 333), (0, 600)):
            got = ctx.posteriors_rows(mel, first, count)
            assert np.abs(got - whole[first:first + count]).max() < 2e-5
        with pytest.raises(capi.LcrcError):
            ctx.posteriors_rows(mel, 590, 20)
        ctx.close()


def test_extreme_inputs_saturate_like_the_reference(capi, oracle_mod, tmp_path):
    """large-magnitude features drive sigmoids/softmax into FEXP's tails"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 128, 33, seed=4)
    o = oracle_mod.Oracle(d, 15)
    ctx = capi.Lcrc(d, 15)
    mel = modelgen.synth_mel(64, 15, seed=9) * np.float32(25.0)
    mel[10] = 300.0
    mel[11] = -300.0
    want = o.posteriors(mel)
    got = ctx.posteriors(mel)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() < TOL


def test_api_errors(capi, tmp_path):
    with pytest.raises(capi.LcrcError) as e:
        capi.Lcrc(str(tmp_path / "nope"), 15)
    assert e.value.code == capi.LCRC_E_IO and "Loading neural network" in str(e.value)
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 32, 12, seed=1)
    with pytest.raises(capi.LcrcError) as e:
        capi.Lcrc(d, 23)                     # wrong nbanks for these nets
    assert e.value.code == capi.LCRC_E_MODEL
    other = capi.Lcrc(d, 15, trap_len=21)    # another length: the general kernels (test_general_geometry)
    assert other.kernel_name == "lcrc_general"
    other.close()
    ctx = capi.Lcrc(d, 15)
    assert ctx.posteriors(np.zeros((0, 15), np.float32)).shape == (0, 12)
    with pytest.raises(capi.LcrcError):
        ctx.posteriors_batch(np.zeros((4, 15), np.float32), np.array([0, 3, 2, 4], np.int32))


def test_allocation_failure_leaves_the_context_usable(capi, tmp_path, monkeypatch):
    """lcrc_debug_fail_alloc: each of the staging allocations of a host-pointer call fails in turn (device and
    pinned, frame and offset buffers); the call returns LCRC_E_NOMEM, nothing stays half-allocated, and the
    next call on the same context succeeds with the right result.  The hook is inert without LCRC_FAULT_INJECTION=1."""
    assert capi.load().lcrc_debug_fail_alloc(0) == capi.LCRC_E_UNSUPPORTED
    monkeypatch.setenv("LCRC_FAULT_INJECTION", "1")
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 48, 20, seed=8)
    ctx = capi.Lcrc(d, 15)
    mel = modelgen.synth_mel(300, 15, seed=4)
    off = np.array([0, 100, 300], np.int32)
    want = ctx.posteriors_batch(mel, off)
    try:
        for nth in range(6):                            # 4 frame buffers + 2 offset buffers
            big_mel = np.concatenate([mel] * 2 ** (nth + 1))          # twice any earlier call: both groups regrow
            big_off = np.arange(0, big_mel.shape[0] + 1, 10, dtype=np.int32)
            big_off = np.concatenate([big_off, np.full(300 * 2 ** nth, big_off[-1], np.int32)])   # + empty utterances
            ctx.L.lcrc_debug_fail_alloc(nth)
            with pytest.raises(capi.LcrcError) as e:
                ctx.posteriors_batch(big_mel, big_off)
            assert e.value.code == capi.LCRC_E_NOMEM, (nth, str(e.value))
            ctx.L.lcrc_debug_fail_alloc(-1)
            assert np.array_equal(ctx.posteriors_batch(mel, off), want), nth
            got = ctx.posteriors_batch(big_mel, big_off)
            assert got.shape[0] == big_mel.shape[0] and np.abs(got.sum(axis=1) - 1).max() < 1e-5
    finally:
        ctx.L.lcrc_debug_fail_alloc(-1)
    ctx.close()


def test_device_pointer_entry_and_determinism(capi, tmp_path):
    import torch
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 96, 30, seed=2)
    ctx = capi.Lcrc(d, 15)
    mel = modelgen.synth_mel(1000, 15, seed=1)
    host = ctx.posteriors(mel)
    t_mel = torch.from_numpy(mel).cuda()
    t_post = torch.empty((1000, 30), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream()
    ctx.posteriors_device(t_mel.data_ptr(), 1000, t_post.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    assert np.array_equal(t_post.cpu().numpy(), host)
    assert ctx.last_kernel_ms() > 0
    again = ctx.posteriors(mel)
    assert np.array_equal(again, host), "the kernel must be run-to-run deterministic"


# ---- BASELINE.json full-size configurations: size-independent properties + oracle spot checks ----

def _spot_check(ctx, o, mel, post, starts, width=48):
    """oracle rows [a, a+width) of a long utterance need only mel[a-15 : a+width+15]"""
    n = mel.shape[0]
    for a in starts:
        lo, hi = max(0, a - 15), min(n, a + width + 15)
        want = o.posteriors(mel[lo:hi])[a - lo:a - lo + width]
        assert np.abs(post[a:a + width] - want).max() < TOL, a


@pytest.mark.parametrize("system,batch", [("PHN_EN_TIMIT_LCRC_N500", 4096), ("PHN_CZ_SPDAT_LCRC_N1500", 8192)])
def test_baseline_batch_sizes(capi, oracle_mod, system, batch):
    """configs[1] (EN, 4096 frames) and configs[2] (CZ, 8192 frames) at full size.  The oracle takes ~0.6 ms per
    frame, so only 4 x 48 rows (first, an early, a middle, the last window) are compared with it directly; the
    rest of the batch is covered by size-independent properties: every row is a distribution, the launch is
    idempotent, cutting the batch changes only the rows within 15 frames of the cut, and each part equals the
    stand-alone run of that part bit for bit"""
    spec = modelgen.SYSTEMS[system]
    nb = spec["nbanks"]
    mel = modelgen.synth_mel(batch, nb, seed=77, mean_norm=spec["sent_mean_norm"])
    ctx = capi.Lcrc(model_dir(system), nb)
    ctx.set_hidden_split(1)                # the bit-for-bit comparisons below span launch sizes
    o = oracle_mod.Oracle(model_dir(system), nb)
    post = ctx.posteriors(mel)
    assert np.isfinite(post).all() and (post >= 0).all()
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5                 # every frame is a distribution
    _spot_check(ctx, o, mel, post, [0, 17, batch // 2 - 5, batch - 48])
    # idempotence / determinism at full size
    assert np.array_equal(ctx.posteriors(mel), post)
    # locality: cutting the batch into two utterances changes only rows within 15 frames of the cut
    cut = batch // 2 + 7
    two = ctx.posteriors_batch(mel, np.array([0, cut, batch], np.int32))
    same = np.ones(batch, bool)
    same[cut - 15:cut + 15] = False
    assert np.array_equal(two[same], post[same])
    assert not np.array_equal(two[cut - 15:cut + 15], post[cut - 15:cut + 15])
    # and each half equals the stand-alone run of that half
    assert np.array_equal(two[:cut], ctx.posteriors(mel[:cut]))
    assert np.array_equal(two[cut:], ctx.posteriors(mel[cut:]))


@pytest.mark.parametrize("system,batch", [("PHN_EN_TIMIT_LCRC_N500", 4096), ("PHN_CZ_SPDAT_LCRC_N1500", 8192)])
def test_baseline_batches_every_row_against_the_oracle(capi, oracle_mod, system, batch):
    """configs[1] / configs[2] at full size, EVERY row against the oracle (its frames spread over the host's usable
    cores: a few seconds), for the launch forms a batch of that size can take: the launcher's own choice, forced
    16- and 32-frame workgroups, and the batch cut into ragged utterances"""
    import bench
    spec = modelgen.SYSTEMS[system]
    nb = spec["nbanks"]
    mel = modelgen.synth_mel(batch, nb, seed=78, mean_norm=spec["sent_mean_norm"])
    o = oracle_mod.Oracle(model_dir(system), nb)
    want = o.posteriors(mel, threads=bench.usable_cpus())
    assert want.shape[0] == batch and np.abs(want.sum(axis=1) - 1).max() < 1e-4
    ctx = capi.Lcrc(model_dir(system), nb)
    for frames in (0, 16, 32):
        ctx.set_tile_frames(frames)
        err = np.abs(ctx.posteriors(mel) - want).max(axis=1)
        assert err.max() < TOL, (frames, int(err.argmax()), float(err.max()))
    ctx.set_tile_frames(0)
    rng = np.random.default_rng(3)
    cuts = np.sort(rng.choice(np.arange(1, batch), size=23, replace=False))
    off = np.concatenate([[0], cuts, [batch]]).astype(np.int32)
    want_b = np.concatenate([o.posteriors(mel[a:b], threads=bench.usable_cpus()) for a, b in zip(off[:-1], off[1:])])
    assert np.abs(ctx.posteriors_batch(mel, off) - want_b).max() < TOL
    ctx.close()


def test_clone_shares_the_model_and_outlives_its_source(capi, oracle_mod):
    """lcrc_clone: a further context over the same weights on the same GPU (what the CLI's second and third context per
    GPU are).  Same bits as the source; its own settings, staging and streaming state; usable from another thread at the
    same time; and it keeps the shared device buffers alive after the source is destroyed.  lcrc_device_warmup is the
    start-up half a caller may run on a helper thread."""
    import threading
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    assert capi.load().lcrc_device_warmup(0) == 0
    assert capi.load().lcrc_device_warmup(4096) == capi.LCRC_E_DEVICE
    # the PCI address the CLI places a GPU's threads by ("0000:c1:00.0"); a sysfs node of that name exists
    import ctypes
    buf = ctypes.create_string_buffer(64)
    assert capi.load().lcrc_device_pci_bus_id(0, buf, 64) == 0
    bus = buf.value.decode().lower()
    assert len(bus.split(":")) == 3 and os.path.isdir("/sys/bus/pci/devices/" + bus), bus
    assert capi.load().lcrc_device_pci_bus_id(0, buf, 4) == capi.LCRC_E_ARG
    assert capi.load().lcrc_device_pci_bus_id(4096, buf, 64) == capi.LCRC_E_DEVICE
    a = capi.Lcrc(model_dir(system), 15)
    a.set_hidden_split(1)
    b = a.clone()
    assert b.n_out == a.n_out and b.kernel_name == a.kernel_name and b.net_dims(2) == a.net_dims(2)
    b.set_hidden_split(1)
    mel = modelgen.synth_mel(700, 15, seed=31, mean_norm=True)
    want = a.posteriors(mel)
    assert np.array_equal(b.posteriors(mel), want)
    # settings are per context: the clone in the split-f16 arithmetic, the source stays on f32
    b.set_arithmetic(capi.ARITH_SPLIT_F16)
    assert np.array_equal(a.posteriors(mel), want)
    assert 0 < np.abs(b.posteriors(mel) - want).max() < 2e-5
    b.set_arithmetic(capi.ARITH_F32)
    # both at once, each on its own thread / stream / staging buffers
    out = {}

    def work(ctx, key, seed):
        m = modelgen.synth_mel(3000, 15, seed=seed, mean_norm=True)
        out[key] = (m, [ctx.posteriors(m) for _ in range(5)])

    ts = [threading.Thread(target=work, args=(a, "a", 1)), threading.Thread(target=work, args=(b, "b", 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for key, ctx in (("a", a), ("b", b)):
        m, res = out[key]
        assert all(np.array_equal(r, res[0]) for r in res)
    o = oracle_mod.Oracle(model_dir(system), 15)
    rows = slice(1000, 1064)
    assert np.abs(out["b"][1][0][rows] - o.posteriors(out["b"][0][1000 - 15:1064 + 15])[15:15 + 64]).max() < TOL
    # streaming state is the clone's own
    a.reset()
    b.reset()
    pa = a.push(mel[:40])
    assert np.array_equal(b.push(mel[:40]), pa)
    a.close()                                   # the clone holds the weights from here on
    assert np.array_equal(b.posteriors(mel), want)
    c = b.clone()                               # (a clone starts from the default settings: automatic hidden split)
    c.set_hidden_split(1)
    b.close()
    assert np.array_equal(c.posteriors(mel), want)
    c.close()


def test_cut_launches_give_the_bits_of_uncut_ones(capi):
    """lcrc_launch cuts a launch into whole rounds of 32-frame workgroups plus a cheaper tail (16-frame workgroups, or the
    split-hidden path when that is allowed).  Sizes around every cut point: with the fused kernels only (as the CLI runs)
    the result is bit-identical to the uncut launch of forced 32-frame tiles -- also for ragged utterances that straddle
    the cuts; with the split tail allowed, within its usual distance of the fused kernels"""
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    ctx = capi.Lcrc(model_dir(system), 15)
    rng = np.random.default_rng(11)
    for n in (4097, 4112, 5000, 6144, 6200, 8193, 8200, 12288, 12305, 16500):
        mel = modelgen.synth_mel(n, 15, seed=n, mean_norm=True)
        cuts = np.sort(rng.choice(np.arange(1, n), size=9, replace=False))
        off = np.concatenate([[0], cuts, [n]]).astype(np.int32)
        ctx.set_hidden_split(1)
        ctx.set_tile_frames(32)
        want, want_b = ctx.posteriors(mel), ctx.posteriors_batch(mel, off)
        ctx.set_tile_frames(0)
        assert np.array_equal(ctx.posteriors(mel), want), n
        assert np.array_equal(ctx.posteriors_batch(mel, off), want_b), n
        ctx.set_hidden_split(0)
        got = ctx.posteriors(mel)
        assert np.abs(got - want).max() < 1e-5 and np.abs(got.sum(axis=1) - 1).max() < 1e-5, n
        assert np.abs(ctx.posteriors_batch(mel, off) - want_b).max() < 1e-5, n
    ctx.close()


def test_fuzzed_models(capi, oracle_mod):
    """tools/fuzz_parity.py with a fixed seed: 30 random model geometries (run-time-shape kernels: banks, hidden sizes,
    outputs, ragged batches with empty utterances, 16- / 32-frame workgroups, forced hidden splits) and 12 models of
    the shipped shape classes in the split-f16 arithmetic -- each against the oracle at the 1e-4 bar"""
    from tools import fuzz_parity
    ran, worst, worst_s, worst_d = fuzz_parity.fuzz(seed=20261004, n_models=30, n_h2=12, log=lambda *a: None)
    assert ran >= 20 and worst < TOL and worst_s < TOL and worst_d < 5e-5
    # ... and 25 random geometries of the general kernels (any length 2..64, C0 on / off, 1..16 values per band, all four
    # systems) against the run-time-geometry oracle
    assert fuzz_parity.fuzz_geometry(seed=20261004, n_models=25, log=lambda *a: None) < TOL


def test_sharded_file_list_shape(capi, oracle_mod, tmp_path):
    """configs[3]-like: the shipped HU weights, a list of ragged 3-15 s utterances in multi-utterance launches"""
    d = model_dir("PHN_HU_SPDAT_LCRC_N1500")
    ctx = capi.Lcrc(d, 15)
    ctx.set_hidden_split(1)                # as the CLI does: outputs independent of the packing into launches
    assert ctx.kernel_name == "hu_42_93_12"
    o = oracle_mod.Oracle(d, 15)
    rng = np.random.default_rng(9)
    lens = rng.integers(300, 1501, size=40)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    mel = np.concatenate([modelgen.synth_mel(int(n), 15, seed=500 + i, mean_norm=True) for i, n in enumerate(lens)])
    post = ctx.posteriors_batch(mel, off)
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5
    for u in (0, 7, 39):                                          # whole utterances, bit for bit
        a, b = int(off[u]), int(off[u + 1])
        assert np.array_equal(post[a:b], ctx.posteriors(mel[a:b]))
    u = 3                                                          # first / last frames of an utterance vs the oracle
    a, b = int(off[u]), int(off[u + 1])
    want_head = o.posteriors(mel[a:a + 40 + 15])[:40]
    want_tail = o.posteriors(mel[b - 40 - 15:b])[-40:]
    assert np.abs(post[a:a + 40] - want_head).max() < TOL
    assert np.abs(post[b - 40:b] - want_tail).max() < TOL


def test_large_launch(capi):
    """a 200k-frame launch (6250 workgroups): grid-stride independence, no overflow in addressing"""
    system = "PHN_EN_TIMIT_LCRC_N500"
    ctx = capi.Lcrc(model_dir(system), 23)
    mel = np.tile(modelgen.synth_mel(1000, 23, seed=3, mean_norm=False), (200, 1))
    post = ctx.posteriors(mel)
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5
    # interior repetitions of the 1000-frame block see identical contexts except across block seams
    assert np.array_equal(post[1000 + 15:2000 - 15], post[150000 + 15:151000 - 15])


def test_sleeping_waits_change_nothing(capi, tmp_path):
    """lcrc_set_wait_mode: the host-pointer entries wait for the device by querying an event between short sleeps instead of
    spinning in hipStreamSynchronize -- same results on every entry point, bad intervals refused"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 96, 30, seed=4)
    ctx = capi.Lcrc(d, 15)
    lens = [40, 0, 300, 7]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    mel = modelgen.synth_mel(int(off[-1]), 15, seed=2)
    want = ctx.posteriors_batch(mel, off)
    ctx.reset()
    pushed = ctx.push(mel[:20])
    for us in (1, 50, 1000, 0):
        ctx.set_wait_mode(us)
        assert np.array_equal(ctx.posteriors_batch(mel, off), want)
        assert np.array_equal(ctx.posteriors_staged(mel, off), want)
        assert np.array_equal(ctx.posteriors(mel[:40]), want[:40])
        ctx.reset()
        assert np.array_equal(ctx.push(mel[:20]), pushed)
    for bad in (-1, 100001):
        with pytest.raises(capi.LcrcError):
            ctx.set_wait_mode(bad)
    ctx.close()


def test_staged_zero_copy_entry_equals_batch(capi, tmp_path):
    """lcrc_stage_buffers / lcrc_stage_run (what the CLI uses) vs lcrc_posteriors_batch, incl. buffer regrowth"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 80, 27, seed=6)
    ctx = capi.Lcrc(d, 15)
    for lens in ([40, 3, 0, 25], [700, 1, 900], [5]):
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mel = modelgen.synth_mel(int(off[-1]), 15, seed=len(lens))
        assert np.array_equal(ctx.posteriors_staged(mel, off), ctx.posteriors_batch(mel, off))
    assert ctx.posteriors_staged(np.zeros((0, 15), np.float32), np.array([0, 0], np.int32)).shape == (0, 27)


def test_generic_kernel_shape_sweep(capi, oracle_mod, tmp_path):
    """the generic variant over awkward shapes: hidden sizes that leave waves without tiles or with a
    partial last tile, band/merger nets with different hidden sizes, every ksteps % 4, 1..23 banks,
    output counts around the 16-tile boundaries -- each against the oracle on ragged batches"""
    rng = np.random.default_rng(2024)
    cases = [(3, 5, 6), (4, 16, 16), (5, 17, 15), (7, 33, 17), (9, 48, 31), (11, 64, 32), (13, 70, 33),
             (15, 100, 47), (16, 130, 48), (19, 40, 49), (21, 250, 90), (23, 96, 120), (15, 1, 3), (8, 15, 208)]
    for idx, (nb, hid, nout) in enumerate(cases):
        d = str(tmp_path / ("m%d" % idx))
        modelgen.write_model_dir(d, nb, hid, nout, seed=100 + idx, hidden_merger=hid + 7 * (idx % 3))
        ctx = capi.Lcrc(d, nb)
        # (23 banks, 120 outputs) has the k-steps / output tiles of the EN variant: hidden size is a run-time value there
        assert ctx.kernel_name == "en_64_60_8" if (nb, nout) == (23, 120) else ctx.kernel_name.startswith("generic_")
        o = oracle_mod.Oracle(d, nb)
        lens = [int(v) for v in rng.integers(0, 45, size=5)] + [33]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mel = modelgen.synth_mel(int(off[-1]), nb, seed=idx, mean_norm=bool(idx & 1))
        got = ctx.posteriors_batch(mel, off)
        want = o.posteriors_batch(mel, off)
        assert np.abs(got - want).max() < TOL, (nb, hid, nout, np.abs(got - want).max())
        ctx.close()


def test_posterior_writer_path_on_device(capi, tmp_path):
    """lcrc_output_configure ("next" row f2): the softening functions of srec.cpp:164-176 and the HTK byte
    order applied in the kernel's epilogue, against the same f32 expressions evaluated by numpy on the plain
    posteriors of the same context (log-domain tolerance 2e-6 = a few ulp of logf at |ln p| <= 20)"""
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 90, 45, seed=11)
    ctx = capi.Lcrc(d, 15)
    off = np.array([0, 37, 37, 150], np.int32)
    mel = modelgen.synth_mel(150, 15, seed=3)
    plain = ctx.posteriors_batch(mel, off)
    f32 = np.float32

    def igor(v, mid, right, left):
        lo = np.log(v * (f32(1) / f32(mid))) / np.log(f32(left))
        hi = f32(-1) * np.log((f32(1) + f32(-1) * v) * (f32(1) / (f32(1) - f32(mid)))) / np.log(f32(right))
        return np.where(v < f32(mid), lo, hi).astype(np.float32)

    with np.errstate(divide="ignore", invalid="ignore"):
        cases = [
            ((), plain),
            (("log",), np.log(plain)),
            (("none", "log"), np.log(plain)),
            (("gmm_bypass",), np.sqrt(f32(-2) * np.log(plain))),
            ((("igor", 0.5, 10.0, 10.0),), igor(plain, 0.5, 10.0, 10.0)),
            (("log", "none"), np.log(plain)),
        ]
        for stages, want in cases:
            for be in (False, True):
                ctx.configure_output(stages, big_endian=be)
                got = ctx.posteriors_batch(mel, off)
                if be:
                    got = got.view(">f4").astype(np.float32)
                ok = np.isfinite(want)
                assert np.array_equal(np.isfinite(got), ok), stages
                assert np.abs(got[ok] - want[ok]).max() <= 2e-6 * max(1.0, np.abs(want[ok]).max()), (stages, be)
                # the staged entry (what the CLI uses) goes through the same epilogue
                st = ctx.posteriors_staged(mel, off)
                assert np.array_equal(st.view(np.uint32), ctx.posteriors_batch(mel, off).view(np.uint32))
    ctx.configure_output((), big_endian=False)
    assert np.array_equal(ctx.posteriors_batch(mel, off), plain)
    with pytest.raises(capi.LcrcError):
        ctx.configure_output(("log", "log", "log"))


def test_writer_path_vs_reference_dumps(capi):
    """lcrc_output_configure against the REFERENCE CLI's `-t post` dumps with posteriors/softening_func = log,
    igor (two parameter sets), gmm_bypass (tools/make_golden_softening.py; srec.cpp:164-176,1062-1070): waveform
    in, softened big-endian posteriors out, fused kernel and split-hidden path"""
    from tools.make_golden_softening import CASES
    raw = open(os.path.join(GOLD, "test.raw"), "rb").read()[:20000]
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    ctx = capi.Lcrc(model_dir(system), 15)
    ctx.configure_frontend(wave_format="lin16", sent_mean_norm=True)
    for name, value in CASES.items():
        f = value.split()
        stage = f[0] if f[0] != "igor" else ("igor", float(f[1]), float(f[2]), float(f[3]))
        want = read_htk(os.path.join(GOLD, "cli", "soft_%s.lop" % name))
        for split in (1, 0):
            ctx.set_hidden_split(split)
            ctx.configure_output((stage,), big_endian=True)
            got, _ = ctx.wave_to_posteriors([raw])
            got = got.view(">f4").astype(np.float32)
            # log-type functions turn the RELATIVE error of a posterior (up to ~5e-5 at p ~ 1e-10, far inside
            # the 1e-4 absolute bar) into an absolute one
            assert got.shape == want.shape and np.abs(got - want).max() < 1e-3, (name, split)
    ctx.close()


def test_both_workgroup_tile_sizes(capi, oracle_mod, tmp_path):
    """16- and 32-frame workgroups (lcrc_set_tile_frames; the launcher picks by launch size) on the same
    ragged batches, shipped-shape (ring loop) and generic kernels: both within the bar of the oracle and
    bit-identical to each other (the softmax sums are grouped the same way in both)"""
    for nb, hid, nout, name in ((15, 200, 138, "cz_42_69_9"), (23, 96, 120, "en_64_60_8"), (11, 70, 33, "generic")):
        d = str(tmp_path / name)
        modelgen.write_model_dir(d, nb, hid, nout, seed=21)
        ctx = capi.Lcrc(d, nb)
        ctx.set_hidden_split(1)
        assert ctx.kernel_name.startswith(name)
        o = oracle_mod.Oracle(d, nb)
        lens = [1, 15, 16, 17, 31, 32, 33, 0, 47, 100]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mel = modelgen.synth_mel(int(off[-1]), nb, seed=9)
        want = o.posteriors_batch(mel, off)
        got = {}
        for frames in (16, 32, 0):
            ctx.set_tile_frames(frames)
            got[frames] = ctx.posteriors_batch(mel, off)
            assert np.abs(got[frames] - want).max() < TOL, (name, frames)
        assert np.array_equal(got[16], got[32])
        assert np.array_equal(got[0], got[16])
        with pytest.raises(capi.LcrcError):
            ctx.set_tile_frames(24)
        ctx.close()


def test_general_geometry(capi, oracle_mod, tmp_path):
    """Geometries the reference's Traps accepts but no shipped model uses (missing #4): posteriors/length other
    than 31 -- odd and even --, LCRC without C0 or with another number of coefficients per band, and the other systems
    at other lengths.  They run on the general features kernel + the MLP kernels (three launches), checked against the
    reference's own outputs (tests/golden/geometry.npz) and the run-time-geometry oracle, on ragged batches with empty
    utterances, through the staged entry, a clone, the streaming form and the device-side writer path."""
    from tools.make_golden_systems import GEOM_CASES, write_geometry_model
    gold = np.load(os.path.join(GOLD, "geometry.npz"))
    for name, system, nb, hid, nout, seed, kw, lens, trap_len in GEOM_CASES:
        d = str(tmp_path / name)
        write_geometry_model(d, system, nb, hid, nout, seed, kw, trap_len)
        add_c0, hamming = kw.get("add_c0", True), kw.get("hamming", False)
        ctx = capi.Lcrc(d, nb, system=system, add_c0=add_c0, hamming=hamming, trap_len=trap_len)
        assert ctx.kernel_name == ("lcrc_general" if system == "LCRC" else "traps_" + system.lower()) and ctx.n_out == nout
        assert ctx.L.lcrc_trap_shift(ctx.h) == (trap_len - 1) // 2
        mel, off = gold[name + "/mel"], gold[name + "/off"]
        got = ctx.posteriors_batch(mel, off)
        assert np.abs(got - gold[name + "/post"]).max() < TOL, (name, np.abs(got - gold[name + "/post"]).max())
        o = oracle_mod.TrapsOracle(d, system, nb, add_c0, hamming, trap_len=trap_len)
        lens2 = [0, 5, 300, 0, 64, 1]
        off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
        mel2 = modelgen.synth_mel(int(off2[-1]), nb, seed=seed)
        got2 = ctx.posteriors_batch(mel2, off2)
        assert np.abs(got2 - o.posteriors_batch(mel2, off2)).max() < TOL, name
        assert np.abs(got2.sum(axis=1) - 1).max() < 1e-5
        assert np.array_equal(ctx.posteriors_batch(mel2, off2), got2)                  # deterministic
        assert np.array_equal(ctx.posteriors_staged(mel2, off2), got2)
        twin = ctx.clone()
        assert np.array_equal(twin.posteriors_batch(mel2, off2), got2)
        twin.close()
        # streaming form == whole-utterance form: the estimate of frame r appears when frame r + (L - 1) / 2 is pushed
        shift = (trap_len - 1) // 2
        u = mel2[int(off2[2]):int(off2[3])][:60]
        whole = ctx.posteriors(u)
        ctx.reset()
        pushed = np.concatenate([ctx.push(u[i:i + 7]) for i in range(0, len(u), 7)])
        tail = ctx.push(np.repeat(u[-1:], shift, axis=0))
        assert np.abs(np.concatenate([pushed, tail])[shift:] - whole).max() < 1e-6, name
        ctx.configure_output(("log",), big_endian=True)
        with np.errstate(divide="ignore"):
            want = np.log(got2)
        be = ctx.posteriors_batch(mel2, off2).view(">f4").astype(np.float32)
        ok = np.isfinite(want)
        assert np.abs(be[ok] - want[ok]).max() <= 2e-6 * max(1.0, np.abs(want[ok]).max())
        ctx.configure_output(())
        # what exists for the fused LCRC kernel only
        with pytest.raises(capi.LcrcError) as e:
            ctx.posteriors_rows(mel2[:40], 5, 10)
        assert e.value.code == capi.LCRC_E_UNSUPPORTED
        ctx.close()
    # band nets wider than the many-nets kernels' 256 inputs (23 banks x 16 values), and the shortest length there is
    for nb, coefs, L, c0 in ((23, 16, 31, False), (9, 2, 2, True)):
        d = str(tmp_path / ("wide%d" % L))
        modelgen.write_model_dir(d, nb, 70, 33, seed=60 + L, coefs=coefs, trap_len=L, add_c0=c0)
        ctx = capi.Lcrc(d, nb, trap_len=L, add_c0=c0)
        o = oracle_mod.TrapsOracle(d, "LCRC", nb, c0, False, trap_len=L)
        mel = modelgen.synth_mel(90, nb, seed=L)
        off = np.array([0, 50, 50, 90], np.int32)
        assert np.abs(ctx.posteriors_batch(mel, off) - o.posteriors_batch(mel, off)).max() < TOL
        ctx.close()
    # lengths outside 2 .. 255, and band nets whose inputs nbanks does not divide
    d = str(tmp_path / "bad")
    modelgen.write_model_dir(d, 15, 32, 21, seed=1)
    for L in (1, 0, 256):
        with pytest.raises(capi.LcrcError) as e:
            capi.Lcrc(d, 15, trap_len=L)
        assert e.value.code == capi.LCRC_E_ARG
    with pytest.raises(capi.LcrcError) as e:
        capi.Lcrc(d, 14)
    assert e.value.code == capi.LCRC_E_MODEL


def test_other_posterior_systems(capi, oracle_mod, tmp_path):
    """posteriors/system = 1BT_DCT, 1BT, 3BT ("next" row f4) through lcrc_create_system: the feature kernel +
    MLP kernel composition against the reference's goldens (tests/golden/systems.npz) and the oracle, the
    streaming form, the device-side writer path, and the calls that exist for LCRC only"""
    from tools.make_golden_systems import CASES
    gold = np.load(os.path.join(GOLD, "systems.npz"))
    for name, system, nb, hid, nout, seed, kw, lens in CASES:
        d = str(tmp_path / name)
        modelgen.write_traps_dir(d, system, nb, hid, nout, seed=seed, **kw)
        add_c0, hamming = kw.get("add_c0", True), kw.get("hamming", False)
        ctx = capi.Lcrc(d, nb, system=system, add_c0=add_c0, hamming=hamming)
        assert ctx.kernel_name == "traps_" + system.lower() and ctx.n_out == nout
        mel, off = gold[name + "/mel"], gold[name + "/off"]
        got = ctx.posteriors_batch(mel, off)
        assert np.abs(got - gold[name + "/post"]).max() < TOL, (name, np.abs(got - gold[name + "/post"]).max())
        assert np.abs(got.sum(axis=1) - 1).max() < 1e-5
        o = oracle_mod.TrapsOracle(d, system, nb, add_c0, hamming)
        # a longer ragged batch than the goldens hold, incl. empty utterances and a multi-workgroup one
        lens2 = [0, 5, 300, 0, 64, 1]
        off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
        mel2 = modelgen.synth_mel(int(off2[-1]), nb, seed=seed)
        got2 = ctx.posteriors_batch(mel2, off2)
        assert np.abs(got2 - o.posteriors_batch(mel2, off2)).max() < TOL, name
        assert np.array_equal(ctx.posteriors_batch(mel2, off2), got2)                  # deterministic
        assert np.array_equal(ctx.posteriors_staged(mel2, off2), got2)
        # 16- and 32-frame workgroups give the same bits
        for frames in (16, 32):
            ctx.set_tile_frames(frames)
            assert np.array_equal(ctx.posteriors_batch(mel2, off2), got2), (name, frames)
        ctx.set_tile_frames(0)
        # the separate-launch forms (PHNREC_TRAPS_UNFUSED=1: features kernel, band nets with grid.y = band, merger) against
        # the one-launch defaults: the last bits differ -- 1BT_DCT's fused projection runs as MFMA products (fused
        # multiply-adds) where the features kernel multiplies and adds separately; 1BT / 3BT's fused kernel groups the band
        # nets' softmax sums differently (four strided partials per frame)
        os.environ["PHNREC_TRAPS_UNFUSED"] = "1"
        try:
            ctx2 = capi.Lcrc(d, nb, system=system, add_c0=add_c0, hamming=hamming)
        finally:
            del os.environ["PHNREC_TRAPS_UNFUSED"]
        other = ctx2.posteriors_batch(mel2, off2)
        assert np.abs(other - got2).max() < 5e-6 and np.abs(other - o.posteriors_batch(mel2, off2)).max() < TOL, name
        ctx2.close()
        # streaming form == whole-utterance form (window ending at pushed frame i is centred at i - 15)
        u = mel2[int(off2[2]):int(off2[3])][:60]
        whole = ctx.posteriors(u)
        ctx.reset()
        pushed = np.concatenate([ctx.push(u[i:i + 7]) for i in range(0, len(u), 7)])
        tail = ctx.push(np.repeat(u[-1:], 15, axis=0))
        assert np.abs(np.concatenate([pushed, tail])[15:] - whole).max() < 1e-6
        # writer path in the merger's epilogue
        ctx.configure_output(("log",), big_endian=True)
        with np.errstate(divide="ignore"):
            want = np.log(got2)
        be = ctx.posteriors_batch(mel2, off2).view(">f4").astype(np.float32)
        ok = np.isfinite(want)
        assert np.abs(be[ok] - want[ok]).max() <= 2e-6 * max(1.0, np.abs(want[ok]).max())
        ctx.configure_output(())
        with pytest.raises(capi.LcrcError):
            ctx.posteriors_probe(u)
        ctx.close()
    # what the reference rejects or cannot run is an error here too, with its message
    d = str(tmp_path / "bad")
    modelgen.write_traps_dir(d, "1BT", 15, 30, 12, seed=1)
    with pytest.raises(capi.LcrcError, match="Unknown posterior estimator system"):
        capi.Lcrc(d, 15, system="2BT")
    os.remove(os.path.join(d, "weights", "band14.nbin"))
    with pytest.raises(capi.LcrcError, match="ERROR: Loading neural network"):
        capi.Lcrc(d, 15, system="1BT")
    with pytest.raises(capi.LcrcError, match="merger input size"):     # 3BT over 16 banks: 14 nets x 12 != 180
        capi.Lcrc(d, 16, system="3BT")


def test_kernel_done_callback(capi, tmp_path):
    """lcrc_set_kernel_done_callback: fn(arg) on the calling thread, once per call that launches, before the call returns
    (host-pointer, batch, staged and streaming entries; not for calls that launch nothing; off again with NULL)"""
    import ctypes as C
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 64, 138, seed=3)
    ctx = capi.Lcrc(d, 15)
    L = ctx.L
    calls = []
    FN = C.CFUNCTYPE(None, C.c_void_p)
    fn = FN(lambda arg: calls.append(arg))
    L.lcrc_set_kernel_done_callback.argtypes = [C.c_void_p, FN, C.c_void_p]
    assert L.lcrc_set_kernel_done_callback(ctx.h, fn, C.c_void_p(7)) == 0
    mel = modelgen.synth_mel(5000, 15, seed=1)
    a = ctx.posteriors(mel)                                   # copy-back in pieces (2.7 MB)
    assert calls == [7]
    ctx.posteriors(mel[:40])                                  # small: the plain copy-back
    off = np.array([0, 100, 100, 900], np.int32)
    ctx.posteriors_batch(mel[:900], off)
    ctx.posteriors_staged(mel[:900], off)
    assert calls == [7, 7, 7, 7]
    ctx.posteriors_batch(mel[:0], np.array([0, 0], np.int32))  # nothing to launch: no call
    assert len(calls) == 4
    ctx.reset()
    ctx.push(mel[:31], needed=False)                          # history only: no launch
    assert len(calls) == 4
    ctx.push(mel[31:36])
    assert len(calls) == 5
    assert L.lcrc_set_kernel_done_callback(ctx.h, FN(0), None) == 0
    b = ctx.posteriors(mel)
    assert len(calls) == 5 and np.array_equal(a, b)
    ctx.close()


def test_launch_order_across_the_contexts_of_a_device(capi):
    """lcrc_set_launch_order: three contexts of one device (a model and two clones, as the CLI's three per GPU), each on its
    own thread, fifty staged calls each of different sizes at once -- with the order on, every call's posterior kernels wait
    on the device for those of the call queued before them; the results are those of the same calls without the order, bit
    for bit, nothing deadlocks, and a context may switch the order off and on between calls."""
    import threading
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    a = capi.Lcrc(model_dir(system), 15)
    ctxs = [a, a.clone(), a.clone()]
    for c in ctxs:
        c.set_hidden_split(1)
    mels = [modelgen.synth_mel(n, 15, seed=40 + k, mean_norm=True) for k, n in enumerate((700, 5000, 33, 12000))]
    offs = [np.array([0, m.shape[0]], np.int32) for m in mels]
    want = [a.posteriors_staged(m, o) for m, o in zip(mels, offs)]

    def work(c, k0, out):
        for i in range(50):
            k = (k0 + i) % len(mels)
            if i == 20:
                c.set_launch_order(False)
            if i == 25:
                c.set_launch_order(True)
            out.append(np.array_equal(c.posteriors_staged(mels[k], offs[k]).view(np.uint32), want[k].view(np.uint32)))

    for c in ctxs:
        c.set_launch_order(True)
    res = [[] for _ in ctxs]
    th = [threading.Thread(target=work, args=(c, k, res[k])) for k, c in enumerate(ctxs)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a context's calls did not come back"
    assert all(len(r) == 50 and all(r) for r in res)
    for c in ctxs[1:]:
        c.close()
    a.close()


def test_decoder_on_the_device(capi, oracle_mod, tmp_path):
    """lcrc_decoder_configure ("next" row f3): the PhnDec kernel behind the posterior kernel against the
    decoder oracle run on the SAME log-posteriors (bit-identical labels, times and scores: both do the same
    f32 additions), on ragged batches incl. empty, 1-frame and shorter-than-the-pruning-horizon utterances,
    for several state counts / horizons / penalties; with and without posterior read-back"""
    for nb, hid, nout, P, S, prune, wpen in ((15, 64, 138, 45, 3, 40, -4.6875), (11, 40, 48, 12, 4, 7, -1.5),
                                             (9, 30, 20, 20, 1, 63, 0.0), (13, 30, 64, 21, 3, 1, -0.25),
                                             (15, 50, 192, 64, 3, 40, -2.0)):
        d = str(tmp_path / ("m%d_%d" % (P, S)))
        modelgen.write_model_dir(d, nb, hid, nout, seed=P)
        ctx = capi.Lcrc(d, nb)
        ctx.configure_output(("log",))
        lens = [0, 1, 2, prune, prune + 1, 3 * prune + 5, 400, 0, 37]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        mel = modelgen.synth_mel(int(off[-1]), nb, seed=S)
        logpost = ctx.posteriors_batch(mel, off)
        assert ctx.last_labels() == []                          # decoder not configured yet
        ctx.configure_decoder(P, S, prune, wpen)
        again = ctx.posteriors_batch(mel, off)
        assert np.array_equal(again, logpost)
        got = ctx.last_labels()
        assert len(got) == len(lens)
        for u in range(len(lens)):
            want = oracle_mod.phndec(logpost[off[u]:off[u + 1]], P, S, prune, wpen)
            assert got[u] == want, (P, S, prune, u, got[u][:3], want[:3])
            assert all(0 <= a[0] < a[1] <= lens[u] for a in want)                    # within the utterance,
            assert all(a[1] <= b[0] for a, b in zip(want, want[1:]))                  # in time order
        # staged entry without posterior read-back: labels only
        ctx.set_posterior_readback(False)
        ctx.posteriors_staged(mel, off)
        assert ctx.last_labels() == got
        # single-utterance form
        u = mel[off[5]:off[6]]
        ctx.set_posterior_readback(True)
        lp = ctx.posteriors(u)
        assert ctx.last_labels() == [oracle_mod.phndec(lp, P, S, prune, wpen)]
        # more utterances than one workgroup of the decoder kernel holds (sixteen, one per wave): three workgroups, the
        # last one partly empty, lengths all over the place
        rng = np.random.default_rng(P)
        lens2 = [int(x) for x in rng.integers(0, 3 * prune + 40, 37)]
        off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.int32)
        mel2 = modelgen.synth_mel(int(off2[-1]), nb, seed=S + 7)
        lp2 = ctx.posteriors_batch(mel2, off2)
        got2 = ctx.last_labels()
        assert len(got2) == len(lens2)
        for k in range(len(lens2)):
            assert got2[k] == oracle_mod.phndec(lp2[off2[k]:off2[k + 1]], P, S, prune, wpen), (P, S, prune, k, lens2[k])
        ctx.configure_decoder(0)
        ctx.posteriors(u)
        assert ctx.last_labels() == []
        with pytest.raises(capi.LcrcError):
            ctx.configure_decoder(65, 3, 40, 0.0)
        with pytest.raises(capi.LcrcError):
            ctx.configure_decoder(P, S, 64, 0.0)                # the winner history lives one slot per lane
        with pytest.raises(capi.LcrcError):
            ctx.configure_decoder(nout, 3, 40, 0.0)
        ctx.close()


# ---- split-f16 arithmetic (lcrc_set_arithmetic): f32 products as three exact f16 MFMA products ----------------------

@pytest.mark.parametrize("system", list(modelgen.SYSTEMS))
def test_split_f16_real_system_vs_reference_golden(capi, oracle_mod, system):
    """bundled test.raw in the split-f16 arithmetic: same bar against the reference CLI's own -t post dump, and as
    close to it as the f32 kernels are (the split drops less than f32's own rounding of the sums)"""
    spec = modelgen.SYSTEMS[system]
    mel = _norm_mel(oracle_mod, system)
    lop = read_htk(os.path.join(GOLD, system, "test.lop"))
    g = capi.Lcrc(model_dir(system), spec["nbanks"])
    f32 = g.posteriors(mel)
    g.set_arithmetic(capi.ARITH_SPLIT_F16)
    post = g.posteriors(mel)
    assert np.abs(post - lop).max() < TOL
    assert np.abs(post - lop).max() < 2 * np.abs(f32 - lop).max() + 2e-6
    assert np.abs(post - f32).max() < 2e-5
    assert np.abs(post.sum(axis=1) - 1).max() < 1e-5
    assert (post.argmax(axis=1) != lop.argmax(axis=1)).mean() < 0.005
    g.set_arithmetic(capi.ARITH_F32)
    assert np.array_equal(g.posteriors(mel), f32)


@pytest.mark.parametrize("shape", [(15, 100, 138), (15, 1500, 137), (15, 1390, 159), (15, 48, 186), (23, 500, 120), (23, 17, 119)])
def test_split_f16_vs_oracle_shapes_and_sizes(capi, oracle_mod, tmp_path, shape):
    """seeded models of the shipped shape classes (odd and even numbers of hidden tiles, fewer outputs than the class
    holds), utterances from 1 frame to several 32-frame tiles, as one utterance and as a batch"""
    nb, hid, no = shape
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, nb, hid, no, seed=hid)
    o = oracle_mod.Oracle(d, nb)
    g = capi.Lcrc(d, nb)
    assert not g.kernel_name.startswith("generic")
    g.set_arithmetic(capi.ARITH_SPLIT_F16)
    lens = [1, 2, 15, 16, 17, 31, 33, 64, 131]
    mel = modelgen.synth_mel(sum(lens), nb, seed=3)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = o.posteriors_batch(mel, off)
    got = g.posteriors_batch(mel, off)
    assert np.abs(got - want).max() < 2e-5
    for u in (0, 3, 8):                                  # an utterance alone = the same bits as inside the batch
        a, b = off[u], off[u + 1]
        assert np.array_equal(g.posteriors(mel[a:b]), got[a:b])
    for frames in (16, 32):
        g.set_tile_frames(frames)
        assert np.array_equal(g.posteriors_batch(mel, off), got)


def test_split_f16_streaming_rows_and_large_launch(capi, oracle_mod):
    """push, row ranges and a BASELINE-size launch in the split-f16 arithmetic: every form runs the fused kernel, so
    a frame's bits do not depend on how it is batched"""
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    g = capi.Lcrc(model_dir(system), 15)
    g.set_arithmetic(capi.ARITH_SPLIT_F16)
    mel = modelgen.synth_mel(8192, 15, seed=5)
    whole = g.posteriors(mel)
    assert np.abs(whole.sum(axis=1) - 1).max() < 1e-5
    o = oracle_mod.Oracle(model_dir(system), 15)
    want = o.posteriors(mel[:600], threads=8)
    assert np.abs(whole[:585] - want[:585]).max() < 2e-5          # (rows whose right context lies inside the cut)
    assert np.array_equal(g.posteriors_rows(mel, 4000, 100), whole[4000:4100])
    # the ProcessOffline sequence (srec.cpp:1035-1059) through the streaming entry points, in bunches of 5
    n = 300
    short = g.posteriors(mel[:n])
    g.reset()
    assert g.push(mel[:15], needed=False) is None
    out = [g.push(mel[i:i + 5]) for i in range(15, n, 5)]
    out.append(g.push(np.repeat(mel[n - 1:n], 15, axis=0)))
    assert np.array_equal(np.concatenate(out), short)


def test_split_f16_refused_where_it_does_not_exist(capi, tmp_path):
    """run-time-shape models have no split-f16 kernels: LCRC_E_UNSUPPORTED, the context stays on the f32 kernels.  Weights of
    any finite size DO have the form (each matrix is scaled by a power of two before it is split); normalised inputs beyond
    +-1023 are clamped"""
    d = str(tmp_path / "gen")
    modelgen.write_model_dir(d, 13, 64, 40, seed=2)
    g = capi.Lcrc(d, 13)
    assert g.kernel_name.startswith("generic")
    mel = modelgen.synth_mel(50, 13, seed=1)
    before = g.posteriors(mel)
    with pytest.raises(capi.LcrcError) as e:
        g.set_arithmetic(capi.ARITH_SPLIT_F16)
    assert e.value.code == capi.LCRC_E_UNSUPPORTED
    assert np.array_equal(g.posteriors(mel), before)
    with pytest.raises(capi.LcrcError):
        g.set_arithmetic(7)
    # a shipped shape with one weight of 1e5 (beyond f16's range as it stands): scaled into range, same bar
    d2 = str(tmp_path / "big")
    modelgen.write_model_dir(d2, 15, 64, 138, seed=3)
    from oracle import binding as ob
    net = ob.Net(nbin=os.path.join(d2, "weights", "band0.nbin"))
    w = np.ctypeslib.as_array(net.n.W1, shape=(net.n.nHid16, net.n.nInp16))
    w[3, 5] = 1.0e5
    # (its input is made tiny so that the product stays inside FEXP's sane range: beyond |x| ~ 700 the reference's own
    #  sigmoid is garbage that no other arithmetic reproduces)
    np.ctypeslib.as_array(net.n.dev, shape=(net.n.nInp16,))[5] = 1.0e-8
    net.save_nbin(os.path.join(d2, "weights", "band0.nbin"))
    g2 = capi.Lcrc(d2, 15)
    g2.set_arithmetic(capi.ARITH_SPLIT_F16)
    mel2 = modelgen.synth_mel(80, 15, seed=12)
    assert np.abs(g2.posteriors(mel2) - ob.Oracle(d2, 15).posteriors(mel2)).max() < TOL
    # a non-finite weight has no such form
    d4 = str(tmp_path / "nan")
    modelgen.write_model_dir(d4, 15, 64, 138, seed=3)
    net = ob.Net(nbin=os.path.join(d4, "weights", "merger.nbin"))
    w = np.ctypeslib.as_array(net.n.W2, shape=(net.n.nOut16, net.n.nHid16))
    w[1, 2] = np.inf
    net.save_nbin(os.path.join(d4, "weights", "merger.nbin"))
    g4 = capi.Lcrc(d4, 15)
    with pytest.raises(capi.LcrcError) as e:
        g4.set_arithmetic(capi.ARITH_SPLIT_F16)
    assert e.value.code == capi.LCRC_E_UNSUPPORTED
    # large inputs: FEXP's tails as in the f32 kernels; beyond +-1023 (no audio gets there) the normalised input is
    # clamped -- finite, normalised posteriors, a documented deviation
    d3 = str(tmp_path / "ok")
    modelgen.write_model_dir(d3, 15, 64, 138, seed=3)
    o3 = ob.Oracle(d3, 15)
    g3 = capi.Lcrc(d3, 15)
    g3.set_arithmetic(capi.ARITH_SPLIT_F16)
    mel = modelgen.synth_mel(64, 15, seed=9) * np.float32(25.0)
    mel[10] = 300.0
    mel[11] = -300.0
    assert np.abs(g3.posteriors(mel) - o3.posteriors(mel)).max() < TOL
    mel[7] = 3.0e6
    mel[9] = -3.0e6
    got = g3.posteriors(mel)
    assert np.isfinite(got).all() and np.abs(got.sum(axis=1) - 1).max() < 1e-5
    assert np.abs(got[40:] - o3.posteriors(mel)[40:]).max() < TOL          # rows whose context holds no such frame


@pytest.mark.parametrize("scale", [1.0e-3, 4.0])
def test_split_f16_keeps_its_precision_for_small_and_large_weights(capi, tmp_path, scale):
    """The (high, low) f16 split holds 22 bits only while the low half is a normal f16; the packer therefore scales every
    weight matrix by a power of two first.  A model whose weights are 1e-3 (or 4) times the usual size must come out
    as close to the oracle in the split-f16 arithmetic as on the f32 kernels (unscaled, the 1e-3 model would lose ten
    bits of every weight: an error floor ~1e-6 in the pre-activations, against ~1e-8 in f32)."""
    from oracle import binding as ob
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 96, 138, seed=21)
    for name in ("band0", "band1", "merger"):
        f = os.path.join(d, "weights", name + ".nbin")
        net = ob.Net(nbin=f)
        for arr, shape in ((net.n.W1, (net.n.nHid16, net.n.nInp16)), (net.n.W2, (net.n.nOut16, net.n.nHid16))):
            np.ctypeslib.as_array(arr, shape=shape)[:] *= np.float32(scale)
        net.save_nbin(f)
    mel = modelgen.synth_mel(200, 15, seed=5)
    want = ob.Oracle(d, 15).posteriors(mel)
    g = capi.Lcrc(d, 15)
    err32 = float(np.abs(g.posteriors(mel) - want).max())
    g.set_arithmetic(capi.ARITH_SPLIT_F16)
    err16 = float(np.abs(g.posteriors(mel) - want).max())
    assert err16 < TOL and err16 <= 4.0 * err32 + 2e-7, (err16, err32)
    g.close()


def test_large_calls_in_two_launches_equal_their_utterances_alone(capi):
    """lcrc_posteriors_batch from 8192 rows on computes the rows in two launches that store straight into the pinned buffer
    (two_part_output, lcrc_api.cpp): a ragged batch of 10 001 rows in seven utterances -- the cut between the launches falls
    inside an utterance -- equals every utterance computed alone, bit for bit, and the one-utterance form equals itself
    computed as the front part of a longer call"""
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    nb = modelgen.SYSTEMS[system]["nbanks"]
    ctx = capi.Lcrc(model_dir(system), nb)
    ctx.set_hidden_split(1)
    lens = [1, 977, 3000, 16, 2999, 2900, 108]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    assert off[-1] == 10001
    mel = modelgen.synth_mel(int(off[-1]), nb, seed=91)
    got = ctx.posteriors_batch(mel, off)
    for u in range(len(lens)):
        alone = ctx.posteriors(mel[off[u]:off[u + 1]])
        assert np.array_equal(got[off[u]:off[u + 1]].view(np.uint32), alone.view(np.uint32)), u
    whole = ctx.posteriors(mel)                      # one utterance of 10 001 rows: two launches too
    assert np.array_equal(whole[:4000].view(np.uint32), ctx.posteriors(mel[:4100])[:4000].view(np.uint32))
