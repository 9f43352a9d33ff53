#!/usr/bin/env python3
"""Randomised parity run (GPU): random LCRC model shapes (1-23 banks, hidden 1-399, 2-208 outputs, independent merger
hidden size) on ragged batches, 16- and 32-frame workgroups and forced hidden splits, each against the oracle at the
1e-4 bar; then models of the shipped shape classes (random hidden sizes, weight scales and input scales) in the split-f16
arithmetic, against the oracle and against the f32 kernels.
usage: fuzz_parity.py [seed [models [split_f16_models [geometry_models]]]]      (also tests/test_gpu_parity.py::test_fuzzed_models)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fuzz(seed=0, n_models=60, n_h2=24, log=print):
    from oracle import binding as ob
    from phnrec_amd import capi, modelgen
    capi.load()
    rng = np.random.default_rng(seed)
    worst, ran = 0.0, 0
    for it in range(n_models):
        nb = int(rng.integers(1, 24))
        hid = int(rng.integers(1, 400))
        nout = int(rng.integers(2, 209))
        hm = int(rng.integers(1, 400))
        with tempfile.TemporaryDirectory() as d:
            modelgen.write_model_dir(d, nb, hid, nout, seed=int(rng.integers(1 << 30)), hidden_merger=hm)
            try:
                ctx = capi.Lcrc(d, nb)
            except capi.LcrcError as e:
                if e.code != capi.LCRC_E_UNSUPPORTED:
                    raise
                log("skip", nb, hid, nout, hm, str(e)[:60])
                continue
            o = ob.Oracle(d, nb)
            lens = [int(v) for v in rng.integers(0, 120, size=int(rng.integers(1, 7)))]
            if sum(lens) == 0:
                lens.append(5)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            mel = modelgen.synth_mel(int(off[-1]), nb, seed=it, mean_norm=bool(it & 1))
            want = o.posteriors_batch(mel, off)
            for fr, split in ((16, 1), (32, 1), (0, 2), (0, 5), (0, 64)):     # fused kernels, then forced hidden splits
                ctx.set_tile_frames(fr)
                ctx.set_hidden_split(split)
                got = ctx.posteriors_batch(mel, off)
                err = float(np.abs(got - want).max())
                worst = max(worst, err)
                assert err < 1e-4, (nb, hid, nout, hm, fr, split, err)
            ctx.close()
            ran += 1
    log("fuzz ok: %d models, worst max-abs %g" % (ran, worst))
    worst_s, worst_d = 0.0, 0.0
    for it in range(n_h2):
        nb, nout = [(15, 138), (15, 137), (15, 186), (15, 185), (15, 159), (15, 160), (23, 120), (23, 119)][it % 8]
        hid, hm = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        with tempfile.TemporaryDirectory() as d:
            modelgen.write_model_dir(d, nb, hid, nout, seed=int(rng.integers(1 << 30)), hidden_merger=hm)
            ctx = capi.Lcrc(d, nb)
            assert not ctx.kernel_name.startswith("generic"), ctx.kernel_name
            o = ob.Oracle(d, nb)
            lens = [int(v) for v in rng.integers(0, 150, size=int(rng.integers(1, 7)))]
            if sum(lens) == 0:
                lens.append(5)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            mel = modelgen.synth_mel(int(off[-1]), nb, seed=it, mean_norm=bool(it & 1)) * np.float32(rng.choice([0.2, 1.0, 4.0]))
            want = o.posteriors_batch(mel, off)
            f32 = ctx.posteriors_batch(mel, off)
            ctx.set_arithmetic(capi.ARITH_SPLIT_F16)
            for fr in (16, 32):
                ctx.set_tile_frames(fr)
                got = ctx.posteriors_batch(mel, off)
                err, dif = float(np.abs(got - want).max()), float(np.abs(got - f32).max())
                worst_s, worst_d = max(worst_s, err), max(worst_d, dif)
                assert err < 1e-4 and dif < 5e-5, (nb, hid, nout, hm, fr, err, dif)
            ctx.close()
    log("split-f16 fuzz ok: %d models, worst max-abs vs oracle %g, vs the f32 kernels %g" % (n_h2, worst_s, worst_d))
    return ran, worst, worst_s, worst_d


def fuzz_geometry(seed=0, n_models=40, log=print):
    """Random geometries of the general kernels: any posteriors/length 2..64 (odd and even), add_c0 on / off, 1..16 values
    per band, all four systems, Hamming window on / off -- against the run-time-geometry oracle."""
    from oracle import binding as ob
    from phnrec_amd import capi, modelgen
    capi.load()
    rng = np.random.default_rng(seed)
    worst = 0.0
    for it in range(n_models):
        system = ["LCRC", "LCRC", "1BT_DCT", "1BT", "3BT"][int(rng.integers(0, 5))]
        L = int(rng.integers(2, 65))
        nb = int(rng.integers(3, 24))
        c0 = bool(rng.integers(0, 2))
        hamm = bool(rng.integers(0, 2)) and system != "LCRC"
        coefs = int(rng.integers(1, 17))
        if system == "1BT_DCT":
            coefs = min(coefs, L + (1 if c0 else 0))          # at most L cosine rows (+ C0)
        hid, nout = int(rng.integers(1, 200)), int(rng.integers(2, 100))
        with tempfile.TemporaryDirectory() as d:
            if system == "LCRC":
                if L == 31 and c0 and coefs == 11:
                    coefs = 10
                modelgen.write_model_dir(d, nb, hid, nout, seed=int(rng.integers(1 << 30)), coefs=coefs, trap_len=L, add_c0=c0)
            else:
                modelgen.write_traps_dir(d, system, nb, hid, nout, seed=int(rng.integers(1 << 30)), coefs=coefs, add_c0=c0,
                                         hamming=hamm, trap_len=L, band_out=int(rng.integers(2, 30)), band_hidden=int(rng.integers(1, 60)))
            ctx = capi.Lcrc(d, nb, system=system, add_c0=c0, hamming=hamm, trap_len=L)
            o = ob.TrapsOracle(d, system, nb, c0, hamm, trap_len=L)
            lens = [int(v) for v in rng.integers(0, 90, size=int(rng.integers(1, 6)))]
            if sum(lens) == 0:
                lens.append(3)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            mel = modelgen.synth_mel(int(off[-1]), nb, seed=it)
            got = ctx.posteriors_batch(mel, off)
            err = float(np.abs(got - o.posteriors_batch(mel, off)).max())
            worst = max(worst, err)
            assert err < 1e-4, (system, L, nb, c0, hamm, coefs, err)
            assert np.array_equal(ctx.posteriors_staged(mel, off), got), (system, L)
            # the streaming form: frame r's estimate appears when frame r + (L - 1) / 2 has been pushed
            u = mel[int(off[-2]):int(off[-1])] if off[-1] > off[-2] else mel[:int(off[1])]
            if len(u):
                shift = (L - 1) // 2
                whole = ctx.posteriors(u)
                ctx.reset()
                parts = [ctx.push(u[i:i + 5]) for i in range(0, len(u), 5)]
                if shift:
                    parts.append(ctx.push(np.repeat(u[-1:], shift, axis=0)))
                pushed = np.concatenate(parts)[shift:]
                assert np.abs(pushed - whole).max() < 1e-6, (system, L, nb, len(u), float(np.abs(pushed - whole).max()))
            ctx.close()
    log("geometry fuzz ok: %d models, worst max-abs %g" % (n_models, worst))
    return worst


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    fuzz(*a[:3])
    fuzz_geometry(a[0] if a else 0, a[3] if len(a) > 3 else 40)
