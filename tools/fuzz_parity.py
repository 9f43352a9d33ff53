#!/usr/bin/env python3
"""Randomised parity run (GPU): random LCRC model shapes (1-23 banks, hidden 1-399, 2-208 outputs, independent merger
hidden size) on ragged batches, 16- and 32-frame workgroups and forced hidden splits, each against the oracle at the
1e-4 bar; then models of the shipped shape classes (random hidden sizes, weight scales and input scales) in the split-f16
arithmetic, against the oracle and against the f32 kernels.
usage: fuzz_parity.py [seed [models [split_f16_models]]]      (also tests/test_gpu_parity.py::test_fuzzed_models)"""
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


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    fuzz(*a)
