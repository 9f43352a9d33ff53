#!/usr/bin/env python3
"""Small-launch regime of the posterior path: kernel time per launch for launches of 5 .. 8192 frames with
the fused kernel (split 1) and the split-hidden path (automatic / forced workgroups per tile), and the
streaming entry (lcrc_reset / lcrc_push) at the shipped bunch of 5 and at 512.  Needs a GPU.
usage: small_launch_sweep.py [SYSTEM ...]   (default: CZ and EN)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from phnrec_amd import capi, modelgen  # noqa: E402

PEAK = 157.3e12


def flop_per_frame(spec):
    k1, h, o = spec["nbanks"] * 11, spec["hidden"], spec["n_out"]
    return 2 * (2 * (k1 * h + h * o) + (2 * o * h + h * o))


def time_launches(ctx, mel, post, n, reps=200):
    s = torch.cuda.current_stream()
    for _ in range(20):
        ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
    e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) / reps


def push_rate(ctx, nb, bunch, seconds=1.5):
    mel = modelgen.synth_mel(max(bunch, 5) * 64, nb, seed=2)
    ctx.reset()
    ctx.push(mel[:15], needed=False)
    for i in range(20):
        ctx.push(mel[i * bunch:(i + 1) * bunch])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for i in range(64):
            ctx.push(mel[i * bunch:(i + 1) * bunch])
        n += 64
    dt = time.perf_counter() - t0
    return n * bunch / dt, dt / n * 1e6


def main():
    systems = sys.argv[1:] or ["PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"]
    capi.load()
    for system in systems:
        spec = modelgen.SYSTEMS[system]
        nb = spec["nbanks"]
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        ctx = capi.Lcrc(mdir, nb)
        ctx.set_timing(False)
        fpf = flop_per_frame(spec)
        print("%s (%s)" % (system, ctx.kernel_name))
        print("%7s %6s %9s %10s %7s" % ("frames", "split", "ms", "Mframe/s", "frac"))
        big = torch.from_numpy(modelgen.synth_mel(8192, nb, seed=1)).cuda()
        post = torch.empty((8192, ctx.n_out), device="cuda")
        for n in (5, 16, 64, 256, 512, 1024, 2048, 3072, 4096, 8192):
            for split in (1, 0, 2, 3, 4, 6, 8, 12, 16, 24):
                tiles = (n + 15) // 16
                if split > 1 and (512 // tiles < split or split > 47):
                    continue
                if split == 0 and tiles > 128:
                    continue
                ctx.set_hidden_split(split)
                ms = time_launches(ctx, big, post, n)
                print("%7d %6s %9.4f %10.3f %7.3f" % (n, "auto" if split == 0 else split, ms, n / ms / 1e3,
                                                       n * fpf / (ms * 1e-3) / PEAK), flush=True)
        ctx.set_hidden_split(0)
        for bunch in (5, 512):
            for split in (0, 1):
                ctx.set_hidden_split(split)
                fps, us = push_rate(ctx, nb, bunch)
                print("push bunch %4d split %-4s: %10.0f frames/s  %8.1f us per call" % (bunch, "auto" if split == 0 else "1", fps, us), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
