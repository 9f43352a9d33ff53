#!/usr/bin/env python3
"""The split-f16 arithmetic (lcrc_set_arithmetic) beside the f32-MFMA kernels: distance of both to the oracle on the
four shipped systems (real weights), and kernel time per launch at BASELINE batch sizes.  Needs a GPU.
usage: split_f16_bench.py [frames ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import binding as ob  # noqa: E402   (dev tool: the oracle is the checker here)
from phnrec_amd import capi, modelgen  # noqa: E402

PEAK_F32, PEAK_F16 = 157.3e12, 2516.6e12


def flop_per_frame(spec):
    k1, h, o = spec["nbanks"] * 11, spec["hidden"], spec["n_out"]
    return 2 * (2 * (k1 * h + h * o) + (2 * o * h + h * o))


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 32768]
    capi.load()
    for system, spec in modelgen.SYSTEMS.items():
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        if not os.path.isdir(mdir):
            continue
        nb = spec["nbanks"]
        ctx = capi.Lcrc(mdir, nb)
        o = ob.Oracle(mdir, nb)
        mel = modelgen.synth_mel(1200, nb, seed=11)
        ref = o.posteriors(mel, threads=8)
        ctx.set_arithmetic(capi.ARITH_F32)
        a = ctx.posteriors(mel)
        ctx.set_arithmetic(capi.ARITH_SPLIT_F16)
        b = ctx.posteriors(mel)
        print("%s (%s), 1200 frames: max |f32 kernels - oracle| %.2e   max |split-f16 - oracle| %.2e   max |split - f32| %.2e"
              % (system, ctx.kernel_name, np.abs(a - ref).max(), np.abs(b - ref).max(), np.abs(a - b).max()), flush=True)
        s = torch.cuda.current_stream()
        ctx.set_timing(False)
        for n in sizes:
            d_mel = torch.from_numpy(modelgen.synth_mel(n, nb, seed=1)).cuda()
            post = torch.empty((n, ctx.n_out), device="cuda")
            ms = []
            for ar in (capi.ARITH_F32, capi.ARITH_SPLIT_F16):
                ctx.set_arithmetic(ar)
                for _ in range(200):
                    ctx.posteriors_device(d_mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(200):
                    ctx.posteriors_device(d_mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
                e1.record(s)
                s.synchronize()
                ms.append(e0.elapsed_time(e1) / 200)
            fl = n * flop_per_frame(spec)
            print("  %6d frames: f32 %.4f ms (%.3f of the f32 MFMA peak)   split-f16 %.4f ms = %.1f M frames/s, %.0f TFLOP/s "
                  "algorithmic = %.2f x the f32 MFMA peak, %.3f of the f16 MFMA peak at 3 products   x%.2f"
                  % (n, ms[0], fl / (ms[0] * 1e-3) / PEAK_F32, ms[1], n / ms[1] / 1e3, fl / (ms[1] * 1e-3) / 1e12,
                     fl / (ms[1] * 1e-3) / PEAK_F32, 3 * fl / (ms[1] * 1e-3) / PEAK_F16, ms[0] / ms[1]), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
