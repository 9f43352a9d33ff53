#!/usr/bin/env python3
"""Timing of the non-LCRC `posteriors/system` variants (feature kernel + MLP kernels) on seeded synthetic
models, 8192 frames resident in HBM.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from phnrec_amd import capi, modelgen  # noqa: E402


PREHEAT, REPS = 400, 200


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    print("pre-heat %d launches, %d timed launches per line" % (PREHEAT, REPS))
    capi.load()
    # (system, model options, separate launches, posteriors/length); the last line: LCRC with CZ's dimensions at a geometry
    # no shipped model has (length 21, no C0, 10 coefficients per band) -- the general kernels, three launches
    for system, kw, unfused, length in (("1BT_DCT", dict(coefs=11), False, 31), ("1BT_DCT", dict(coefs=11), True, 31),
                                        ("1BT", dict(band_out=24, band_hidden=100), False, 31),
                                        ("LCRC", dict(coefs=10, add_c0=False), True, 21)):
        with tempfile.TemporaryDirectory() as d:
            os.environ.pop("PHNREC_TRAPS_UNFUSED", None)
            if unfused:
                os.environ["PHNREC_TRAPS_UNFUSED"] = "1"   # round 1's form: features kernel + MLP kernel
            if system == "LCRC":
                modelgen.write_model_dir(d, 15, 1500, 138, seed=5, trap_len=length, **kw)
            else:
                modelgen.write_traps_dir(d, system, 15, 1500, 138, seed=5, **kw)
            ctx = capi.Lcrc(d, 15, system=system, trap_len=length, add_c0=kw.get("add_c0", True))
            mel = torch.from_numpy(modelgen.synth_mel(n, 15, seed=1)).cuda()
            post = torch.empty((n, ctx.n_out), device="cuda")
            s = torch.cuda.current_stream()
            ctx.set_timing(False)
            # disclosed pre-heat, as bench.py's: after idling the device needs ~25 ms of load to reach its steady clock
            # (round 2's figures of this tool were taken over launches 6-25 of the process: 30-50 % slower)
            for _ in range(PREHEAT):
                ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(REPS):
                ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            e1.record(s)
            s.synchronize()
            ms = e0.elapsed_time(e1) / REPS
            dims = [ctx.net_dims(i) for i in range((0 if system == "1BT_DCT" else 2 if system == "LCRC" else 15) + 1)]
            flop = sum(2 * (a * b + b * c) for a, b, c in dims)
            print("%-8s%s %d frames: %.3f ms per batch = %.2f M frames/s; nets %s ...; %.1f TFLOP/s algorithmic = %.0f %% of f32 MFMA peak"
                  % (system, " length %d (general kernels)" % length if system == "LCRC" else " (features + MLP launches)" if unfused else " (one launch)" if system == "1BT_DCT" else "", n, ms, n / ms / 1e3, dims[-1], n * flop / (ms * 1e-3) / 1e12, 100 * n * flop / (ms * 1e-3) / 157.3e12))
            ctx.close()


if __name__ == "__main__":
    main()
