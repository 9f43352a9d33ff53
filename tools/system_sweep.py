#!/usr/bin/env python3
"""Kernel time and f32-MFMA roofline fraction of the fused posterior kernel for the four shipped LCRC
shapes (real weights where the model directory is committed, seeded synthetic weights of the same shape
otherwise) at BASELINE batch sizes.  Needs a GPU.  usage: system_sweep.py [frames ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from phnrec_amd import capi, modelgen  # noqa: E402

PEAK = 157.3e12


def flop_per_frame(spec):
    k1, h, o = spec["nbanks"] * 11, spec["hidden"], spec["n_out"]
    return 2 * (2 * (k1 * h + h * o) + (2 * o * h + h * o))


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 32768]
    capi.load()
    print("%-26s %-12s %7s %9s %9s %7s" % ("system", "kernel", "frames", "ms", "Mframe/s", "frac"))
    for system, spec in modelgen.SYSTEMS.items():
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        if not os.path.isdir(mdir):
            mdir = "/tmp/sweep_model_" + system
            modelgen.write_system(mdir, system, seed=1)
        ctx = capi.Lcrc(mdir, spec["nbanks"])
        if os.environ.get("SWEEP_NO_SPLIT"):        # as the CLI runs: fused kernels only (batch-invariant bits)
            ctx.set_hidden_split(1)
        if os.environ.get("SWEEP_SPLIT"):           # forced workgroups per tile (tuning)
            ctx.set_hidden_split(int(os.environ["SWEEP_SPLIT"]))
        if os.environ.get("SWEEP_SYSTEMS") and system.split("_")[1] not in os.environ["SWEEP_SYSTEMS"].split(","):
            ctx.close()
            continue
        for n in sizes:
            mel = torch.from_numpy(modelgen.synth_mel(n, spec["nbanks"], seed=1)).cuda()
            post = torch.empty((n, ctx.n_out), device="cuda")
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.set_timing(False)
            for _ in range(300):          # (~30 ms of load: the clock is up)
                ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            reps = 100
            e0.record(s)
            for _ in range(reps):
                ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            e1.record(s)
            s.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("%-26s %-12s %7d %9.4f %9.2f %7.3f" % (system, ctx.kernel_name, n, ms, n / ms / 1e3,
                                                       n * flop_per_frame(spec) / (ms * 1e-3) / PEAK), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
