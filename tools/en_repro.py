#!/usr/bin/env python3
"""Why did a fresh box read 0.0893 ms for EN at 8192 frames (BENCH_r04 `systems.en_8192`) where every other record reads
0.078-0.080?  Traces of per-window launch times of one system at one size behind different histories:

  idle      2 s of nothing, then 30 windows of 40 launches back to back (the clock ramp as this size sees it)
  push      1 s of lcrc_push(5) calls of a CZ context (the device idles between calls), then the same 30 windows
  heated    the windows again, straight behind the previous trace

usage: en_repro.py [SYSTEM] [frames]        (needs a GPU; run it under rocprofv3 --kernel-trace for per-launch durations)
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from phnrec_amd import capi, modelgen  # noqa: E402


def main():
    system = sys.argv[1] if len(sys.argv) > 1 else "PHN_EN_TIMIT_LCRC_N500"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    spec = modelgen.SYSTEMS[system]
    nb = spec["nbanks"]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    ctx = capi.Lcrc(os.path.join(ROOT, "tests", "golden", "models", system), nb, device=0)
    ctx.set_timing(False)
    fpf = bench.algorithmic_flops_per_frame([ctx.net_dims(i) for i in range(3)])
    d_mel = torch.from_numpy(modelgen.synth_mel(n, nb, seed=7, mean_norm=spec["sent_mean_norm"])).to(dev)
    d_post = torch.empty((n, ctx.n_out), dtype=torch.float32, device=dev)
    cz = capi.Lcrc(os.path.join(ROOT, "tests", "golden", "models", bench.SYSTEM), 15, device=0)
    push = cz.L.lcrc_push
    push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    mel5 = modelgen.synth_mel(5 * 64, 15, seed=11)
    post5 = np.empty((5, cz.n_out), np.float32)

    def trace(label):
        ws = bench.time_windows(ctx, stream, d_mel, d_post, n, windows=30, per=40)
        fr = [n * fpf / (t * 1e-3) / 1e12 / bench.PEAK_F32_MFMA_TFLOPS for t in ws]
        print("%-8s ms per launch by window: %s" % (label, " ".join("%.4f" % t for t in ws)))
        print("%-8s first %.4f (%.3f)  median %.4f (%.3f)  last %.4f (%.3f)  windows to within 1 %% of the median: %d"
              % ("", ws[0], fr[0], sorted(ws)[15], sorted(fr)[15], ws[-1], fr[-1],
                 next((i for i, t in enumerate(ws) if t <= sorted(ws)[15] * 1.01), 30)))

    print("%s, %d frames, kernel %s" % (system, n, ctx.kernel_name))
    for rnd in range(2):
        time.sleep(2.0)
        trace("idle")
        trace("heated")
        cz.reset()
        push(cz.h, mel5.ctypes.data, 15, None, 0)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for i in range(32):
                push(cz.h, mel5[i * 5:].ctypes.data, 5, post5.ctypes.data, 1)
        trace("push")
        trace("heated")
        # the bench's entry as it is now: time-based pre-heat, 7 windows
        time.sleep(1.0)
        print("entry    %s" % {k: v for k, v in bench.launch_entry(ctx, stream, d_mel, d_post, n, fpf).items()
                               if k.startswith("kernel_ms") or k.startswith("frac")})


if __name__ == "__main__":
    main()
