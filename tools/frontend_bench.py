#!/usr/bin/env python3
"""Waveform entry (GPU front-end + posteriors) on the BASELINE configs[2] input: one 81.9 s utterance of
synthetic 8 kHz A-law audio = 8192 frames.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnrec_amd import capi, modelgen  # noqa: E402


def alaw_encode(x):
    """nearest-value search in the 256-entry expansion table"""
    table = np.array([capi_alaw(b) for b in range(256)], np.float32)
    order = np.argsort(table)
    idx = np.searchsorted(table[order], x)
    idx = np.clip(idx, 1, 255)
    lo, hi = table[order][idx - 1], table[order][idx]
    pick = np.where(np.abs(x - lo) <= np.abs(hi - x), idx - 1, idx)
    return order[pick].astype(np.uint8)


def capi_alaw(b):
    a = b ^ 0x55
    mant = (a & 0x0F) << 4
    seg = (a & 0x70) >> 4
    mant = mant + 8 if seg == 0 else (mant + 0x108) << (seg - 1)
    return float(mant if a & 0x80 else -mant)


def main():
    system = "PHN_CZ_SPDAT_LCRC_N1500"
    spec = modelgen.SYSTEMS[system]
    frames = 8192
    n = (frames - 1) * 80 + 200
    rng = np.random.default_rng(1235)
    t = np.arange(n) / 8000.0
    sig = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, n)
    raw = alaw_encode(np.clip(sig, -32768, 32767).astype(np.float32)).tobytes()
    ctx = capi.Lcrc(os.path.join(ROOT, "tests", "golden", "models", system), spec["nbanks"])
    ctx.configure_frontend(wave_format="alaw", sent_mean_norm=True)
    assert ctx.frontend_frames(len(raw)) == frames
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx.wave_to_posteriors([raw])
    t0 = time.perf_counter()
    for _ in range(reps):
        post, foff = ctx.wave_to_posteriors([raw])
    dt = (time.perf_counter() - t0) / reps
    print("waveform -> posteriors, %d frames (%d bytes A-law): %.3f ms per call incl. H2D/D2H = %.2f M frames/s; "
          "rows sum to 1: %s" % (frames, len(raw), dt * 1e3, frames / dt / 1e6, bool(np.abs(post.sum(1) - 1).max() < 1e-5)))
    ctx.close()
    # the FFT-512 form of melbank_kernel (EN: 16 kHz lin16, 400-sample frames; 22 KiB of LDS per workgroup, seven per CU)
    import bench
    system = "PHN_EN_TIMIT_LCRC_N500"
    raw = bench.config1_lin16_signal(8192)
    ctx = capi.Lcrc(os.path.join(ROOT, "tests", "golden", "models", system), modelgen.SYSTEMS[system]["nbanks"])
    ctx.configure_frontend(wave_format="lin16", sent_mean_norm=False, sample_freq=16000, vector_size=400, vector_step=160,
                           lower_freq=0.0, higher_freq=8000.0)
    assert ctx.frontend_frames(len(raw)) == 8192
    ctx.wave_to_posteriors([raw])
    t0 = time.perf_counter()
    for _ in range(reps):
        post, foff = ctx.wave_to_posteriors([raw])
    dt = (time.perf_counter() - t0) / reps
    print("EN (FFT 512) waveform -> posteriors, 8192 frames (%d bytes lin16): %.3f ms per call incl. H2D/D2H = %.2f M frames/s; "
          "rows sum to 1: %s" % (len(raw), dt * 1e3, 8192 / dt / 1e6, bool(np.abs(post.sum(1) - 1).max() < 1e-5)))


if __name__ == "__main__":
    main()
