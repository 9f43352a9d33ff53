#!/usr/bin/env python3
"""A/B of two builds of the library on small launches, same GPU, same process, interleaved: kernel time per launch at
5 ... 2048 frames (device buffers) and the streaming entry lcrc_push at a bunch of 5 (host buffers).
usage: ab_small.py LIB_A LIB_B"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from phnrec_amd import capi, modelgen  # noqa: E402


def ctx_of(lib, mdir, nb):
    vp = C.c_void_p
    h = vp()
    lib.lcrc_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.lcrc_posteriors_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.lcrc_push.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    lib.lcrc_reset.argtypes = [vp]
    lib.lcrc_num_outputs.argtypes = [vp]
    lib.lcrc_set_timing.argtypes = [vp, C.c_int]
    assert lib.lcrc_create(C.byref(h), mdir.encode(), nb, 31, 1, 0) == 0
    lib.lcrc_set_timing(h, 0)
    return h


def main():
    capi._load_hip_runtime()
    libs = [C.CDLL(os.path.join(ROOT, p)) for p in sys.argv[1:3]]
    s = torch.cuda.current_stream()
    for system in ("PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"):
        spec = modelgen.SYSTEMS[system]
        nb = spec["nbanks"]
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        hs = [ctx_of(L, mdir, nb) for L in libs]
        n_out = libs[0].lcrc_num_outputs(hs[0])
        mel = torch.from_numpy(modelgen.synth_mel(2048, nb, seed=1)).cuda()
        posts = [torch.empty((2048, n_out), device="cuda") for _ in libs]
        print(system)
        for n in (5, 64, 256, 512, 1024, 1536, 2048):
            t = [[], []]
            for rnd in range(7):
                for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for _ in range(100):
                        libs[k].lcrc_posteriors_device(hs[k], mel.data_ptr(), None, 1, n, posts[k].data_ptr(), s.cuda_stream)
                    e0.record(s)
                    for _ in range(200):
                        libs[k].lcrc_posteriors_device(hs[k], mel.data_ptr(), None, 1, n, posts[k].data_ptr(), s.cuda_stream)
                    e1.record(s)
                    s.synchronize()
                    t[k].append(e0.elapsed_time(e1) / 200)
            a, b = np.median(t[0][1:]), np.median(t[1][1:])
            same = bool(torch.equal(posts[0][:n], posts[1][:n]))
            diff = float((posts[0][:n] - posts[1][:n]).abs().max().item())
            print("  %5d frames: A %.4f ms  B %.4f ms  B/A %.3f  identical: %s (max |A-B| %.2g)" % (n, a, b, b / a, same, diff))
        hmel = modelgen.synth_mel(4000, nb, seed=2)
        out = np.empty((5, n_out), np.float32)
        t = [[], []]
        for rnd in range(7):
            for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                libs[k].lcrc_reset(hs[k])
                for i in range(0, 500, 5):
                    libs[k].lcrc_push(hs[k], hmel[i:i + 5].ctypes.data, 5, out.ctypes.data, 1)
                t0 = time.perf_counter()
                for i in range(500, 3500, 5):
                    libs[k].lcrc_push(hs[k], hmel[i:i + 5].ctypes.data, 5, out.ctypes.data, 1)
                t[k].append((time.perf_counter() - t0) / 600)
        a, b = np.median(t[0][1:]), np.median(t[1][1:])
        print("  push bunch 5: A %.1f us  B %.1f us per call  B/A %.3f" % (a * 1e6, b * 1e6, b / a))
        # the host-pointer entry (lcrc_posteriors: one utterance in host memory, posteriors back in host memory)
        for L in libs:
            L.lcrc_posteriors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        for n in (50, 300, 2000, 8192):
            hm = np.ascontiguousarray(modelgen.synth_mel(n, nb, seed=3))
            outs = [np.empty((n, n_out), np.float32) for _ in libs]
            t = [[], []]
            reps = 300 if n <= 300 else 60
            for rnd in range(7):
                for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                    for _ in range(20):
                        libs[k].lcrc_posteriors(hs[k], hm.ctypes.data, n, outs[k].ctypes.data)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        libs[k].lcrc_posteriors(hs[k], hm.ctypes.data, n, outs[k].ctypes.data)
                    t[k].append((time.perf_counter() - t0) / reps)
            a, b = np.median(t[0][1:]), np.median(t[1][1:])
            print("  lcrc_posteriors %5d frames (host buffers): A %.1f us  B %.1f us per call  B/A %.3f  identical: %s"
                  % (n, a * 1e6, b * 1e6, b / a, bool(np.array_equal(outs[0], outs[1]))))


if __name__ == "__main__":
    main()
