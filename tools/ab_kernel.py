#!/usr/bin/env python3
"""A/B timing of two builds of the library on the SAME GPU in the SAME process, interleaved (boxes differ by
a few percent, so numbers from separate runs cannot be compared at that level).

usage: ab_kernel.py LIB_A LIB_B [frames]      (paths relative to the repo root; needs a GPU)
LCRC_ARITH=1 in the environment: both sides on the split-f16 kernels.  AB_UTTS=k: the frames as a batch of k utterances.
Each library is loaded under its own handle; per system the two are timed alternately (order swapped every
round), 9 rounds of 100 launches, first round dropped, medians reported.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from phnrec_amd import capi, modelgen  # noqa: E402


class Ctx:
    def __init__(self, lib, mdir, nbanks):
        self.L = lib
        self.h = C.c_void_p()
        vp = C.c_void_p
        lib.lcrc_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.lcrc_posteriors_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        lib.lcrc_num_outputs.argtypes = [vp]
        lib.lcrc_set_timing.argtypes = [vp, C.c_int]
        assert lib.lcrc_create(C.byref(self.h), mdir.encode(), nbanks, 31, 1, 0) == 0
        lib.lcrc_set_timing(self.h, 0)
        if os.environ.get("LCRC_ARITH", "0") != "0":          # split-f16 arithmetic on both sides
            lib.lcrc_set_arithmetic.argtypes = [vp, C.c_int]
            assert lib.lcrc_set_arithmetic(self.h, int(os.environ["LCRC_ARITH"])) == 0
        self.n_out = lib.lcrc_num_outputs(self.h)

    def run(self, mel, post, n, stream):
        off = getattr(self, "d_off", None)        # AB_UTTS=k: a batch of k utterances (offsets resident on the device)
        assert self.L.lcrc_posteriors_device(self.h, mel.data_ptr(), off.data_ptr() if off is not None else None,
                                             self.n_utts if off is not None else 1, n, post.data_ptr(), stream) == 0


def main():
    pa, pb = (os.path.join(ROOT, p) for p in sys.argv[1:3])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    capi._load_hip_runtime()
    libs = [C.CDLL(pa), C.CDLL(pb)]
    s = torch.cuda.current_stream()
    print("A = %s\nB = %s\n%d frames; ms per launch, medians of 8 interleaved rounds of 100" % (sys.argv[1], sys.argv[2], n))
    for system, spec in modelgen.SYSTEMS.items():
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        if not os.path.isdir(mdir):
            mdir = "/tmp/ab_model_" + system
            modelgen.write_system(mdir, system, seed=1)
        ctxs = [Ctx(L, mdir, spec["nbanks"]) for L in libs]
        k = int(os.environ.get("AB_UTTS", "0"))
        if k > 1:
            off = torch.tensor(np.linspace(0, n, k + 1).astype(np.int32), device="cuda")
            for c in ctxs:
                c.d_off, c.n_utts = off, k
                c.L.lcrc_set_hidden_split.argtypes = [C.c_void_p, C.c_int]
                c.L.lcrc_set_hidden_split(c.h, 1)              # as the CLI runs: fused kernels only
        mel = torch.from_numpy(modelgen.synth_mel(n, spec["nbanks"], seed=1)).cuda()
        posts = [torch.empty((n, c.n_out), device="cuda") for c in ctxs]
        for c, p in zip(ctxs, posts):
            for _ in range(20):
                c.run(mel, p, n, s.cuda_stream)
        s.synchronize()
        same = bool(torch.equal(posts[0], posts[1]))
        t = [[], []]
        for rnd in range(9):                     # order alternates; the first round (clock ramp) is dropped
            for k in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(100):
                    ctxs[k].run(mel, posts[k], n, s.cuda_stream)
                e1.record(s)
                s.synchronize()
                t[k].append(e0.elapsed_time(e1) / 100)
        a, b = np.array(t[0][1:]), np.array(t[1][1:])
        print("%-26s A %.4f (+-%.4f)  B %.4f (+-%.4f)  B/A %.4f  identical output: %s"
              % (system, np.median(a), a.std(), np.median(b), b.std(), np.median(b) / np.median(a), same))


if __name__ == "__main__":
    main()
