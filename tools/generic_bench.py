#!/usr/bin/env python3
"""Kernel time of the run-time-shape (`generic`) LCRC variant on seeded synthetic models whose shapes match none of the
shipped systems.  usage: generic_bench.py [LIB_A [LIB_B]]  (paths relative to the repo root; default: the product
library; with two libraries their outputs are compared too -- see tools/build_ab_lib.sh).  Needs a GPU."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np, torch
from phnrec_amd import capi, modelgen
def bench(libpath, d, nb, n=8192):
    capi._load_hip_runtime()
    L = C.CDLL(libpath)
    vp = C.c_void_p
    L.lcrc_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lcrc_posteriors_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lcrc_num_outputs.argtypes = [vp]; L.lcrc_set_timing.argtypes = [vp, C.c_int]
    L.lcrc_kernel_name.argtypes = [vp]; L.lcrc_kernel_name.restype = C.c_char_p
    h = vp(); assert L.lcrc_create(C.byref(h), d.encode(), nb, 31, 1, 0) == 0
    L.lcrc_set_timing(h, 0)
    no = L.lcrc_num_outputs(h)
    mel = torch.from_numpy(modelgen.synth_mel(n, nb, seed=1)).cuda(); post = torch.empty((n, no), device="cuda")
    s = torch.cuda.current_stream()
    for _ in range(400): L.lcrc_posteriors_device(h, mel.data_ptr(), None, 1, n, post.data_ptr(), s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(100): L.lcrc_posteriors_device(h, mel.data_ptr(), None, 1, n, post.data_ptr(), s.cuda_stream)
    e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / 100, L.lcrc_kernel_name(h).decode(), post.cpu().numpy()
for nb, hid, nout in ((16, 1500, 138), (15, 1000, 100), (23, 700, 150)):
    with tempfile.TemporaryDirectory() as d:
        modelgen.write_model_dir(d, nb, hid, nout, seed=5)
        flop = 2 * (2 * (nb * 11 * hid + hid * nout) + (2 * nout * hid + hid * nout))
        libs = sys.argv[1:3] or ["phnrec_amd/lib/libphnrec_lcrc.so"]
        res = [bench(l, d, nb) for l in libs]
        line = "banks %d hidden %d out %d [%s]:" % (nb, hid, nout, res[-1][1])
        for l, (ms, _, _) in zip(libs, res):
            line += "  %s %.4f ms (%.3f of peak)" % (os.path.basename(l), ms, 8192 * flop / ms / 1e-3 / 157.3e12)
        if len(res) == 2:
            line += "  identical output: %s" % bool(np.array_equal(res[0][2], res[1][2]))
        print(line)
