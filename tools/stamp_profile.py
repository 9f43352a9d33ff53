#!/usr/bin/env python3
"""Phase breakdown of the fused kernel from in-kernel s_memtime stamps (DIAGNOSTIC build).

Needs `make -C phnrec_amd/csrc stamps` and a GPU.  Prints, per phase, the median over
workgroups of the cycles the slowest wave spent there -- read the SHARES, not the total
(the stamped build forbids overlaps the product kernel has).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from phnrec_amd import capi, modelgen  # noqa: E402

NAMES = ["stage0 stage (mel tile, tables, zero images)", "stage1 projection",
         "band0 hidden loop", "band0 fold+softmax", "band0 ln epilogue",
         "band1 hidden loop", "band1 fold+softmax", "band1 ln epilogue",
         "merger hidden loop", "merger fold+softmax", "store"]
ORDER = [0, 1, 10, 2, 3, 4, 5, 6, 7, 8, 9, 11]     # stamp indices in program order


def main():
    system = sys.argv[1] if len(sys.argv) > 1 else "PHN_CZ_SPDAT_LCRC_N1500"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    dbg = os.environ.get("LCRC_DBG", "0")
    capi.LIB_PATH = os.path.join(ROOT, "phnrec_amd", "lib",
                                 "libphnrec_lcrc_stamps%s.so" % ("" if dbg == "0" else "_dbg" + dbg))
    L = capi.load()
    L.lcrc_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    spec = modelgen.SYSTEMS[system]
    mdir = os.path.join(ROOT, "tests", "golden", "models", system)
    if not os.path.isdir(mdir):
        mdir = "/tmp/stamp_model_" + system
        modelgen.write_system(mdir, system, seed=1)
    ctx = capi.Lcrc(mdir, spec["nbanks"])
    print("ablation build LCRC_DBG =", dbg)
    mel = torch.from_numpy(modelgen.synth_mel(n, spec["nbanks"], seed=1)).cuda()
    post = torch.empty((n, ctx.n_out), device="cuda")
    grid = (n + 31) // 32
    stamps = torch.zeros((grid, 8, 16), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    for it in range(5):
        if it == 4:
            L.lcrc_debug_set_stamps(ctx.h, stamps.data_ptr())
        ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    st = stamps.cpu().numpy()[:, :4, :].astype(np.int64)      # 4 waves per workgroup
    seq = st[:, :, ORDER]
    d = np.diff(seq, axis=2)                                   # [grid][wave][phase]
    tot = seq[:, :, -1] - seq[:, :, 0]
    print("%s, %d frames, %d workgroups; cycles (s_memtime ticks = shader clocks)" % (system, n, grid))
    print("%-46s %10s %10s %7s" % ("phase", "median", "max-wave", "share"))
    total_med = np.median(tot.max(axis=1))
    for i, name in enumerate(NAMES):
        per_wg = d[:, :, i].max(axis=1)
        print("%-46s %10.0f %10.0f %6.1f%%" % (name, np.median(d[:, :, i]), np.median(per_wg),
                                              100.0 * np.median(per_wg) / total_med))
    print("%-46s %10.0f" % ("workgroup total (median of slowest wave)", total_med))
    # finer split of the LAST net's tail: loop end (8) -> fold done (12) -> softmax math done (13) -> epilogue done (9)
    fine = st[:, :, [8, 12, 13, 9]]
    df = np.diff(fine, axis=2)
    for i, name in enumerate(("merger: fold (barriers + slab traffic)", "merger: softmax math", "merger: epilogue + barrier")):
        print("%-46s %10.0f %10.0f" % (name, np.median(df[:, :, i]), np.median(df[:, :, i].max(axis=1))))
    print("kernel ms (events, stamped build): %.4f" % ctx.last_kernel_ms())
    span = seq[:, :, -1].max() - seq[:, :, 0].min()
    print("first stamp -> last stamp over the whole grid: %d cycles" % span)


if __name__ == "__main__":
    main()
