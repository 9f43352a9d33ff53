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
         "band nets (side by side) hidden loop", "band nets softmax + ln epilogue",
         "merger hidden loop", "merger fold+softmax", "store"]
ORDER = [0, 1, 10, 2, 3, 8, 9, 11]     # stamp indices in program order


def main():
    system = sys.argv[1] if len(sys.argv) > 1 else "PHN_CZ_SPDAT_LCRC_N1500"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    capi.LIB_PATH = os.path.join(ROOT, "phnrec_amd", "lib", "libphnrec_lcrc_stamps.so")
    L = capi.load()
    L.lcrc_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    spec = modelgen.SYSTEMS[system]
    mdir = os.path.join(ROOT, "tests", "golden", "models", system)
    if not os.path.isdir(mdir):
        mdir = "/tmp/stamp_model_" + system
        modelgen.write_system(mdir, system, seed=1)
    ctx = capi.Lcrc(mdir, spec["nbanks"])
    mel = torch.from_numpy(modelgen.synth_mel(n, spec["nbanks"], seed=1)).cuda()
    post = torch.empty((n, ctx.n_out), device="cuda")
    bm = int(os.environ.get("LCRC_BM", "32"))                 # frames per workgroup (forced below)
    ctx.set_tile_frames(bm)
    arith = int(os.environ.get("LCRC_ARITH", "0"))            # 1 = split-f16 arithmetic
    if arith:
        ctx.L.lcrc_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
        ctx.set_arithmetic(arith)
        print("split-f16 arithmetic")
    grid = (n + bm - 1) // bm
    stamps = torch.zeros((grid, 8, 16), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream()
    for it in range(5):
        if it == 4:
            L.lcrc_debug_set_stamps(ctx.h, stamps.data_ptr())
        ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    nw = 4                                                    # waves per workgroup
    st = stamps.cpu().numpy()[:, :nw, :].astype(np.int64)
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
    # residency: workgroups grouped by the CU they ran on (slot 14 = XCC_ID << 32 | HW_ID; HW_ID bits:
    # simd 5:4, cu 11:8, sh 12, se 15:13); one clock per CU group is assumed
    hw = stamps.cpu().numpy()[:, 0, 14].astype(np.int64)
    cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 8) & 0xFF)
    b = seq[:, :, 0].min(axis=1)
    e = seq[:, :, -1].max(axis=1)
    ids = np.unique(cu)
    per_cu = np.array([(cu == c).sum() for c in ids])
    print("CUs used: %d; workgroups per CU: min %d max %d" % (len(ids), per_cu.min(), per_cu.max()))
    ov, spans, start_gap, end_gap = [], [], [], []
    for c in ids:
        bb, ee = b[cu == c], e[cu == c]
        spans.append(ee.max() - bb.min())
        if len(bb) == 2:
            ov.append((min(ee) - max(bb)) / float(max(ee) - min(bb)))
            start_gap.append(max(bb) - min(bb))
            end_gap.append(max(ee) - min(ee))
    print("busy span per CU (first start -> last end): median %d max %d" % (np.median(spans), np.max(spans)))
    if per_cu.min() == 2 and per_cu.max() == 2:
        # the pair's two workgroups separately: phase by phase, the one that ends first and the one that ends last
        first, last = [], []
        for c in ids:
            w = np.nonzero(cu == c)[0]
            w = w[np.argsort(e[w])]
            first.append(w[0]); last.append(w[1])
        print("%-46s %12s %12s" % ("phase (slowest wave), pairs", "ends first", "ends last"))
        for i, name in enumerate(NAMES):
            per_wg = d[:, :, i].max(axis=1)
            print("%-46s %12.0f %12.0f" % (name, np.median(per_wg[first]), np.median(per_wg[last])))
        print("%-46s %12.0f %12.0f" % ("workgroup total", np.median(tot.max(axis=1)[first]), np.median(tot.max(axis=1)[last])))
        lo = np.array([min(w) for w in zip(first, last)])
        print("the workgroup that ends first is the one with the lower index in %d of %d pairs" % ((np.array(first) == lo).sum(), len(first)))
    if ov:
        print("CUs with two workgroups: %d; overlap of their lifetimes: median %.2f min %.2f" % (len(ov), np.median(ov), np.min(ov)))
        print("  second workgroup starts after the first by: median %d min %d max %d cycles; ends after it by: median %d min %d max %d"
              % (np.median(start_gap), np.min(start_gap), np.max(start_gap), np.median(end_gap), np.min(end_gap), np.max(end_gap)))
    print("XCC ids seen:", sorted(set(((hw >> 32) & 0xF).tolist())), " sample HW_ID:", [hex(int(x & 0xFFFFFFFF)) for x in hw[:4]])


if __name__ == "__main__":
    main()
