#!/usr/bin/env python3
"""lcrc_push of 5 frames (host buffers in, posteriors out) against the forced number of workgroups per frame tile of the
split-hidden path (lcrc_set_hidden_split): where kSplitMax = 12 comes from.  Needs a GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phnrec_amd import capi, modelgen
for system in ("PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500"):
    nb = modelgen.SYSTEMS[system]["nbanks"]
    ctx = capi.Lcrc(os.path.join(ROOT, "tests", "golden", "models", system), nb)
    hmel = modelgen.synth_mel(4000, nb, seed=2)
    ref = None
    for split in (0, 8, 12, 16, 20, 24, 32, 46):
        ctx.set_hidden_split(split)
        t = []
        for rnd in range(5):
            ctx.reset()
            for i in range(0, 500, 5):
                ctx.push(hmel[i:i + 5])
            t0 = time.perf_counter()
            outs = [ctx.push(hmel[i:i + 5]) for i in range(500, 3500, 5)]
            t.append((time.perf_counter() - t0) / 600)
        out = np.concatenate(outs)
        if ref is None: ref = out
        print(system, "split", split, "push5 %.1f us" % (np.median(t) * 1e6), "max|diff vs auto| %.2g" % np.abs(out - ref).max(), flush=True)
    ctx.close()
