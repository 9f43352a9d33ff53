#!/usr/bin/env python3
"""Can an utterance's time axis be split for the device decoder?  A second wave would start at frame
s - W with fresh tokens and its labels would be spliced behind frame s.  This probe runs the decoder ORACLE (CPU, this
is a study, not the product) on the reference's posterior dumps of the bundled utterance: the whole utterance against a
start at s - W, for every s on a grid and several W, and counts the spliced tails whose labels (times, phonemes) and whose
SCORES equal the whole run's bit for bit.  usage: decoder_split_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from tests.util import GOLD, read_htk  # noqa: E402

SYS = {"PHN_CZ_SPDAT_LCRC_N1500": (45, -4.6875), "PHN_HU_SPDAT_LCRC_N1500": (61, -2.8125),
       "PHN_RU_SPDAT_LCRC_N1500": (52, -0.9375), "PHN_EN_TIMIT_LCRC_N500": (39, -2.03125)}
for name, (P, wpen) in SYS.items():
    post = read_htk(os.path.join(GOLD, name, "test.lop"))
    lp = np.log(np.maximum(post, 1e-37)).astype(np.float32)
    whole = ob.phndec(lp, P, 3, 40, wpen)                       # [(start, end, phn, score)]
    n = len(lp)
    for W in (40, 80, 160, 320):
        same_lab = same_all = tried = 0
        for s in range(W + 40, n - 80, 37):
            tail = ob.phndec(lp[s - W:], P, 3, 40, wpen)
            tail = [(a + s - W, b + s - W, p, sc) for (a, b, p, sc) in tail]
            ref = [x for x in whole if x[0] >= s]
            got = [x for x in tail if x[0] >= s]
            if not ref:
                continue
            tried += 1
            lab = [x[:3] for x in ref] == [x[:3] for x in got]
            same_lab += lab
            same_all += lab and all(np.float32(a[3]).tobytes() == np.float32(b[3]).tobytes() for a, b in zip(ref, got))
        print("%s  %d frames, %d labels; restart %3d frames ahead of the splice: %3d splice points, labels equal at %3d, "
              "labels AND scores bit-equal at %3d" % (name[4:6], n, len(whole), W, tried, same_lab, same_all))
