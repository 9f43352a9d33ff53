#!/usr/bin/env python3
"""One CPU worker of bench.py's `cpu_baseline` leg (test infrastructure, like everything under oracle/).

Runs the REFERENCE's Traps/NeuralNet (oracle/_ref, built from /root/reference by oracle/Makefile) on a seeded
synthetic utterance for a bounded time in a process of its own -- the reference is not thread-safe (FEXP's
workspace is a global, fexp.h:23-31), so "all cores" means one process per core, utterance-sharded, which is
also how the reference would be scaled in practice.  Prints one JSON line.

usage: cpu_worker.py MODEL_DIR NBANKS BUNCH BLAS(0|1) SECONDS SEED FRAMES
"""
import json
import os
import sys
import time

os.environ.setdefault("MKL_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mdir, nbanks, bunch, blas, seconds, seed, frames = sys.argv[1:8]
    from oracle import binding as ob
    from phnrec_amd import modelgen
    mel = modelgen.synth_mel(int(frames), int(nbanks), seed=int(seed))
    t = ob.RefTraps(mdir, int(nbanks), bunch=int(bunch), blas=bool(int(blas)))
    t.process_offline(mel[:64])                      # page in code and weights
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < float(seconds):
        t.process_offline(mel)
        done += mel.shape[0]
    print(json.dumps({"frames": done, "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main()
