"""Shared helpers for the tests (HTK file I/O, fixture paths)."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def read_htk(path):
    """HTK parameter file: BE int32 nSamples, int32 sampPeriod, int16 sampSize,
    int16 paramKind, then BE float32 rows (matrix.h:75-82, 2547-2573)."""
    with open(path, "rb") as f:
        b = f.read()
    n, period, size, kind = struct.unpack(">iihh", b[:12])
    a = np.frombuffer(b[12:12 + n * size], dtype=">f4").reshape(n, size // 4)
    return a.astype(np.float32)


def read_htk_header(path):
    with open(path, "rb") as f:
        return struct.unpack(">iihh", f.read(12))


def write_htk(path, a, period=100000, kind=6):
    a = np.ascontiguousarray(a, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack(">iihh", a.shape[0], period, a.shape[1] * 4, kind))
        f.write(a.astype(">f4").tobytes())


def model_dir(system):
    """Real model directory shipped as test data (all four LCRC systems: CZ, HU, RU, EN); None if absent."""
    p = os.path.join(GOLD, "models", system)
    return p if os.path.isdir(p) else None


def read_rec(path):
    rows = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 4:
                rows.append((int(p[0]), int(p[1]), p[2], float(p[3])))
    return rows
