#!/usr/bin/env python3
"""Condense rocprofv3 --pmc CSV outputs (gpurun_out/pmc_*/) into profiles/<tag>_pmc.json and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE reports half of a wide coalesced read, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "lcrc_fused_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)} for k, v in sorted(agg.items())}
    derived = {}
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        fetch = out["FETCH_SIZE"]["mean"] * 1024 * 2      # gfx950 correction: x2
        write = out["WRITE_SIZE"]["mean"] * 1024
        derived["hbm_bytes_per_launch"] = fetch + write
        derived["fetch_bytes_corrected"] = fetch
        derived["write_bytes"] = write
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w") as f:
            json.dump({"bytes_per_launch": round(fetch + write), "fetch_bytes_x2": round(fetch),
                       "write_bytes": round(write), "source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                       "(separate passes), bench.py batch 8192 CZ, FETCH doubled per the gfx950 note" % tag}, f, indent=1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
        # busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
        derived["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / (out["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
        h, m = out["TCC_HIT_sum"]["mean"], out["TCC_MISS_sum"]["mean"]
        derived["l2_hit_rate"] = h / (h + m)
    res = {"kernel": "lcrc_fused_kernel (CZ, 8192 frames)", "counters": out, "derived": derived}
    path = os.path.join(ROOT, "profiles", "%s_pmc.json" % tag)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
