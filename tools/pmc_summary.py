#!/usr/bin/env python3
"""Condense rocprofv3 --pmc CSV outputs (gpurun_out/.../pmc_*/) into profiles/<tag>_pmc.json and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

Population: only dispatches of `lcrc_fused_kernel` with the HEADLINE grid (8192 rows as pairs of 16-frame
workgroups = 512 workgroups x 256 threads = 131 072 work-items) count, and of those only the steady ones: the
first SKIP such dispatches of every pass (clock ramp, code load) are dropped.  The passes are separate
processes running the same `bench.py --kernel-only` command, so the k-th kept dispatch of one pass is the
same launch of the same program as the k-th kept dispatch of another; counters are paired by that index and
every pass must have kept the same number of dispatches (the summary says so, or fails).

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB, collected in separate passes; on gfx950 FETCH_SIZE reports half of a wide coalesced
read, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADLINE_GRID = 131072          # 8192 rows / 16 rows per workgroup x 256 threads
SKIP = 100                      # dispatches of the headline grid dropped at the start of every pass
KERNEL_MS_HINT = None


def passes(src):
    """{pass directory: {counter: [values of the kept dispatches, in dispatch order]}}, plus kernel durations"""
    res, dur = {}, {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        per = collections.defaultdict(dict)        # dispatch id -> {counter: value}
        t = {}
        # (one process per pass: if an earlier call's files were merged into the same directory, the newest process counts)
        for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                if "lcrc_fused_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) != HEADLINE_GRID:
                    continue
                did = int(r["Dispatch_Id"])
                per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                t[did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        ids = sorted(per)[SKIP:]
        if not ids:
            continue
        cols = collections.defaultdict(list)
        for i in ids:
            for k, v in per[i].items():
                cols[k].append(v)
        res[os.path.basename(d)] = dict(cols)
        dur[os.path.basename(d)] = [t[i] for i in ids]
    return res, dur


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
    per_pass, dur = passes(src)
    if not per_pass:
        raise SystemExit("no headline-grid dispatches of lcrc_fused_kernel under %s/pmc_*" % src)
    kept = {p: len(next(iter(c.values()))) for p, c in per_pass.items()}
    if len(set(kept.values())) != 1:
        raise SystemExit("the passes kept different numbers of dispatches: %s" % kept)
    out, where = {}, {}
    for p, cols in per_pass.items():
        for k, v in cols.items():
            out[k] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v), "pass": p}
            where[k] = p
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
                       "(separate passes of `bench.py --kernel-only`; only 8192-row dispatches, the first %d of each "
                       "pass dropped; FETCH doubled per the gfx950 note)" % (tag, SKIP)}, f, indent=1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in per_pass.get(where.get("GRBM_GUI_ACTIVE", ""), {}):
        # paired per dispatch index: busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
        busy = per_pass[where["SQ_VALU_MFMA_BUSY_CYCLES"]]["SQ_VALU_MFMA_BUSY_CYCLES"]
        act = per_pass[where["GRBM_GUI_ACTIVE"]]["GRBM_GUI_ACTIVE"]
        fr = [b / 1024.0 / (a / 8.0) for b, a in zip(busy, act)]
        derived["mfma_busy_frac"] = sum(fr) / len(fr)
        derived["mfma_busy_frac_min_max"] = [min(fr), max(fr)]
    if "SQ_INSTS_MFMA" in out and "GRBM_GUI_ACTIVE" in out:
        # the instruction-count form: every v_mfma_f32_16x16x4_f32 holds its SIMD's matrix pipe for 32 cycles
        n = out["SQ_INSTS_MFMA"]["mean"]
        derived["mfma_insts_x32_over_1024_simds_over_kernel_cycles"] = n * 32.0 / 1024.0 / (out["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
        derived["mfma_flop_executed"] = n * 2048.0
    if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
        h, m = out["TCC_HIT_sum"]["mean"], out["TCC_MISS_sum"]["mean"]
        derived["l2_hit_rate"] = h / (h + m)
    alld = [x for v in dur.values() for x in v]
    derived["kernel_ms_under_pmc"] = {"mean": sum(alld) / len(alld), "min": min(alld), "max": max(alld)}
    res = {"kernel": "lcrc_fused_kernel (CZ, 8192 frames: grid %d only)" % HEADLINE_GRID,
           "population": "bench.py --kernel-only; per pass the first %d headline dispatches dropped, %d kept" % (SKIP, next(iter(kept.values()))),
           "counters": out, "derived": derived}
    path = os.path.join(ROOT, "profiles", "%s_pmc.json" % tag)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
