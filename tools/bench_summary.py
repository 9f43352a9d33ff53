#!/usr/bin/env python3
"""The figures DESIGN 6 / 7 quote, from bench records: bench_summary.py a.json [b.json ...]"""
import json
import sys

for f in sys.argv[1:]:
    r = json.load(open(f))
    rl = r["roofline"]
    print(f)
    print(" value %.2f M  ms %.4f  frac %.4f  kernel_ms %.4f  cold %s  traffic %s" % (
        r["value"] / 1e6, r["ms_per_step"], rl["frac"], rl["kernel_ms"], rl["cold"], rl.get("traffic")))
    for leg in ("systems", "small_launches"):
        for k, v in r.get(leg, {}).items():
            if isinstance(v, dict) and "frac" in v:
                print("   %-10s %.4f ms  %.4f" % (k, v["kernel_ms"], v["frac"]))
    print("   push5 %.2f us  push512 %.2f us" % (r["push_bunch5"]["us_per_call"], r["push_bunch512"]["us_per_call"]))
    print("   host_path %.4f  zero-copy %.4f  wave %.4f  tree %.4f  wave_en %.4f" % (
        r["host_path"]["ms_per_call"], r["host_path_zero_copy"]["ms_per_call"], r["wave_path"]["ms_per_call"],
        r["wave_path"]["tree_mean_ms_per_call"], r["wave_path_en"]["ms_per_call"]))
    sf = r["single_file"]
    print("   single_file str %.3f (min %.3f)  post %.3f  reference MKL %.3f  create %s" % (
        sf["str"]["process_wall_s"], sf["str"]["min_process_wall_s"], sf["post"]["process_wall_s"],
        sf["reference_cpu_mkl"]["process_wall_s"], sf["str"]["create_trace_ms"]))
    print("   dropin %.0f" % r["dropin_reference_cli"]["value"])
    cb = r["cpu_baseline"]
    print("   cpu sgemv %.0f" % cb["value"], [(k, v.get("value")) for k, v in cb.items() if isinstance(v, dict) and "value" in v])
    sl = r["sharded_list"]
    modes = ("host_frontend", "gpu_energies_E", "gpu_energies_decoder_E_D", "gpu_frontend_F", "gpu_frontend_decoder_F_D")
    for k in modes:
        print("   %-26s %.2f M  host_cpu %.3f s  list %.3f s  process %.3f s  set-up %.3f s  mode %s" % (
            k, sl[k]["value"] / 1e6, sl[k]["host_cpu_s"], sl[k]["list_wall_s"], sl[k]["process_wall_s"], sl[k].get("setup_s", 0), sl[k].get("mode")))
    hc = sl["host_ceiling"]
    print("   ceilings", {k: round(hc[k]["frames_per_s"] / 1e6, 1) for k in modes}, "per-file serial %.0f files/s = %.1f M" % (
        hc["per_file_serial"]["files_per_s"], hc["per_file_serial"]["frames_per_s_at_this_lists_file_length"] / 1e6))
    wl = sl.get("weak_list", {})
    print("   weak list: %s files, %.1f M frames" % (wl.get("files"), wl.get("frames", 0) / 1e6))
    for k in modes + ("as_g8_default",):
        v = wl.get(k)
        if isinstance(v, dict) and "value" in v:
            print("     %-26s %.2f M (process %.2f M, %.2f s)  host_cpu %.2f s  ceiling %.0f M = %.2f x 8 GPUs  mode %s" % (
                k, v["value"] / 1e6, v["process_frames_per_s"] / 1e6, v["process_wall_s"], v["host_cpu_s"],
                v.get("host_ceiling_frames_per_s", 0) / 1e6, v.get("ceiling_over_8_gpus", 0), v.get("mode")))
    print("     F_D_over_F", wl.get("F_D_over_F"))
    cz = sl.get("cz_same_list", {})
    print("   cz_same_list", {k: round(v["value"] / 1e6, 2) for k, v in cz.items() if isinstance(v, dict) and "value" in v})
    fs = r.get("four_systems", {})
    for k in ("default_flags", "gpu_frontend_decoder_F_D"):
        v = fs.get(k)
        if isinstance(v, dict) and "value" in v:
            print("   four_systems %-26s %.2f M frames/s in sum (whole script %.2f s, xRT %.2e; list loops %.1f M)  %s" % (
                k, v["value"] / 1e6, v["process_wall_s"], v["xrt"], v["list_loops_frames_per_s"] / 1e6,
                {n[4:6]: round(p["frames_per_s"] / 1e6, 1) for n, p in v["per_system"].items()}))
    print("   four_systems MLFs equal single-system runs:", fs.get("mlf_equals_single_system_run"))
    print("   split_f16 %.4f ms  %.1f M" % (r["split_f16"]["kernel_ms"], r["split_f16"]["value"] / 1e6))
