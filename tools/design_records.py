#!/usr/bin/env python3
"""Regenerates the tables of measured figures in DESIGN.md (section 6 records, section 7 modes) and README.md from the round's
bench records, between the <!-- records:... --> markers (the sentences around them are written by hand).
    design_records.py [TAG = r06]          reads profiles/TAG_bench_driver_form.json [D], TAG_bench.json [B], TAG_pmc.json,
                                           TAG_kernel_stats.csv, hbm_traffic.json"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda *a: os.path.join(ROOT, "profiles", *a)
D = json.load(open(P(TAG + "_bench_driver_form.json")))
B = json.load(open(P(TAG + "_bench.json")))
PMC = json.load(open(P(TAG + "_pmc.json")))["derived"]
KS = next(r for r in csv.DictReader(open(P(TAG + "_kernel_stats.csv"))) if "lcrc_fused_kernel" in r["Name"])
M = lambda x: "%.1f M" % (x / 1e6)
MODES = ("host_frontend", "gpu_energies_E", "gpu_energies_decoder_E_D", "gpu_frontend_F", "gpu_frontend_decoder_F_D")


def both(f, fmt="%s"):
    return (fmt + " [D], " + fmt + " [B]") % (f(D), f(B))


def records():
    r, rb = D["roofline"], B["roofline"]
    sl, slb = D["sharded_list"], B["sharded_list"]
    wl, wlb = sl["weak_list"], slb["weak_list"]
    g8, g8b = sl["as_g8_on_1x_list"], slb["as_g8_on_1x_list"]
    sf, sfb = D["single_file"], B["single_file"]
    fs, fsb = D["four_systems"], B["four_systems"]
    c = D["cpu_baseline"]
    sysm = lambda rec, k: rec["systems"][k]
    sm = lambda k: D["small_launches"][k]
    rows = [
        ("frames/s, 1 GPU, CZ 8192 frames per launch",
         "**%.2f M** (%.4f ms per step) [D], %.2f M [B]; xRT %.1e" % (D["value"] / 1e6, D["ms_per_step"], B["value"] / 1e6, D["xrt"]), "`value`"),
        ("kernel, roofline",
         "%.4f ms = %.1f TFLOP/s algorithmic = **%.3f** of the f32 MFMA peak [D]; %.4f = %.3f [B]; seven further windows of 40 launches "
         "%.4f–%.4f; launches 1–20 of a process %.2f ms = %.2f; rocprofv3, the 200 timed launches of the traced run: **%.1f µs** avg, %.1f–%.1f"
         % (r["kernel_ms"], r["achieved"], r["frac"], rb["kernel_ms"], rb["frac"], min(r["windows"]["kernel_ms_min"], rb["windows"]["kernel_ms_min"]),
            max(r["windows"]["kernel_ms_max"], rb["windows"]["kernel_ms_max"]), r["cold"]["kernel_ms"], r["cold"]["frac"],
            float(KS["AverageNs"]) / 1e3, float(KS["MinNs"]) / 1e3, float(KS["MaxNs"]) / 1e3), "`roofline`; `%s_kernel_stats.csv`" % TAG),
        ("MFMA pipe, PMC (headline launches only, the first 100 of each pass dropped, 140 kept, paired by dispatch index)",
         "`SQ_INSTS_MFMA` 12 622 848 min = max; `SQ_VALU_MFMA_BUSY_CYCLES` ÷ 1024 SIMDs ÷ kernel cycles = **%.3f** (%.3f–%.3f per dispatch) = the "
         "instruction-count form (%.3f; the PMC passes' launches: %.4f ms); %.2f GFLOP executed for 25.07 algorithmic (3.1 %% tile padding)"
         % (PMC["mfma_busy_frac"], PMC["mfma_busy_frac_min_max"][0], PMC["mfma_busy_frac_min_max"][1],
            PMC["mfma_insts_x32_over_1024_simds_over_kernel_cycles"], PMC["kernel_ms_under_pmc"]["mean"], PMC["mfma_flop_executed"] / 1e9), "`%s_pmc.json`" % TAG),
        ("HBM side, PMC",
         "%.1f MB per launch (FETCH × 2 per the gfx950 note %.1f MB + `WRITE_SIZE` 4416 KiB = 8192 × 552 B exactly) = %.0f GB/s = 4 %% of the HBM "
         "roof; L2 hit %.1f %%; %.1f × the algorithmic 5.0 MB: each of the eight XCD L2s (4 MiB) fetches the 6.2 MB weight set from Infinity "
         "Cache at least once per launch" % (PMC["hbm_bytes_per_launch"] / 1e6, PMC["fetch_bytes_corrected"] / 1e6, r["hbm_gbps"],
                                            PMC["l2_hit_rate"] * 100, PMC["hbm_bytes_per_launch"] / 5.013e6), "`hbm_traffic.json`"),
        ("the other shipped systems at 8192 frames (median of seven windows)",
         "HU %.4f ms = **%.3f**, RU %.4f = **%.3f**, EN %.4f = **%.3f** [D]; %.3f / %.3f / %.3f [B]"
         % (sysm(D, "hu_8192")["kernel_ms"], sysm(D, "hu_8192")["frac"], sysm(D, "ru_8192")["kernel_ms"], sysm(D, "ru_8192")["frac"],
            sysm(D, "en_8192")["kernel_ms"], sysm(D, "en_8192")["frac"], sysm(B, "hu_8192")["frac"], sysm(B, "ru_8192")["frac"], sysm(B, "en_8192")["frac"]), "`systems`"),
        ("small launches",
         "CZ 4096 frames %.4f ms = %.3f, CZ 2048 %.4f = %.3f; EN 4096 (configs[1]) %.4f = **%.3f**, EN 2048 %.4f = %.3f [D] (bound measured in "
         "round 5, `r05_ab_runs.txt` 1; closed)" % (sm("cz_4096")["kernel_ms"], sm("cz_4096")["frac"], sm("cz_2048")["kernel_ms"], sm("cz_2048")["frac"],
                                                      sm("en_4096")["kernel_ms"], sm("en_4096")["frac"], sm("en_2048")["kernel_ms"], sm("en_2048")["frac"]), "`small_launches`"),
        ("streaming, `lcrc_reset` / `lcrc_push`",
         "shipped bunch of 5: **%.1f µs** per call = %.0f k frames/s; 512: %.0f µs = %.1f M frames/s"
         % (D["push_bunch5"]["us_per_call"], D["push_bunch5"]["value"] / 1e3, D["push_bunch512"]["us_per_call"], D["push_bunch512"]["value"] / 1e6),
         "`push_bunch5`, `push_bunch512`"),
        ("host-pointer entries (PCIe-inclusive, never `value`)",
         "`lcrc_posteriors` on reused pageable buffers %.3f ms per 8192 frames; zero-copy `lcrc_stage_run` %.3f ms = %s frames/s; from A-law bytes "
         "%.3f ms = %s; EN 16 kHz lin16, 4096 frames, %.3f ms" % (D["host_path"]["ms_per_call"], D["host_path_zero_copy"]["ms_per_call"],
                                                                  M(D["host_path_zero_copy"]["value"]), D["wave_path"]["ms_per_call"], M(D["wave_path"]["value"]),
                                                                  D["wave_path_en"]["ms_per_call"]), "`host_path`, `host_path_zero_copy`, `wave_path`, `wave_path_en`"),
        ("single file = the reference's smoke test as a process (7.5 s of audio), on a GPU nobody else holds",
         "**%.3f s** (min %.3f; `hipInit` + device + first stream %.0f ms of it) [D], %.3f (min %.3f) [B] — against the reference's MKL build "
         "**%.3f s**: the GPU path still LOSES this case, by the runtime's start-up"
         % (sf["str"]["process_wall_s"], sf["str"]["min_process_wall_s"], sf["str"]["create_trace_ms"]["HIP device, stream, events"],
            sfb["str"]["process_wall_s"], sfb["str"]["min_process_wall_s"], sf["reference_cpu_mkl"]["process_wall_s"]), "`single_file`"),
        ("the literal drop-in: the reference's own CLI over the library",
         "%.0f k frames/s (its single-threaded front-end, decoder and file loop)" % (D["dropin_reference_cli"]["value"] / 1e3), "`dropin_reference_cli`"),
        ("CPU baseline, reference code on the box's host",
         "%.1f k frames/s (1 core, MKL sgemv, shipped `bunch_size=5`); %.1f k (1 core) / %.0f k (%d cores) with sgemm at bunch 512; parity of the "
         "timed launches' output against it %.1e (bar 1e-4)" % (c["value"] / 1e3, c["sgemm_1core"]["value"] / 1e3, c["sgemm_all_cores"]["value"] / 1e3,
                                                                c["sgemm_all_cores"]["cores"], c["parity_max_abs_vs_gpu"]), "`cpu_baseline`"),
        ("configs[3]: HU, 10 000 files (8.94 M frames) → MLF, one GPU, as a PROCESS (exec → exit; each process 0.3 s after the previous one's "
         "exit, the bench process not yet on the GPU)",
         "host front-end %.3f s, `-E` %.3f, `-E -D` %.3f, `-F` **%.3f**, `-F -D` **%.3f** [D]; %.3f / %.3f / %.3f / %.3f / %.3f [B] (round 5, "
         "back to back: 0.72–0.81): %.2f–%.2f s until the first context can take a launch, the list %.2f–%.2f s at %s–%s frames/s while "
         "contexts work on it, the rest around `main` (exec 15 ms; the kernel's teardown of the contexts' GPU state after exit)"
         % (tuple(sl[k]["process_wall_s"] for k in MODES) + tuple(slb[k]["process_wall_s"] for k in MODES)
            + (min(sl[k]["setup_s"] for k in MODES), max(sl[k]["setup_s"] for k in MODES), min(sl[k]["list_wall_s"] for k in MODES),
               max(sl[k]["list_wall_s"] for k in MODES), M(min(sl[k]["value"] for k in MODES)), M(max(sl[k]["value"] for k in MODES)))), "`sharded_list`"),
        ("the same files listed 8 × (71.6 M frames, 2.3 s of list)",
         "host front-end %s, `-E` %s, `-E -D` %s, `-F` **%s**, `-F -D` **%s** frames/s per GPU (`-F -D` / `-F` = %.3f [D], %.3f [B]; process "
         "%s–%s); what `-g 8` picks by itself (`-F -D`, eight logical GPUs on the one device: three contexts, as on any device) %s, with "
         "all 24 planned contexts forced (`as_g8_all_contexts`: the rehearsal of one process with 24 workers) %s; host ceilings %s × what "
         "eight GPUs ask of 16 cores; §7"
         % (tuple(M(wl[k]["value"]) for k in MODES) + (wl["F_D_over_F"], wlb["F_D_over_F"],
            M(min(wl[k]["process_frames_per_s"] for k in MODES[3:])), M(max(wl[k]["process_frames_per_s"] for k in MODES[3:])),
            M(wl["as_g8_default"]["value"]), M(wl["as_g8_all_contexts"]["value"]),
            " / ".join("%.2f" % wl[k]["ceiling_over_8_gpus"] for k in MODES + ("as_g8_default",)))), "`sharded_list.weak_list`"),
        ("configs[3] as `-g 8` takes it on ONE device",
         "process %.3f s [D] / %.3f [B], set-up + list %.3f / %.3f s against `-g 1 -F -D`'s %.3f / %.3f (round 5: 0.90 against 0.58); %d "
         "contexts come up (a physical device gets three at most); MLF equal"
         % (g8["process_wall_s"], g8b["process_wall_s"], g8["setup_plus_list_s"], g8b["setup_plus_list_s"], g8["g1_F_D_setup_plus_list_s"],
            g8b["g1_F_D_setup_plus_list_s"], g8["contexts"]), "`sharded_list.as_g8_on_1x_list`"),
        ("configs[4]: the four systems at once (four `phnrec -g 2` processes; one GPU: all eight logical GPUs on it, labelled oversubscribed), "
         "2500 files each, 8 / 16 kHz",
         "%.2f M frames in %.2f s (default flags) / %.2f s (`-F -D`) of whole-script wall clock = %s / %s frames/s in sum [D] (%.2f / %.2f s [B]), "
         "xRT %.1e–%.1e; the four list loops together %s–%s frames/s; every MLF equals the system's single-process run: %s"
         % (fs["default_flags"]["frames"] / 1e6, fs["default_flags"]["process_wall_s"], fs["gpu_frontend_decoder_F_D"]["process_wall_s"],
            M(fs["default_flags"]["value"]), M(fs["gpu_frontend_decoder_F_D"]["value"]), fsb["default_flags"]["process_wall_s"],
            fsb["gpu_frontend_decoder_F_D"]["process_wall_s"], min(fs[k]["xrt"] for k in ("default_flags", "gpu_frontend_decoder_F_D")),
            max(fs[k]["xrt"] for k in ("default_flags", "gpu_frontend_decoder_F_D")),
            M(min(fs[k]["list_loops_frames_per_s"] for k in ("default_flags", "gpu_frontend_decoder_F_D"))),
            M(max(fs[k]["list_loops_frames_per_s"] for k in ("default_flags", "gpu_frontend_decoder_F_D"))),
            all(fs.get("mlf_equals_single_system_run", {"x": False}).values())), "`four_systems`"),
        ("split-f16 arithmetic, opt-in (§3e)",
         "%.4f ms = %s frames/s; %.1e from the f32 kernels' output" % (D["split_f16"]["kernel_ms"], M(D["split_f16"]["value"]),
                                                                      D["split_f16"]["max_abs_vs_f32_kernels"]), "`split_f16`"),
    ]
    return "| what | figure | record |\n|---|---|---|\n" + "".join("| %s | %s | %s |\n" % x for x in rows)


def modes():
    wl = D["sharded_list"]["weak_list"]
    rows = [("host_frontend", "host front-end (`PHNREC_NO_AUTO_E=1`; round 4's `-g 1` default)"), ("gpu_energies_E", "`-E`"),
            ("gpu_energies_decoder_E_D", "`-E -D`"), ("gpu_frontend_F", "`-F` (`-g 1…3` default)"), ("gpu_frontend_decoder_F_D", "`-F -D` (`-g ≥ 4` default)"),
            ("as_g8_default", "`-g 8` without flags, eight logical GPUs on the one device (three contexts: a device gets three at most)"),
            ("as_g8_all_contexts", "the same with `PHNREC_ALL_CONTEXTS=1`: all 24 planned contexts, one process")]
    t = "| mode | frames/s per GPU (list while contexts work; process) | host CPU-s per 71.6 M frames | host ceiling | ÷ (8 × per-GPU rate) |\n|---|---|---|---|---|\n"
    for k, name in rows:
        r = wl[k]
        t += "| %s | %s; %s | %.1f | %.0f M | %.2f |\n" % (name, M(r["value"]), M(r["process_frames_per_s"]), r["host_cpu_s"],
                                                            r["host_ceiling_frames_per_s"] / 1e6, r["ceiling_over_8_gpus"])
    return t


def replace(path, name, text):
    s = open(path).read()
    a, b = "<!-- records:%s:begin -->\n" % name, "<!-- records:%s:end -->\n" % name
    if a not in s or b not in s:
        raise SystemExit("%s: markers of `%s` missing" % (path, name))
    s = s[:s.index(a) + len(a)] + text + s[s.index(b):]
    open(path, "w").write(s)


replace(os.path.join(ROOT, "DESIGN.md"), "table", records())
replace(os.path.join(ROOT, "DESIGN.md"), "modes", modes())
print("DESIGN.md tables regenerated from profiles/%s_bench*.json" % TAG)
