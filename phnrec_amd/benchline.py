"""The one stdout line of bench.py, kept small enough for the driver to parse.

bench.py builds a FULL record (every leg with its break-down and a sentence saying what it measured).  That record
goes to stderr and to a file; stdout gets `compact(full)`: the contract's keys, `roofline`, `cpu_baseline` and one or
a few NUMBERS per side leg, no prose.  What each leg measures is the table of DESIGN.md section 6.  `fit()` is the
last line of defence: if a compact record is still over the limit, whole side legs are dropped (least important
first) and named under "dropped" -- the contract keys, `roofline` and `cpu_baseline` never are.
"""
import json

LIMIT = 6144          # bytes of the stdout line (round 5's 21.7 KB line was cut by the driver: BENCH_r05.parsed = null)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "ranks")

# side legs in the order fit() gives them up
_DROP_ORDER = ("push_bunch512", "host_path", "wave_path_en", "dropin_reference_cli", "small_launches", "push_bunch5",
               "host_path_zero_copy", "wave_path", "split_f16", "single_file", "systems", "four_systems", "sharded_list")

_MODES = ("host_frontend", "gpu_energies_E", "gpu_energies_decoder_E_D", "gpu_frontend_F", "gpu_frontend_decoder_F_D",
          "as_g8_default", "as_g8_all_contexts")
_MODE_SHORT = {"host_frontend": "host", "gpu_energies_E": "E", "gpu_energies_decoder_E_D": "E_D", "gpu_frontend_F": "F",
               "gpu_frontend_decoder_F_D": "F_D", "as_g8_default": "g8_default", "as_g8_all_contexts": "g8_all_ctx"}


def _pick(d, keys):
    """the named keys of a dict that has them (numbers, booleans and short strings only travel)"""
    if not isinstance(d, dict):
        return None
    if "error" in d:
        return {"error": str(d["error"])[:120]}
    return {k: d[k] for k in keys if k in d and d[k] is not None}


def _sig(x, n=4):
    """floats to n significant digits where that loses nothing a reader of the line needs"""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, list):
        return [_sig(v, n) for v in x]
    return x


def _list_mode(m):
    return _pick(m, ("value", "process_frames_per_s", "process_wall_s", "setup_s", "list_wall_s"))


def _sharded(sl):
    if not isinstance(sl, dict) or "error" in sl:
        return _pick(sl, ())
    out = _pick(sl, ("system", "files", "frames", "gpus", "cores_usable", "frames_per_s", "process_frames_per_s"))
    for k in _MODES:
        if k in sl:
            out[_MODE_SHORT[k]] = _list_mode(sl[k])
    eq = [v for k, v in sl.items() if k.startswith("mlf_") and isinstance(v, bool)]
    if eq:
        out["mlf_all_modes_equal"] = all(eq)
    hc = sl.get("host_ceiling")
    if isinstance(hc, dict):
        out["host_ceiling"] = _pick(hc, ("frames_per_s", "decoder_only_frames_per_s"))
    wl = sl.get("weak_list")
    if isinstance(wl, dict):
        w = _pick(wl, ("files", "frames", "F_D_over_F"))
        for k in _MODES:
            if k in wl:
                w[_MODE_SHORT[k]] = _pick(wl[k], ("value", "ceiling_over_8_gpus"))
                if k.startswith("as_g8"):
                    w[_MODE_SHORT[k]].update(_pick(wl[k], ("process_frames_per_s", "contexts", "mode")))
        out["weak_list"] = w
    cz = sl.get("cz_same_list")
    if isinstance(cz, dict):
        out["cz_same_list"] = {_MODE_SHORT.get(k, k): _pick(v, ("value", "process_frames_per_s"))
                               for k, v in cz.items() if isinstance(v, dict)}
    g8 = sl.get("as_g8_on_1x_list")
    if isinstance(g8, dict):
        out["as_g8_on_1x_list"] = _pick(g8, ("value", "process_frames_per_s", "setup_s", "list_wall_s", "process_wall_s",
                                             "setup_plus_list_s", "g1_F_D_setup_plus_list_s", "g1_F_D_process_wall_s",
                                             "contexts", "mode", "mlf_equals_g1"))
    return out


def _four(fs):
    if not isinstance(fs, dict) or "error" in fs:
        return _pick(fs, ())
    out = _pick(fs, ("files_per_system", "oversubscribed", "gpu_pairs"))
    for k in ("default_flags", "gpu_frontend_decoder_F_D"):
        if k in fs:
            out["F_D" if k != "default_flags" else k] = _pick(fs[k], ("value", "process_wall_s", "xrt",
                                                                      "list_loops_frames_per_s"))
    eq = fs.get("mlf_equals_single_system_run")
    if isinstance(eq, dict):
        out["mlf_equal"] = all(eq.values())
    if "mlf_check_error" in fs:
        out["mlf_check_error"] = str(fs["mlf_check_error"])[:160]
    return out


def _cpu(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ("value", "unit", "cores", "kind", "variant", "sample", "parity_max_abs_vs_gpu",
                    "parity_max_abs_vs_gpu_split_f16"))
    for k in ("sgemm_1core", "sgemm_all_cores"):
        if k in c:
            out[k] = _pick(c[k], ("value", "cores"))
    p = c.get("port")
    if isinstance(p, dict):
        out["port"] = _pick(p, ("value", "cores", "kind", "parity_max_abs_vs_gpu"))
        if isinstance(p.get("all_cores"), dict):
            out["port"]["all_cores"] = _pick(p["all_cores"], ("value", "cores"))
    h = c.get("host") or (p or {}).get("host")
    if isinstance(h, dict):
        out["host"] = _pick(h, ("cpu_model", "cores_visible", "cores_usable"))
    return out


def _roofline(r):
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "flop_per_frame", "hbm_gbps"))
    if isinstance(r.get("traffic_source"), str):
        out["traffic_source"] = r["traffic_source"].split(" ")[0]
    if isinstance(r.get("cold"), dict):
        out["cold"] = _pick(r["cold"], ("launches", "kernel_ms", "frac"))
    if isinstance(r.get("windows"), dict):
        out["windows"] = _pick(r["windows"], ("kernel_ms_median", "kernel_ms_min", "kernel_ms_max"))
    return out


def compact(full):
    """the stdout form of a full bench record"""
    out = {k: full[k] for k in CONTRACT if k in full}
    if isinstance(out.get("config"), dict):
        out["config"] = _pick(out["config"], ("workload", "weights", "kernel", "cli_processes_timed"))
        out["config"]["arithmetic"] = "f32 MFMA 16x16x4, f32 accumulate"
        out["config"]["sharding"] = "replica per GPU, no collective"
    for k in ("preheat_launches", "ms_per_step_before_closing_barrier", "frames_per_s_per_gpu", "xrt", "stub",
              "frames_all_ranks"):
        if k in full:
            out[k] = full[k]
    if "roofline" in full:
        out["roofline"] = _roofline(full["roofline"])
    if "cpu_baseline" in full:
        out["cpu_baseline"] = _cpu(full["cpu_baseline"])
    for k in ("host_path", "host_path_zero_copy", "wave_path", "wave_path_en"):
        if k in full:
            out[k] = _pick(full[k], ("value", "ms_per_call", "frames", "rows_sum_to_one"))
    for k in ("push_bunch5", "push_bunch512"):
        if k in full:
            out[k] = _pick(full[k], ("value", "us_per_call"))
    for k in ("small_launches", "systems"):
        if isinstance(full.get(k), dict):
            out[k] = {n: _pick(v, ("kernel_ms", "frac")) for n, v in full[k].items() if isinstance(v, dict)}
    if "split_f16" in full:
        out["split_f16"] = _pick(full["split_f16"], ("value", "kernel_ms", "speedup_vs_f32_kernel",
                                                     "max_abs_vs_f32_kernels", "rows_sum_to_one"))
    sf = full.get("single_file")
    if isinstance(sf, dict):
        out["single_file"] = {k: _pick(v, ("process_wall_s", "min_process_wall_s", "gpu_create_s"))
                              for k, v in sf.items() if isinstance(v, dict)}
    if "dropin_reference_cli" in full:
        out["dropin_reference_cli"] = _pick(full["dropin_reference_cli"], ("value", "files", "frames"))
    if "sharded_list" in full:
        out["sharded_list"] = _sharded(full["sharded_list"])
    if "four_systems" in full:
        out["four_systems"] = _four(full["four_systems"])
    if "detail" in full:
        out["detail"] = full["detail"]
    head = {k: out[k] for k in CONTRACT if k in out}           # the contract's keys travel as they are
    return {**_sig(out, 4), **head}


def _dumps(rec):
    return json.dumps(rec, separators=(",", ":"))


def fit(rec, limit=LIMIT):
    """rec, with side legs dropped (and named) until its JSON line is at most `limit` bytes"""
    rec = dict(rec)
    dropped = []
    for k in _DROP_ORDER:
        if len(_dumps(rec)) + 1 <= limit:
            break
        if k in rec:
            del rec[k]
            dropped.append(k)
            rec["dropped"] = dropped
    return rec


def stdout_line(full, limit=LIMIT):
    line = _dumps(fit(compact(full), limit))
    if len(line) + 1 > limit:
        raise RuntimeError("bench line is %d bytes with every side leg dropped (limit %d)" % (len(line) + 1, limit))
    return line
