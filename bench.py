#!/usr/bin/env python3
"""bench.py -- LCRC posterior path throughput on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (window assembly + window/DCT projection +
band-L, band-R and merger MLPs, fused in one HIP kernel) over one batch of
8192 frames of the PHN_CZ_SPDAT_LCRC_N1500 system, input log-mel frames already
resident in HBM, output posteriors left in HBM.  N GPUs = N replicas, each on
its own batch (utterances shard with no exchange; weak scaling).

Prints ONE JSON line on rank 0.  `roofline` prices the fused kernel against the
f32 MFMA peak with the ALGORITHMIC flop count (unpadded 2*MAC of the three MLPs,
3.060 MFLOP per CZ frame) and HIP-event kernel times taken on the
launch stream inside the timed region.  `cpu_baseline` times the reference's own
code (oracle/_ref, built from /root/reference in the build container) on this
box's host cores on a bounded sample, and reports the GPU-vs-CPU parity on it.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SYSTEM = "PHN_CZ_SPDAT_LCRC_N1500"
BATCH = 8192
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
_RECORD_OUT = sys.stdout         # main() replaces it with a private copy of the original stdout


def algorithmic_flops_per_frame(dims):
    """2*MAC of the three MLPs, unpadded."""
    return sum(2 * (i * h + h * o) for (i, h, o) in dims)


def model_directory(tmp):
    from phnrec_amd import modelgen
    real = os.path.join(ROOT, "tests", "golden", "models", SYSTEM)
    if os.path.isdir(real):
        return real, "shipped %s .nbin weights" % SYSTEM
    d = os.path.join(tmp, SYSTEM)
    modelgen.write_system(d, SYSTEM, seed=1234)
    return d, "seeded random weights of the %s shape" % SYSTEM


def usable_cpus():
    """CPUs this process may use: the affinity mask capped by the cgroup quota (the GPU boxes show all
    host cores but grant a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_scaling(mdir, nbanks, seconds):
    """The reference code in its sgemm regime (bunch_size=512), on one core and on every usable core
    (one process per core, each on its own utterance: the reference is not thread-safe)."""
    import subprocess
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    res = {}
    for label, procs in (("sgemm_1core", 1), ("sgemm_all_cores", usable_cpus())):
        cmd = lambda i: [sys.executable, worker, mdir, str(nbanks), "512", "1", str(seconds), str(100 + i), "2048"]
        try:
            ps = [subprocess.Popen(cmd(i), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                  for i in range(procs)]
            outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in ps]
        except Exception as e:      # the baseline is reported when it can be measured, never fatal
            res[label] = {"error": repr(e)}
            continue
        res[label] = {"value": round(sum(o["frames"] for o in outs) / max(o["seconds"] for o in outs), 1),
                      "unit": "frames/s", "cores": procs, "kind": "reference",
                      "variant": "USE_BLAS (MKL cblas_sgemm), bunch_size=512, %d process(es)" % procs,
                      "sample": "%d frames in %.1f s" % (sum(o["frames"] for o in outs),
                                                        max(o["seconds"] for o in outs))}
    return res


def cpu_baseline(mdir, nbanks, mel, gpu_post, budget_s, gpu_post_split=None):
    """The reference CPU path on this host, bounded to ~budget_s seconds per variant.  gpu_post_split: the same batch
    from the split-f16 kernels, compared with the same reference rows."""
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import binding as ob
    out = None
    chunk = 512
    for blas in (True, False):
        if ob.ref_lib_path(blas) is None:
            continue
        try:
            t = ob.RefTraps(mdir, nbanks, bunch=5, blas=blas)
        except OSError:
            continue
        done, work, t0, worst, worst_split = 0, 0, time.perf_counter(), 0.0, 0.0
        while time.perf_counter() - t0 < budget_s:
            # chunk with a 15-frame halo on each side == rows of the whole utterance;
            # wraps around the batch until the time budget is used
            pos = done % (mel.shape[0] - chunk + 1)
            a, b = max(0, pos - 15), min(mel.shape[0], pos + chunk + 15)
            post = t.process_offline(mel[a:b])[pos - a:pos - a + chunk]
            worst = max(worst, float(np.abs(post - gpu_post[pos:pos + chunk]).max()))
            if gpu_post_split is not None:
                worst_split = max(worst_split, float(np.abs(post - gpu_post_split[pos:pos + chunk]).max()))
            done += chunk
            work += b - a          # halo frames are real work for the CPU too
        dt = time.perf_counter() - t0
        out = {"value": round(work / dt, 1), "unit": "frames/s", "cores": 1, "kind": "reference",
               "variant": ("USE_BLAS (MKL cblas_sgemv, shipped bunch_size=5)" if blas
                           else "naive loop (no BLAS), bunch_size=5"),
               "sample": "%d frames of the bench batch (512-frame chunks + 15-frame halos), %.1f s" % (work, dt),
               "parity_max_abs_vs_gpu": worst}
        if gpu_post_split is not None:
            out["parity_max_abs_vs_gpu_split_f16"] = worst_split
        if blas:
            out.update(reference_scaling(mdir, nbanks, min(budget_s, 4.0)))
        break
    # this repo's port, single thread and all cores (frames split over threads)
    o = ob.Oracle(mdir, nbanks)
    n1 = min(mel.shape[0], 2048)
    t0 = time.perf_counter()
    p1 = o.posteriors(mel[:n1 + 15])[:n1]
    dt1 = time.perf_counter() - t0
    cores = usable_cpus()
    nall, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(budget_s, 5.0):
        o.posteriors(mel, threads=cores)
        nall += mel.shape[0]
    dta = time.perf_counter() - t0
    port = {"value": round(n1 / dt1, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "first %d frames, %.1f s" % (n1, dt1),
            "parity_max_abs_vs_gpu": float(np.abs(p1 - gpu_post[:n1]).max()),
            **({"parity_max_abs_vs_gpu_split_f16": float(np.abs(p1 - gpu_post_split[:n1]).max())}
               if gpu_post_split is not None else {}),
            "all_cores": {"value": round(nall / dta, 1), "cores": cores, "cores_visible": os.cpu_count(),
                          "sample": "%d frames, %.1f s" % (nall, dta)}}
    host = {"cpu_model": cpu_model(), "cores_visible": os.cpu_count(), "cores_usable": cores}
    if out is None:
        port["host"] = host
        return port
    out["port"] = port
    out["host"] = host
    return out


def time_launches(ctx, stream, d_mel, d_post, n, reps):
    """average ms per launch of `reps` launches of n frames, one HIP event pair on the launch stream"""
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ctx.posteriors_device(d_mel.data_ptr(), n, d_post.data_ptr(), stream=stream.cuda_stream)
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps


PEAK_F16_MFMA_TFLOPS = 2516.6      # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF)


def split_f16_leg(capi, ctx, stream, d_mel, d_post, n, flops_frame, f32_post, f32_ms):
    """The same workload on the split-f16 kernels (lcrc_set_arithmetic(LCRC_ARITH_SPLIT_F16), opt-in): every f32
    product as three exact f16 x f16 MFMA products accumulated in f32.  `value` of the line stays the f32-MFMA
    kernel named by north_star; this leg says what the other arithmetic buys on the same inputs and how far its
    posteriors are from the f32 kernels' (both are compared with the reference in cpu_baseline)."""
    import torch
    ctx.set_arithmetic(capi.ARITH_SPLIT_F16)
    try:
        time_launches(ctx, stream, d_mel, d_post, n, 150)          # its own pre-heat (another clock point)
        ms = time_launches(ctx, stream, d_mel, d_post, n, 200)
        post = d_post.clone()
    finally:
        ctx.set_arithmetic(capi.ARITH_F32)
    alg = n * flops_frame / (ms * 1e-3) / 1e12
    # its own bound: every workgroup (32 frames) streams the whole fragment set through its CU's vector-memory path
    frag = 0
    for i in range(3):
        k, h, o = ctx.net_dims(i)
        pairs, ns, n_ot = (h + 31) // 32, (k + 31) // 32, (o + 15) // 16
        frag += pairs * (2 * ns + n_ot) * 2048
    wgs = (n + 31) // 32
    vmem_peak = 64 * 256 * 2.4e9                     # 64 B/clk per CU, 256 CUs, nominal clock
    leg = {
           "vmem_path": {"bound": "vector-memory path of the CUs (L1/TA, 64 B/clk each)", "bytes_per_workgroup": frag,
                         "achieved": round(frag * wgs / (ms * 1e-3) / 1e12, 2), "peak": round(vmem_peak / 1e12, 1),
                         "unit": "TB/s", "frac": round(frag * wgs / (ms * 1e-3) / vmem_peak, 3)},"value": round(n / ms * 1e3, 1), "unit": "frames/s", "kernel_ms": round(ms, 4),
           "speedup_vs_f32_kernel": round(f32_ms / ms, 3),
           "algorithmic_tflops": round(alg, 1),
           "speed_vs_f32_mfma_peak": round(alg / PEAK_F32_MFMA_TFLOPS, 3),   # a ratio of rates, not a roofline fraction
           # three f16 products per f32 product: the MFMA work actually executed against the dense f16 peak
           "frac_of_f16_mfma_peak_3_products": round(3 * alg / PEAK_F16_MFMA_TFLOPS, 3),
           "bound": "the CU's vector-memory path (64 B/clk): 32 frames per workgroup stream 6.5 MB of weight fragments "
                    "through every CU per launch (DESIGN.md)",
           "max_abs_vs_f32_kernels": float((post - f32_post).abs().max().item()),
           "rows_sum_to_one": bool((post.sum(dim=1) - 1).abs().max().item() < 1e-5),
           "what": "lcrc_set_arithmetic(LCRC_ARITH_SPLIT_F16): f32 operands as (high, low) f16 pairs, products as "
                   "v_mfma_f32_16x16x32_f16 x 3, f32 accumulation; same launches, inputs and outputs as `value`"}
    return leg, post


PREHEAT_MS = 150.0      # time-based pre-heat in front of every timed small-launch / systems entry (stated in the line)
WINDOWS, PER_WINDOW = 7, 40


def heat(ctx, stream, d_mel, d_post, n, ms=PREHEAT_MS):
    """back-to-back launches of n frames for at least `ms` milliseconds of device time (the clock follows LOAD TIME, not a
    launch count: 600 launches of a 0.034 ms kernel are 20 ms, and the legs in between leave the device idle)"""
    import torch
    done, t0 = 0, time.perf_counter()
    while True:
        for _ in range(50):
            ctx.posteriors_device(d_mel.data_ptr(), n, d_post.data_ptr(), stream=stream.cuda_stream)
        done += 50
        stream.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= ms:
            return done


def time_windows(ctx, stream, d_mel, d_post, n, windows=WINDOWS, per=PER_WINDOW):
    """`windows` consecutive windows of `per` launches, all enqueued back to back (no host sync in between: the device
    never idles), one HIP event between windows on the launch stream -> ms per launch of every window"""
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    ev[0].record(stream)
    for w in range(windows):
        for _ in range(per):
            ctx.posteriors_device(d_mel.data_ptr(), n, d_post.data_ptr(), stream=stream.cuda_stream)
        ev[w + 1].record(stream)
    stream.synchronize()
    return [ev[w].elapsed_time(ev[w + 1]) / per for w in range(windows)]


def launch_entry(ctx, stream, d_mel, d_post, n, fpf):
    """one entry of `small_launches` / `systems`: time-based pre-heat, then median / min / max over the windows"""
    heated = heat(ctx, stream, d_mel, d_post, n)
    ws = sorted(time_windows(ctx, stream, d_mel, d_post, n))
    ms = ws[len(ws) // 2]
    frac = lambda t: round(n * fpf / (t * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    return {"kernel_ms": round(ms, 4), "kernel_ms_min": round(ws[0], 4), "kernel_ms_max": round(ws[-1], 4),
            "frames_per_s": round(n / ms * 1e3, 1), "frac": frac(ms), "frac_min": frac(ws[-1]), "frac_max": frac(ws[0]),
            "flops_per_frame": fpf, "kernel": ctx.kernel_name,
            "windows": "%d x %d launches back to back, median (min / max beside it)" % (len(ws), PER_WINDOW),
            "preheat": "%d launches = >= %.0f ms of back-to-back load of this size" % (heated, PREHEAT_MS)}


def small_launch_legs(capi, modelgen, dev, stream):
    """The small-launch regime, beside the headline: roofline fraction of 2048- and 4096-frame launches (the
    launcher picks 16-frame workgroups and, below half of the CUs, the split-hidden kernels), for CZ and for EN
    at its own BASELINE size (configs[1]); every shipped system at the headline's 8192 frames; and -- LAST, they leave
    the device idle between calls -- the streaming entry (Traps::CalcFeaturesBunched semantics, traps.cpp:518-535) at the
    shipped bunch_size=5 (PHN_*/config:11) and at 512."""
    import ctypes as C
    import torch
    out = {}
    # sizes per system: CZ and EN in the small-launch regime (EN 4096 = configs[1]); HU (configs[3]'s system), RU and EN
    # also at the headline's 8192 frames, so that every shipped system's roofline fraction is in the driver-run line
    sizes = {"cz": (2048, 4096), "en": (2048, 4096, 8192), "hu": (8192,), "ru": (8192,)}
    # EN first: its entries are the ones the previous round's driver run disagreed about
    for system in ("PHN_EN_TIMIT_LCRC_N500", "PHN_CZ_SPDAT_LCRC_N1500", "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500"):
        mdir = os.path.join(ROOT, "tests", "golden", "models", system)
        if not os.path.isdir(mdir):
            continue
        spec = modelgen.SYSTEMS[system]
        nb = spec["nbanks"]
        ctx = capi.Lcrc(mdir, nb, device=dev.index)
        ctx.set_timing(False)
        fpf = algorithmic_flops_per_frame([ctx.net_dims(i) for i in range(3)])
        tag = system.split("_")[1].lower()
        n_max = max(sizes[tag])
        d_mel = torch.from_numpy(modelgen.synth_mel(n_max, nb, seed=7, mean_norm=spec["sent_mean_norm"])).to(dev)
        d_post = torch.empty((n_max, ctx.n_out), dtype=torch.float32, device=dev)
        for n in sizes[tag]:
            out["%s_%d" % (tag, n)] = launch_entry(ctx, stream, d_mel, d_post, n, fpf)
        ctx.close()
    mdir = os.path.join(ROOT, "tests", "golden", "models", SYSTEM)
    if os.path.isdir(mdir):
        nb = modelgen.SYSTEMS[SYSTEM]["nbanks"]
        ctx = capi.Lcrc(mdir, nb, device=dev.index)
        ctx.set_timing(False)
        # streaming: raw ctypes calls on preallocated buffers (what a C caller pays), wall clock
        L, h = ctx.L, ctx.h
        push = L.lcrc_push
        push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        for bunch in (5, 512):
            mel = modelgen.synth_mel(bunch * 64, nb, seed=11)
            post = np.empty((bunch, ctx.n_out), np.float32)
            ptrs = [mel[i * bunch:].ctypes.data for i in range(64)]
            ctx.reset()
            push(h, mel.ctypes.data, 15, None, 0)
            for i in range(32):
                push(h, ptrs[i], bunch, post.ctypes.data, 1)
            calls, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                for i in range(64):
                    if push(h, ptrs[i], bunch, post.ctypes.data, 1) != 0:
                        raise RuntimeError("lcrc_push failed")
                calls += 64
            dt = time.perf_counter() - t0
            out["push_bunch%d" % bunch] = {"value": round(calls * bunch / dt, 1), "unit": "frames/s",
                                           "us_per_call": round(dt / calls * 1e6, 2),
                                           "what": "lcrc_reset/lcrc_push(n=%d, needed=1): host frames in, host "
                                                   "posteriors out, synchronous (PCIe-inclusive)" % bunch}
        push.argtypes = [C.c_void_p, capi._f32p, C.c_int, C.c_void_p, C.c_int]
        ctx.close()
    return out


def alaw_encode(x):
    """A-law bytes by nearest-value search in the 256-entry expansion table (srec.cpp:769
    decodes as 8 * table[b])"""
    def expand(b):
        a = b ^ 0x55
        mant, seg = (a & 0x0F) << 4, (a & 0x70) >> 4
        mant = mant + 8 if seg == 0 else (mant + 0x108) << (seg - 1)
        return float(mant if a & 0x80 else -mant)
    table = np.array([expand(b) for b in range(256)], np.float32)
    order = np.argsort(table)
    srt = table[order]
    idx = np.clip(np.searchsorted(srt, x), 1, 255)
    pick = np.where(np.abs(x - srt[idx - 1]) <= np.abs(srt[idx] - x), idx - 1, idx)
    return order[pick].astype(np.uint8)


def config2_alaw_signal(frames=BATCH):
    """BASELINE configs[2]'s input as SURVEY.md 8(d) cfg3 defines it: 8 kHz, (frames-1)*80+200 samples,
    0.3 full-scale mix of 5 sines (200-3400 Hz) + Gaussian noise sigma 1000, seed 1235, A-law"""
    n = (frames - 1) * 80 + 200
    rng = np.random.default_rng(1235)
    t = np.arange(n) / 8000.0
    sig = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, n)
    return alaw_encode(np.clip(sig, -32768, 32767).astype(np.float32)).tobytes()


def config1_lin16_signal(frames=4096):
    """BASELINE configs[1]'s input as SURVEY.md 8(d) cfg2 defines it: 16 kHz lin16 mono, (frames-1)*160+400 samples,
    0.3 full-scale mix of 5 sines (200-3400 Hz) + Gaussian noise sigma 1000, clipped to int16, seed 1234"""
    n = (frames - 1) * 160 + 400
    rng = np.random.default_rng(1234)
    t = np.arange(n) / 16000.0
    sig = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, n)
    return np.clip(sig, -32768, 32767).astype("<i2").tobytes()


def wave_path_leg(capi, mdir, nb, gpu, raw=None, wave_format="alaw", sent_mean_norm=True, what=None):
    """bytes in -> posteriors out through the GPU front-end (lcrc_wave_to_posteriors), PCIe-inclusive"""
    raw = config2_alaw_signal() if raw is None else raw
    ctx = capi.Lcrc(mdir, nb, device=gpu)
    spec = {}
    if wave_format == "lin16" and nb == 23:       # the EN system's front-end (PHN_EN_TIMIT_LCRC_N500/config)
        spec = dict(sample_freq=16000, vector_size=400, vector_step=160, lower_freq=0.0, higher_freq=8000.0)
    ctx.configure_frontend(wave_format=wave_format, sent_mean_norm=sent_mean_norm, **spec)
    frames = ctx.frontend_frames(len(raw))
    # the C entry point on the caller's own, reused buffers (what a C caller pays)
    rawb = np.frombuffer(raw + b"\0", dtype=np.uint8).copy()
    off = np.array([0, len(raw)], np.int64)
    post = np.empty((frames, ctx.n_out), np.float32)
    foff = np.zeros(2, np.int32)
    out = {}
    # sentence mean in the reference's sequential order (the library's default) and as the opt-in tree
    for key, order in (("value", 1), ("tree_mean_value", 0)) if sent_mean_norm else (("value", 1),):
        ctx._check(ctx.L.lcrc_set_mean_order(ctx.h, order))
        # (50 untimed calls first, ~25 ms of load: the legs before this one leave the device idle for long enough that it
        #  clocks down, and 30 calls straight after 3 warm-up calls measured its ramp -- 0.50 to 0.88 ms on the same box)
        for _ in range(50):
            ctx._check(ctx.L.lcrc_wave_to_posteriors(ctx.h, rawb, off, 1, post, foff))
        ts = []                                  # median of single calls: one hiccup of the host does not move it
        for _ in range(40):
            t0 = time.perf_counter()
            ctx.L.lcrc_wave_to_posteriors(ctx.h, rawb, off, 1, post, foff)
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        out[key] = round(frames / dt, 1)
        out["ms_per_call" if key == "value" else "tree_mean_ms_per_call"] = round(dt * 1e3, 4)
    ok = bool(np.abs(post.sum(axis=1) - 1).max() < 1e-5)
    ctx.close()
    out.update({"unit": "frames/s", "frames": frames, "bytes_in": len(raw), "rows_sum_to_one": ok,
                "what": what or "configs[2] input per SURVEY 8(d) cfg3 (A-law, 5 sines + noise, seed 1235): "
                                "lcrc_wave_to_posteriors(): A-law decode + mel bank + sentence mean norm (reference order; "
                                "tree_mean_*: lcrc_set_mean_order(0)) + posteriors on the GPU, host bytes in, host posteriors "
                                "out (reused buffers), synchronous"})
    return out


def synthetic_list(td, n_files, seed=1236, rate=8000):
    """SURVEY 8(d) cfg4's list: raw lin16 files of 3-15 s (slices of one synthetic signal); returns names, frames"""
    rng = np.random.default_rng(seed)
    n_base = 16 * rate
    t = np.arange(n_base) / float(rate)
    base = sum(0.06 * 32767 * np.sin(2 * np.pi * f * t) for f in (200, 700, 1300, 2100, 3400)) + rng.normal(0, 1000, n_base)
    base = np.clip(base, -32768, 32767).astype("<i2")
    names, frames = [], 0
    vs, step = rate // 40, rate // 100
    for i in range(n_files):
        n = int(rng.uniform(3.0, 15.0) * rate)
        o = int(rng.integers(0, n_base - n))
        p = os.path.join(td, "f%05d.raw" % i)
        base[o:o + n].tofile(p)
        names.append(p)
        frames += (n - vs) // step + 1
    lst = os.path.join(td, "list.scp")
    with open(lst, "w") as f:
        f.write("".join(n + "\n" for n in names))
    return lst, names, frames


# A process's GPU state is torn down by the kernel AFTER it has exited, and a process that starts meanwhile waits for that
# inside hipInit: 150-250 ms instead of 50 ms when HIP processes follow each other without a pause, 50 ms from a quarter
# of a second on (profiles/r06_ab_runs.txt 5, tools/exit_probe.py).  Every timed CLI process of this file therefore starts
# SETTLE_S after the previous one has exited -- what a caller who runs ONE command sees -- and the pause is in no figure.
SETTLE_S = 0.3


def run_cli(exe, args, env, timeout=600):
    """one CLI run: wall clock of the process and the figures of its PHNREC_STATS line"""
    import subprocess
    time.sleep(SETTLE_S)
    t0 = time.perf_counter()
    pr = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=timeout)
    wall = time.perf_counter() - t0
    stats = [ln for ln in pr.stderr.splitlines() if ln.startswith("phnrec: files=")]
    if pr.returncode != 0 or not stats:
        return {"error": "rc=%d %s" % (pr.returncode, pr.stderr.strip()[-300:])}, pr
    kv = dict(tok.split("=", 1) for tok in stats[-1].replace("(", "").replace(")", "").split() if "=" in tok)
    # Since round 6 the contexts come up BESIDE the list: the CLI's wall_s runs from the list's first line to its last and
    # contains first_ctx_s, the time until the first context could take a launch (HIP start-up, model load, upload: 0.2-0.3 s
    # of runtime floor on these boxes).  `value` stays what it was in earlier rounds -- the rate of the list loop while
    # contexts work on it: frames / (wall_s - first_ctx_s) --, `setup_s` what lies in front of the first launch, and
    # `process_frames_per_s` (exec to exit) is the figure a caller of the CLI sees.
    first_ctx = float(kv.get("first_ctx_s", 0))
    loop_s = max(1e-6, float(kv["wall_s"]) - first_ctx)
    return {"value": round(float(kv["frames"]) / loop_s, 1), "unit": "frames/s", "list_wall_s": round(loop_s, 3),
            "list_from_first_line_s": float(kv["wall_s"]), "first_ctx_s": first_ctx, "contexts": int(kv.get("contexts", 0)),
            "process_wall_s": round(wall, 3), "xrt": float(kv["xRT"]), "gpu_kernel_ms": float(kv["gpu_kernel_ms"]),
            "setup_s": round(float(kv["setup_s"]) + first_ctx, 3), "create_s": float(kv.get("create_s", 0)),
            "first_launch_s": float(kv.get("first_launch_s", 0)), "main_s": float(kv.get("main_s", 0)),
            "host_cpu_s": float(kv.get("host_cpu_s", 0)), "host_threads": int(kv.get("host_threads", 0)),
            "cpu_s_by_stage": {k: float(kv[k]) for k in ("stage1", "read", "gather", "decode_write", "viterbi") if k in kv},
            # the whole process (exec to exit: start-up, model load, list loop) -- what a caller of the CLI waits for
            "process_frames_per_s": round(float(kv["frames"]) / wall, 1), "mode": kv.get("mode"),
            "max_rss_mb": float(kv.get("max_rss_mb", 0))}, pr


def single_file_leg(mdir, gpu):
    """The reference's own smoke test (test.sh): `phnrec -c DIR -i test.raw -o test.rec` (and -t post) as a PROCESS,
    wall clock from exec to exit, beside the reference's MKL build on the same file in the same run.  For one 7.5 s
    file the GPU's start-up (HIP runtime, code objects, weight upload) is the whole cost; the break-down says where."""
    import subprocess
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    raw = os.path.join(ROOT, "tests", "golden", "test.raw")
    if not os.path.exists(exe) or not os.path.exists(raw):
        return None
    env = dict(os.environ, PHNREC_STATS="1", PHNREC_DEVICE_MAP=str(gpu))
    out = {"file": "tests/golden/test.raw (7.5 s, 747 frames)",
           "what": "process wall clock, median / min of 5 runs after one discarded run, each started %.1f s after the previous "
                   "process had exited; break-down from PHNREC_STATS and LCRC_TRACE_STARTUP of the median run" % SETTLE_S}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for key, extra in (("str", ["-o", os.path.join(td, "x.rec")]), ("post", ["-t", "post", "-o", os.path.join(td, "x.lop")])):
            runs = []
            for i in range(6):
                r, pr = run_cli(exe, ["-c", mdir, "-i", raw] + extra, dict(env, LCRC_TRACE_STARTUP="1"), timeout=120)
                if "error" in r:
                    out[key] = r
                    break
                r["trace"] = {ln.split(":", 1)[1].rsplit(None, 2)[0].strip(): float(ln.split()[-2])
                              for ln in pr.stderr.splitlines() if ln.startswith("lcrc startup:")}
                if i > 0:
                    runs.append(r)
            else:
                runs.sort(key=lambda r: r["process_wall_s"])
                med = runs[len(runs) // 2]
                out[key] = {"process_wall_s": med["process_wall_s"], "min_process_wall_s": runs[0]["process_wall_s"],
                            "main_s": med["main_s"], "gpu_create_s": med["create_s"], "first_launch_s": med["first_launch_s"],
                            "list_wall_s": med["list_wall_s"], "kernel_ms": med["gpu_kernel_ms"], "create_trace_ms": med["trace"]}
        from oracle import binding as ob
        for key, blas in (("reference_cpu_mkl", True), ("reference_cpu_naive", False)):
            ref = ob.ref_cli_path(blas)
            if ref is None:
                continue
            renv = dict(os.environ, MKL_NUM_THREADS="1", MKL_THREADING_LAYER="SEQUENTIAL")
            walls = []
            try:
                for i in range(6):
                    t0 = time.perf_counter()
                    subprocess.run([ref, "-c", mdir, "-i", raw, "-o", os.path.join(td, "r.rec")], env=renv, check=True,
                                   capture_output=True, timeout=120)
                    if i > 0:
                        walls.append(time.perf_counter() - t0)
                walls.sort()
                out[key] = {"process_wall_s": round(walls[len(walls) // 2], 3), "min_process_wall_s": round(walls[0], 3),
                            "kind": "reference", "cores": 1}
            except Exception as e:
                out[key] = {"error": repr(e)}
    return out


HU = "PHN_HU_SPDAT_LCRC_N1500"
# the CLI's roads over a list: host front-end; -E (mel energies on the GPU, ln() and the normalisations on the host: the host
# front-end's bits); -F (whole front-end on the GPU); each with the decoder on the host or on the GPU (-D)
LIST_MODES = (("host_frontend", []), ("gpu_energies_E", ["-E"]), ("gpu_energies_decoder_E_D", ["-E", "-D"]),
              ("gpu_frontend_F", ["-F"]), ("gpu_frontend_decoder_F_D", ["-F", "-D"]))


def sharded_list_leg(n_gpus, dmap, n_files):
    """BASELINE configs[3]: the HU system on a 10 000-file synthetic list (seed 1236), `phnrec -g N`
    over the ranks' GPUs -> posteriors + host Viterbi -> MLF.  The CLI shards by pulling launches of consecutive files
    from one queue (no collective).  host_ceiling: what this host's cores can do on the same list with the GPUs idle --
    reading the files, and the host Viterbi on posterior dumps -- so that a flat curve over N can be read for what it is."""
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    mdir = os.path.join(ROOT, "tests", "golden", "models", HU)
    if not os.path.exists(exe) or not os.path.isdir(mdir):
        return {"error": "CLI or %s model directory missing" % HU}
    cores = usable_cpus()
    out = {"system": HU, "files": n_files, "gpus": n_gpus, "device_map": dmap, "cores_usable": cores,
           "what": "phnrec -c HU -l list -m out.mlf -g N [-E | -E -D | -F | -F -D] (PHNREC_DEVICE_MAP = the ranks' GPUs): raw lin16 8 kHz files of "
                   "3-15 s -> MLF on disk; frames/s of the list loop as the CLI reports it (process start-up and model load "
                   "excluded; process_wall_s includes them)"}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        lst, names, frames = synthetic_list(td, n_files)
        out["frames"] = frames
        # (from two GPUs on the CLI called without flags takes the GPU front-end by itself -- -F where its ln() is this host's
        #  libm's own sequence, else -E; the same output bytes either way: `host_frontend` then IS that road ("mode" says
        #  which); PHNREC_NO_AUTO_E=1 would keep the pure host front-end)
        out["host_frontend_takes_gpu_road"] = n_gpus >= 2
        out["default_flags_on_one_gpu"] = "a list of ~100 files or more takes -F by itself (mode F,auto): see gpu_frontend_F"
        # (and from four GPUs on a list that ends in labels decodes on the GPUs by itself -- bit-identical labels, the
        #  Viterbi off the host's cores: every mode below is then `... -D`; PHNREC_NO_AUTO_D=1 would keep the host decoder)
        out["every_mode_decodes_on_the_gpu"] = n_gpus >= 4
        env = dict(os.environ, PHNREC_STATS="1", PHNREC_DEVICE_MAP=",".join(str(d) for d in dmap))
        mlfs = {}
        # (on ONE GPU a list of this length called without flags takes the GPU front-end by itself too -- `gpu_frontend_F` is
        #  that road; `host_frontend` there is run with PHNREC_NO_AUTO_E=1 so that it keeps measuring the host front-end)
        host_env = dict(env, PHNREC_NO_AUTO_E="1") if n_gpus == 1 else env
        for key, extra in LIST_MODES:
            mlf = os.path.join(td, key + ".mlf")
            try:
                best = None
                for _ in range(2):                    # the better of two runs (the first also warms the page cache)
                    r, _pr = run_cli(exe, ["-c", mdir, "-l", lst, "-m", mlf, "-g", str(n_gpus)] + extra,
                                     host_env if key == "host_frontend" else env)
                    if "error" in r or best is None or r["value"] > best["value"]:
                        best = r
                    if "error" in r:
                        break
                out[key] = best
                mlfs[key] = mlf
            except Exception as e:
                out[key] = {"error": repr(e)}
        if isinstance(out.get("gpu_frontend_F"), dict) and "value" in out["gpu_frontend_F"]:
            out["frames_per_s"] = out["gpu_frontend_F"]["value"]
            out["process_frames_per_s"] = out["gpu_frontend_F"]["process_frames_per_s"]
        # N GPUs finish the 10 000 files in 0.3 s / N of list loop behind 0.45 s of process start-up: at N = 8 the loop is
        # 40 ms, most of it the ramp until every context has its first launch -- nothing a scaling curve, or a host
        # ceiling, can be read from.  So beside it the SAME files listed 8 x N times (fixed work per GPU, nothing more to
        # generate; the MLF repeats its entries): a list loop of >= 2 s per GPU, every mode, one run each.  The criteria
        # of the N-GPU plan are taken from these runs: per mode the host ceiling (frames x cores / host CPU seconds of the
        # run) over 8 x the per-GPU rate, and -F -D against -F.
        reps = 8 * n_gpus
        import shutil
        while reps > n_gpus and 40e6 * reps * (n_files / 10000.0) > shutil.disk_usage(td).free / 2:      # (the MLF: ~30 MB per 10 000 files)
            reps //= 2
        rep_lst = os.path.join(td, "list_x%d.scp" % reps)
        with open(rep_lst, "w") as f:
            for _ in range(reps):
                f.write("".join(n + "\n" for n in names))
        weak = {"files": n_files * reps, "frames": frames * reps,
                "what": "the same list with every file listed %d times (8 x N GPUs: fixed work per GPU, list loop >= 2 s); "
                        "value = frames/s of the list loop, process_frames_per_s = frames / process wall clock (exec to "
                        "exit); host_ceiling_frames_per_s = frames x cores_usable / host_cpu_s of the same run; "
                        "ceiling_over_8_gpus = that / (8 x value / N)" % reps}
        wmlf = os.path.join(td, "weak.mlf")

        def weak_run(extra, g, e):
            r, _pr = run_cli(exe, ["-c", mdir, "-l", rep_lst, "-m", wmlf, "-g", str(g)] + extra, e, timeout=900)
            if "error" not in r and r.get("host_cpu_s", 0) > 0:
                r["host_ceiling_frames_per_s"] = round(frames * reps * cores / r["host_cpu_s"], 1)
                r["ceiling_over_8_gpus"] = round(r["host_ceiling_frames_per_s"] / (8.0 * r["value"] / n_gpus), 3)
            return r
        for key, extra in LIST_MODES:
            try:
                weak[key] = weak_run(extra, n_gpus, host_env if key == "host_frontend" else env)
            except Exception as e:
                weak[key] = {"error": repr(e)}
        if n_gpus == 1:
            # What `phnrec -g 8 -l ... -m ...`, called as the reference is called (no flags), selects by itself -- the GPU
            # front-end and, from four GPUs on, -D; sleeping waits -- with all eight logical GPUs mapped onto this
            # box's one device: the per-GPU rate is that of ONE GPU, the host CPU seconds are those the auto-selected path
            # costs per frame.  as_g8_default: as the CLI runs it (a physical device gets three contexts at most, however
            # many logical GPUs are mapped onto it); as_g8_all_contexts: PHNREC_ALL_CONTEXTS=1, every planned context -- 24:
            # one process, 24 worker threads, one launch queue -- the rehearsal of the eight-GPU arrangement's host side that
            # rounds 4 and 5 recorded under the first name.
            for key8, extra_env in (("as_g8_default", {}), ("as_g8_all_contexts", {"PHNREC_ALL_CONTEXTS": "1"})):
                try:
                    e8 = dict(env, PHNREC_DEVICE_MAP=",".join([str(dmap[0])] * 8), **extra_env)
                    r = weak_run([], 8, e8)
                    if "ceiling_over_8_gpus" in r:       # (eight logical GPUs, ONE device: value is one GPU's rate)
                        r["ceiling_over_8_gpus"] = round(r["host_ceiling_frames_per_s"] / (8.0 * r["value"]), 3)
                    r["what"] = ("PHNREC_DEVICE_MAP=%s %sphnrec -g 8 without -F / -E / -D: the mode the CLI picks by itself ('mode'), "
                                 "eight logical GPUs on one device" % (e8["PHNREC_DEVICE_MAP"], "PHNREC_ALL_CONTEXTS=1 " if extra_env else ""))
                    weak[key8] = r
                except Exception as e:
                    weak[key8] = {"error": repr(e)}
        try:
            weak["F_D_over_F"] = round(weak["gpu_frontend_decoder_F_D"]["value"] / weak["gpu_frontend_F"]["value"], 4)
        except Exception:
            pass
        out["weak_list"] = weak
        if n_gpus == 1:
            # configs[3] itself (the 1x list) as `-g 8` takes it, eight logical GPUs on this box's one device: what the run
            # picks by itself (F+D,auto; 2 contexts per GPU planned).  A process-level criterion: set-up + list no longer than
            # `-g 1` needs on the same device (round 5: 0.60 + 0.30 against 0.30 + 0.28 s -- 24 contexts built in front of a
            # 0.28 s list; now the contexts come up beside the list, share ONE copy of the model per device, and those the
            # list would not live to see are left out).  Better of two runs, as the modes above.
            try:
                e8 = dict(env, PHNREC_DEVICE_MAP=",".join([str(dmap[0])] * 8))
                best = None
                for _ in range(2):
                    r, _pr = run_cli(exe, ["-c", mdir, "-l", lst, "-m", os.path.join(td, "g8.mlf"), "-g", "8"], e8)
                    if "error" in r or best is None or r["process_wall_s"] < best["process_wall_s"]:
                        best = r
                    if "error" in r:
                        break
                if "error" not in best:
                    best["mlf_equals_g1"] = open(os.path.join(td, "g8.mlf")).read() == open(mlfs["gpu_frontend_F"]).read()
                    g1 = out.get("gpu_frontend_decoder_F_D", {})
                    if "process_wall_s" in g1:
                        best["g1_F_D_process_wall_s"] = g1["process_wall_s"]
                        best["g1_F_D_setup_plus_list_s"] = round(g1["setup_s"] + g1["list_wall_s"], 3)
                    best["setup_plus_list_s"] = round(best["setup_s"] + best["list_wall_s"], 3)
                out["as_g8_on_1x_list"] = best
            except Exception as e:
                out["as_g8_on_1x_list"] = {"error": repr(e)}
        # the headline's system (CZ) through the same list and modes (round 3's cli_e2e leg timed a 0.06-0.1 s loop of
        # 2000 files: inside the start-up ramp this list exists to amortise)
        cz_dir = os.path.join(ROOT, "tests", "golden", "models", "PHN_CZ_SPDAT_LCRC_N1500")
        if os.path.isdir(cz_dir):
            cz = {}
            for key, extra in (("host_frontend", []), ("gpu_frontend_F", ["-F"])):
                try:
                    r, _pr = run_cli(exe, ["-c", cz_dir, "-l", lst, "-m", os.path.join(td, "cz.mlf"), "-g", str(n_gpus)] + extra,
                                     host_env if key == "host_frontend" else env)
                    cz[key] = r
                except Exception as e:
                    cz[key] = {"error": repr(e)}
            out["cz_same_list"] = cz
        try:
            a, b = open(mlfs["gpu_frontend_F"]).read(), open(mlfs["gpu_frontend_decoder_F_D"]).read()
            out["mlf_F_equals_F_D"] = a == b
            out["mlf_entries"] = a.count('"\n') if a else 0
            # -E's features are the host front-end's bit for bit: so is its MLF
            out["mlf_E_equals_host_frontend"] = open(mlfs["gpu_energies_E"]).read() == open(mlfs["host_frontend"]).read()
            out["mlf_E_D_equals_host_frontend"] = open(mlfs["gpu_energies_decoder_E_D"]).read() == open(mlfs["host_frontend"]).read()
            # ... and so are -F's, with ln() taken as this host's libm takes it (lcrc_frontend_set_ln)
            out["mlf_F_equals_host_frontend"] = a == open(mlfs["host_frontend"]).read()
        except Exception:
            pass
        # ---- what the host alone can do on this list ----
        # The CLI times its host stages (CPU seconds summed over its threads: stat / read of the files into pinned
        # memory, gather, Viterbi + label formatting + MLF): with the GPUs infinitely fast the list would still need
        # that much CPU.  ceiling = frames x usable cores / host CPU seconds of the SAME run.
        ceil = {"what": "frames x cores_usable / host_cpu_s of the run above (PHNREC_STATS: CPU seconds of the host stages -- "
                        "file reads into pinned memory, gather, host Viterbi, label / MLF formatting -- summed over the "
                        "pool's threads); the rate the host side of `phnrec -g N` cannot exceed on this box however many "
                        "GPUs serve it"}
        for key, _extra in LIST_MODES:
            r = out.get(key)
            if isinstance(r, dict) and r.get("host_cpu_s", 0) > 0:
                ceil[key] = {"frames_per_s": round(frames * cores / r["host_cpu_s"], 1), "host_cpu_s": r["host_cpu_s"],
                             "cpu_s_by_stage": r.get("cpu_s_by_stage")}
        if "gpu_frontend_F" in ceil:
            ceil["frames_per_s"] = ceil["gpu_frontend_F"]["frames_per_s"]
        try:
            # the decoder leg alone, GPU idle: posterior dumps of the first files through `phnrec -s post -l ... -m`
            sub = min(n_files, 1000)
            sub_lst = os.path.join(td, "sub.scp")
            with open(sub_lst, "w") as f:
                for i in range(sub):
                    f.write("%s %s\n" % (names[i], os.path.join(td, "p%05d.lop" % i)))
            r, _pr = run_cli(exe, ["-c", mdir, "-l", sub_lst, "-t", "post", "-F", "-g", str(n_gpus)], env)
            if "error" not in r:
                lop_lst = os.path.join(td, "lop.scp")
                with open(lop_lst, "w") as f:
                    f.write("".join("%s\n" % os.path.join(td, "p%05d.lop" % i) for i in range(sub)))
                best = None
                for _ in range(2):
                    v, _pr = run_cli(exe, ["-c", mdir, "-s", "post", "-l", lop_lst, "-m", os.path.join(td, "v.mlf")], env)
                    if "error" not in v and (best is None or v["value"] > best["value"]):
                        best = v
                if best:
                    ceil["decoder_only_frames_per_s"] = best["value"]
                    ceil["decoder_only_sample"] = "%d posterior dumps: `phnrec -s post` (HTK read + Viterbi + MLF), no GPU" % sub
        except Exception as e:
            ceil["decoder_only_error"] = repr(e)
        # ---- ... and what the pipeline's SERIAL per-file work allows ----
        # CPU seconds say what the cores must deliver in sum; the feeder, the window's lock and the in-order writer handle
        # every file one after the other.  Files so short that the GPU has next to nothing to do (0.25 s = 23 frames)
        # measure that: files per second of the `-F` pipeline, times this list's frames per file = the frame rate the
        # serial part could feed with configs[3]'s files.
        try:
            tiny = 20000
            sig = np.clip(np.random.default_rng(5).normal(0, 3000, 2000), -32768, 32767).astype("<i2")
            tdir = os.path.join(td, "tiny")
            os.mkdir(tdir)
            tnames = []
            for i in range(tiny):
                pth = os.path.join(tdir, "s%05d.raw" % i)
                sig.tofile(pth)
                tnames.append(pth)
            tl = os.path.join(td, "tiny.scp")
            with open(tl, "w") as f:
                f.write("".join(x + "\n" for x in tnames))
            best = None
            for _ in range(2):
                r, _pr = run_cli(exe, ["-c", mdir, "-l", tl, "-m", os.path.join(td, "tiny.mlf"), "-g", str(n_gpus), "-F"], env)
                if "error" not in r and (best is None or r["list_wall_s"] < best["list_wall_s"]):
                    best = r
            if best:
                fps = tiny / best["list_wall_s"]
                ceil["per_file_serial"] = {"files_per_s": round(fps, 1), "frames_per_s_at_this_lists_file_length": round(fps * frames / n_files, 1),
                                           "sample": "%d files of 0.25 s (23 frames), -F: list loop %.3f s" % (tiny, best["list_wall_s"])}
        except Exception as e:
            ceil["per_file_serial_error"] = repr(e)
        out["host_ceiling"] = ceil
    return out


def four_systems_leg(n_gpus, dmap, n_files=2500, variants=None):
    """BASELINE configs[4]: all four LCRC systems at once, two GPUs each, mixed 8 / 16 kHz lists, end to end.
    tools/run_four_systems.sh starts four `phnrec -g 2` processes (CZ, HU, RU on 8 kHz lin16 files, EN on 16 kHz), each on its
    GPU pair; the pairs come from the ranks' device map -- 8 GPUs: 0,1 2,3 4,5 6,7; fewer: the pairs wrap around; ONE GPU:
    all four systems' eight logical GPUs on device 0 ("oversubscribed": the figure is then what one MI355X does for four
    systems at once, not the 8-GPU arrangement's).  xRT = wall clock / seconds of audio (one frame = 10 ms)."""
    import subprocess
    script = os.path.join(ROOT, "tools", "run_four_systems.sh")
    systems = ("PHN_CZ_SPDAT_LCRC_N1500", "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500")
    dirs = [os.path.join(ROOT, "tests", "golden", "models", x) for x in systems]
    exe = os.path.join(ROOT, "phnrec_amd", "bin", "phnrec")
    if not os.path.exists(exe) or not all(os.path.isdir(d) for d in dirs):
        return {"error": "CLI or a model directory missing"}
    pairs = ["%d,%d" % (dmap[(2 * i) % len(dmap)], dmap[(2 * i + 1) % len(dmap)]) for i in range(4)]
    out = {"systems": list(systems), "files_per_system": n_files, "gpu_pairs": pairs,
           "oversubscribed": len(set(dmap)) < 8,
           "what": "tools/run_four_systems.sh: four concurrent `phnrec -c SYS -l list -m mlf -g 2` processes, one per system, on "
                   "the GPU pairs named (with fewer than 8 GPUs the pairs share devices: oversubscribed); raw lin16 files of "
                   "3-15 s, 8 kHz for CZ / HU / RU (the same files, one list each) and 16 kHz for EN; per system the CLI's own "
                   "list-loop figures, in sum frames / wall clock of the whole script (process start-up, model loads, "
                   "list loops, MLFs on disk) and xRT = that wall clock / seconds of audio"}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        d8, d16 = os.path.join(td, "k8"), os.path.join(td, "k16")
        os.mkdir(d8)
        os.mkdir(d16)
        _l8, names8, _f8 = synthetic_list(d8, n_files, seed=1240, rate=8000)
        lst16, _n16, _f16 = synthetic_list(d16, n_files, seed=1241, rate=16000)
        lists = []
        for k, x in enumerate(systems):
            if x.startswith("PHN_EN"):
                lists.append(lst16)
            else:
                # (one list file per system -- the script writes each MLF next to its list --, and one file fewer per system
                #  down the line so that the four stats lines can be told apart by their file counts)
                p = os.path.join(td, "%s.scp" % x[4:6].lower())
                with open(p, "w") as f:
                    f.write("".join(n + "\n" for n in names8[:n_files - 1 - k]))
                lists.append(p)
        env = dict(os.environ, PHNREC_GPU_PAIRS=" ".join(pairs), PHNREC_BIN=exe)
        env.pop("PHNREC_DEVICE_MAP", None)
        args = [a for pr in zip(dirs, lists) for a in pr]
        # variants (tools/four_systems_ab.py): (key, extra flags, extra environment)
        for key, extra, env_extra in (variants or (("default_flags", [], {}), ("gpu_frontend_decoder_F_D", ["-F", "-D"], {}))):
            best = None
            for _ in range(2):               # the better of two runs (the first warms the page cache)
                time.sleep(SETTLE_S)
                t0 = time.perf_counter()
                pr = subprocess.run(["bash", script] + args + extra, env=dict(env, **env_extra), capture_output=True, text=True, timeout=900)
                wall = time.perf_counter() - t0
                stats = [ln for ln in pr.stderr.splitlines() if ln.startswith("phnrec: files=")]
                if pr.returncode != 0 or len(stats) != 4:
                    best = {"error": "rc=%d %s" % (pr.returncode, pr.stderr.strip()[-300:])}
                    break
                per, tot = {}, 0
                for ln in stats:
                    kv = dict(tok.split("=", 1) for tok in ln.replace("(", "").replace(")", "").split() if "=" in tok)
                    nf = int(kv["files"])
                    name = systems[3] if nf == n_files else systems[n_files - 1 - nf]
                    # (the list loop while contexts work on it, as in run_cli: wall_s minus the time until the first context was up)
                    loop_s = max(1e-6, float(kv["wall_s"]) - float(kv.get("first_ctx_s", 0)))
                    per[name] = {"files": nf, "frames": int(kv["frames"]), "frames_per_s": round(int(kv["frames"]) / loop_s, 1),
                                 "list_wall_s": round(loop_s, 3), "first_ctx_s": float(kv.get("first_ctx_s", 0)),
                                 "main_s": float(kv["main_s"]), "xrt": float(kv["xRT"]), "contexts": int(kv.get("contexts", 0)),
                                 "mode": kv.get("mode"), "host_cpu_s": float(kv["host_cpu_s"])}
                    tot += int(kv["frames"])
                r = {"per_system": per, "frames": tot, "process_wall_s": round(wall, 3),
                     "value": round(tot / wall, 1), "unit": "frames/s (four systems in sum, whole script)",
                     "xrt": round(wall / (tot * 0.01), 8),
                     "list_loops_frames_per_s": round(tot / max(p["list_wall_s"] for p in per.values()), 1)}
                if best is None or r["value"] > best["value"]:
                    best = r
            out[key] = best
        # the MLFs of the last four-system run against each system run alone on one GPU (same flags)
        try:
            same = {}
            for d, l, x in zip(dirs, lists, systems):
                solo = os.path.join(td, x + ".solo.mlf")
                e1 = dict(os.environ, PHNREC_DEVICE_MAP=str(dmap[0]))
                time.sleep(SETTLE_S)
                sp = subprocess.run([exe, "-c", d, "-l", l, "-m", solo, "-F", "-D"], env=e1, capture_output=True, text=True, timeout=600)
                if sp.returncode != 0:
                    raise RuntimeError("single-system run of %s: rc=%d %s" % (x, sp.returncode, sp.stderr.strip()[-300:]))
                same[x] = open(solo).read() == open(l[:-4] + ".mlf").read()
            out["mlf_equals_single_system_run"] = same
        except Exception as e:
            out["mlf_check_error"] = repr(e)
    return out


def reference_cli_leg(mdir, gpu):
    """The LITERAL drop-in: the reference's own command line (its phnrec.cpp / srec.cpp / melbanks.cpp / phndec.cpp,
    compiled from /root/reference in the build container by tests/integration/Makefile) with class Traps forwarding
    to this library (INTEGRATION.md option (a)).  Single-threaded host code as the reference wrote it; offline,
    one Traps::CalcFeaturesBunched call per file."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "integration", "_build", "phnrec_ref_lcrc")
    if not os.path.exists(exe):
        return None
    raw = os.path.join(ROOT, "tests", "golden", "test.raw")
    if not os.path.exists(raw):
        return None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        blob = open(raw, "rb").read()
        n_files, per = 40, 8                       # 40 files of 8 x test.raw (59.9 s each, 5990 frames)
        names = []
        for i in range(n_files):
            p = os.path.join(td, "f%03d.raw" % i)
            with open(p, "wb") as f:
                f.write(blob * per)
            names.append(p)
        lst = os.path.join(td, "list.scp")
        with open(lst, "w") as f:
            f.write("".join(n + "\n" for n in names))
        frames = n_files * ((len(blob) * per // 2 - 200) // 80 + 1)
        env = dict(os.environ, PHNREC_DEVICE=str(gpu))
        out = {"files": n_files, "frames": frames,
               "what": "reference CLI (its own single-threaded front-end, decoder and file loop) over libphnrec_lcrc.so "
                       "through the Traps binding of INTEGRATION.md; wall clock of the whole process minus that of a "
                       "1-file run (start-up, model load)"}
        try:
            t0 = time.perf_counter()
            subprocess.run([exe, "-c", mdir, "-i", names[0], "-o", os.path.join(td, "one.rec")], env=env, check=True,
                           capture_output=True, timeout=120)
            t_one = time.perf_counter() - t0
            t0 = time.perf_counter()
            subprocess.run([exe, "-c", mdir, "-l", lst, "-m", os.path.join(td, "out.mlf")], env=env, check=True,
                           capture_output=True, timeout=300)
            t_all = time.perf_counter() - t0
            per_file = (t_all - t_one) / (n_files - 1)
            out.update({"value": round(frames / n_files / per_file, 1), "unit": "frames/s",
                        "process_wall_s": round(t_all, 3), "one_file_process_wall_s": round(t_one, 3)})
        except Exception as e:
            out["error"] = repr(e)
        return out


def emit(full, detail_out):
    """The record leaves in three forms: the FULL one (every leg's break-down and its sentence of what it measured)
    to a file and, leg by leg, to stderr; stdout gets ONE line of at most benchline.LIMIT bytes -- the contract's
    keys, `roofline`, `cpu_baseline` and numbers only per side leg (the driver parses a bounded line: round 5's
    21.7 KB one was cut and its record lost)."""
    from phnrec_amd import benchline
    path = detail_out or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        full = dict(full, detail=os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path)
    except OSError as e:                      # a read-only tree: stderr still has it
        sys.stderr.write("bench.py: detail file not written: %r\n" % (e,))
    head = {k: v for k, v in full.items() if not isinstance(v, (dict, list))}
    sys.stderr.write("bench detail [head]: %s\n" % json.dumps(head))
    for k, v in full.items():
        if isinstance(v, (dict, list)):
            sys.stderr.write("bench detail [%s]: %s\n" % (k, json.dumps(v)))
    sys.stderr.flush()
    _RECORD_OUT.write(benchline.stdout_line(full) + "\n")
    _RECORD_OUT.flush()


def cli_process_legs(args, dmap, mdir):
    """The legs that time whole CLI PROCESSES (single file, the reference's CLI over the library, configs[3]'s list by mode,
    configs[4]).  A plain one-GPU run calls this BEFORE its own process touches the GPU: a process that starts while another
    one holds the device takes 20-50 ms longer over hipInit and its first stream (profiles/r06_ab_runs.txt 5), and what these
    legs report is what ONE command costs on a GPU nobody else holds."""
    out = {}
    if not args.no_extras:
        for key, leg in (("single_file", lambda: single_file_leg(mdir, dmap[0])),
                         ("dropin_reference_cli", lambda: reference_cli_leg(mdir, dmap[0]))):
            try:
                val = leg()
            except Exception as e:      # side legs are reported when they can be measured, never fatal
                val = {"error": repr(e)}
            if val is not None:
                out[key] = val
    if args.list_files > 0:
        # the thing north_star asks to scale: the sharded file list through the CLI, -g N over the ranks' GPUs
        try:
            out["sharded_list"] = sharded_list_leg(len(dmap), dmap, args.list_files)
        except Exception as e:
            out["sharded_list"] = {"error": repr(e)}
        # BASELINE configs[4]: the four systems at once on the ranks' GPUs (one GPU: all on it, labelled)
        try:
            out["four_systems"] = four_systems_leg(len(dmap), dmap, max(4, args.list_files // 4))
        except Exception as e:
            out["four_systems"] = {"error": repr(e)}
    return out


def stub_main(args, ranks):
    """Launcher self-test: everything of the N-rank harness except the GPU (see --stub)."""
    from phnrec_amd import distrun
    ranks.init("gloo")
    state = {"n": 0}

    def step():
        time.sleep(0.002 * (1 + ranks.rank))         # the last rank is the slowest: MAX over ranks must see it
        state["n"] += BATCH

    elapsed = distrun.timed_steps(ranks, step, lambda: None, args.steps, args.warmup)
    total = ranks.sum_int(state["n"])
    if ranks.rank == 0:
        emit({"stub": True, "metric": "launcher self-test (no GPU work)", "n_gpus": 0,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
              "frames_all_ranks": total,
              "ranks": {"world": ranks.world, "launcher": ranks.launcher, "backend": ranks.backend,
                        "device_map": distrun.device_map(ranks.world)}}, args.detail_out)
    ranks.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    # Disclosed pre-heat, OUTSIDE warmup/steps: after idling the device needs ~100 launches (~25 ms of load) to
    # reach its steady clock (0.236 ms per launch at launch 11-20, 0.210 ms from launch ~100 on; profiles/README.md).
    # The line states it ("preheat_launches") and also carries the cold figure of the first launches
    # ("roofline.cold"), so a short --warmup neither hides nor includes the ramp silently.  0 switches it off.
    ap.add_argument("--preheat", type=int, default=300)
    ap.add_argument("--list-files", type=int, default=10000,
                    help="files of the sharded_list leg (BASELINE configs[3]: 10 000; 0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the small-launch / push / wave / CLI legs")
    ap.add_argument("--kernel-only", action="store_true",
                    help="nothing but the headline launches (PMC passes: every lcrc_fused_kernel dispatch of the process is then "
                         "one of the 8192-row launches; implies --no-extras --no-cpu --list-files 0 and skips the host-pointer legs)")
    ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    # launcher self-test (tests/test_distrun.py): the same launch / rendezvous / barrier / MAX-over-ranks
    # code with a sleeping step and gloo instead of the GPU step and RCCL; its line says "stub": true
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--detail-out", default=None,
                    help="file for the full record (default gpurun_out/bench_detail.json); stdout carries its compact form")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.kernel_only:
        args.no_extras, args.no_cpu, args.list_files = True, True, 0

    from phnrec_amd import distrun

    if args.gpus > 1 and "RANK" not in os.environ:
        # No launcher around us: start the N ranks ourselves, as fresh child processes, BEFORE anything in
        # this process touches the GPU (counting devices does not initialise HIP on this image).
        if not args.stub:
            import torch
            have, need = torch.cuda.device_count(), max(distrun.device_map(args.gpus)) + 1
            if have < need:
                raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); refusing to report a %d-GPU "
                                 "figure from fewer devices (PHNREC_DEVICE_MAP=0,0,... maps several ranks onto "
                                 "one GPU for a functional test and labels the output accordingly)"
                                 % (args.gpus, have, args.gpus))
        sys.exit(distrun.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # (a self-launching parent has left above: its ranks inherit the untouched stdout)
    # stdout carries exactly ONE line, the JSON record.  Libraries print there too (gloo announces its ranks on stdout
    # from C++, child programs inherit it), so file descriptor 1 is pointed at stderr for the life of the process and the
    # record goes out through a private copy of the original stdout.
    global _RECORD_OUT
    sys.stdout.flush()
    _RECORD_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    ranks = distrun.Ranks(args.gpus)
    if ranks.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, ranks.world))
    dmap = distrun.device_map(ranks.world)
    oversubscribed = len(set(dmap)) < len(dmap)
    if args.stub:
        return stub_main(args, ranks)
    # A plain one-GPU run (the driver's N = 1 form): the CLI's processes are timed first, before this process holds the GPU
    # (the shipped weights must be there; with synthetic ones, and in N-rank runs, they are timed behind the headline as before)
    pre, real_mdir = {}, os.path.join(ROOT, "tests", "golden", "models", SYSTEM)
    if ranks.world == 1 and not ranks.launched and not args.kernel_only and os.path.isdir(real_mdir):
        pre = cli_process_legs(args, dmap, real_mdir)
    import torch
    from phnrec_amd import capi, modelgen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the LCRC path has no CPU fallback")
    gpu = dmap[ranks.local_rank]
    if gpu >= torch.cuda.device_count():
        raise SystemExit("rank %d wants GPU %d, this node shows %d" % (ranks.rank, gpu, torch.cuda.device_count()))
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    # RCCL refuses two ranks on one device: an oversubscribed functional run makes its rendezvous over gloo
    ranks.init("gloo" if oversubscribed else "nccl")
    red_dev = None if oversubscribed else dev      # where the MAX / SUM reductions' tensors live

    spec = modelgen.SYSTEMS[SYSTEM]
    nb = spec["nbanks"]
    with tempfile.TemporaryDirectory() as tmp:
        mdir, weights_desc = model_directory(tmp)
        ctx = capi.Lcrc(mdir, nb, device=gpu)
        dims = [ctx.net_dims(i) for i in range(3)]
        flops_frame = algorithmic_flops_per_frame(dims)

        # synthetic log-mel batch of this rank, resident in HBM
        mel = modelgen.synth_mel(args.batch, nb, seed=1235 + ranks.rank, mean_norm=spec["sent_mean_norm"])
        d_mel = torch.from_numpy(mel).to(dev)
        d_post = torch.empty((args.batch, ctx.n_out), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev)
        ctx.set_timing(False)       # per-step kernel times come from the events below

        # Kernel time: ONE pair of HIP events on the launch stream around the K timed launches (the average
        # launch duration over the timed region, dispatch gaps included); an event pair per launch would put
        # two more packets between consecutive kernels and inflate both this figure and the step time.
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        state = {"i": -args.warmup}

        def step():
            i = state["i"]
            if i == 0:
                ev0.record(stream)
            ctx.posteriors_device(d_mel.data_ptr(), args.batch, d_post.data_ptr(),
                                  stream=stream.cuda_stream)
            if i == args.steps - 1:
                ev1.record(stream)
            state["i"] = i + 1

        def sync():
            torch.cuda.synchronize(dev)

        # cold figure: the first 20 launches of this process (code load, clock ramp), then the disclosed pre-heat
        cold_ms = time_launches(ctx, stream, d_mel, d_post, args.batch, 20)
        if args.preheat > 0:
            time_launches(ctx, stream, d_mel, d_post, args.batch, args.preheat)
        timing = {}
        elapsed = distrun.timed_steps(ranks, step, sync, args.steps, args.warmup, device=red_dev, detail=timing)
        kernel_ms = ev0.elapsed_time(ev1) / args.steps
        kernel_ms = ranks.max_float(kernel_ms, device=red_dev)
        total_frames = args.batch * args.steps * max(1, ranks.world)
        fps = total_frames / elapsed

        line = None
        if ranks.rank == 0:
            achieved = args.batch * flops_frame / (kernel_ms * 1e-3) / 1e12
            # HBM bytes per launch are NOT measured by this run: PMC counters need rocprofv3 around the process.
            # The figure is the committed PMC pass of the same workload; its source says which round and how.
            traffic, traffic_source = None, None
            pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(pmc):
                try:
                    with open(pmc) as f:
                        rec = json.load(f)
                    traffic = rec.get("bytes_per_launch")
                    traffic_source = "profiles/hbm_traffic.json (not measured by this run): " + str(rec.get("source"))
                except Exception:
                    traffic = None
            line = {
                "metric": "frames/sec (LCRC posterior path)", "value": round(fps, 1), "unit": "frames/s",
                "n_gpus": len(set(dmap)), "steps": args.steps, "warmup": args.warmup,
                "preheat_launches": args.preheat,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                # (the contract's bracket: barrier + synchronise on both sides; beside it each rank's clock at its own
                #  synchronise, MAX over ranks -- what the closing collective adds at N > 1)
                "ms_per_step_before_closing_barrier": round(timing["before_closing_barrier"] / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "%s, batch=%d frames per GPU per step (BASELINE configs[2]), "
                                       "log-mel frames resident in HBM, posteriors left in HBM"
                                       % (SYSTEM, args.batch),
                           "weights": weights_desc, "kernel": ctx.kernel_name,
                           "arithmetic": "f32 operands on v_mfma_f32_16x16x4_f32, f32 accumulation (LCRC_ARITH_F32, the default; "
                                         "the opt-in split-f16 arithmetic is the separate leg `split_f16`)",
                           "sharding": "one replica per GPU, utterances never exchanged (no collective)"},
                # what actually ran: ranks as the process group counted them, how they were started, where
                "ranks": {"world": ranks.world, "launcher": ranks.launcher, "backend": ranks.backend,
                          "device_map": dmap, "oversubscribed": oversubscribed},
                "frames_per_s_per_gpu": round(fps / max(1, ranks.world), 1),
                "xrt": round(100.0 / (fps / max(1, ranks.world)), 8),
                "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                             "traffic": traffic, "traffic_source": traffic_source,
                             "kernel_ms": round(kernel_ms, 4), "flop_per_frame": flops_frame,
                             "launches_timed": "%d launches after %d pre-heat + %d warm-up launches"
                                               % (args.steps, args.preheat + 20, args.warmup),
                             # the same kernel straight after start-up: launches 1-20 of this process
                             "cold": {"launches": "1-20", "kernel_ms": round(cold_ms, 4),
                                      "frac": round(args.batch * flops_frame / (cold_ms * 1e-3) / 1e12
                                                    / PEAK_F32_MFMA_TFLOPS, 4)},
                             # secondary figure: HBM-side bytes (PMC, per launch) over the live kernel time,
                             # against the ~8 TB/s HBM3E peak -- the kernel is nowhere near memory-bound
                             "hbm_gbps": (round(traffic / (kernel_ms * 1e-3) / 1e9, 1) if traffic else None),
                             "hbm_frac_of_8TBps": (round(traffic / (kernel_ms * 1e-3) / 8e12, 4) if traffic else None)},
            }
            if ranks.world == 1 and not args.kernel_only:
                # the spread of the headline's launches, outside the timed region: median / min / max over windows
                ws = sorted(time_windows(ctx, stream, d_mel, d_post, args.batch))
                line["roofline"]["windows"] = {"kernel_ms_median": round(ws[len(ws) // 2], 4), "kernel_ms_min": round(ws[0], 4),
                                               "kernel_ms_max": round(ws[-1], 4),
                                               "what": "%d x %d further launches back to back behind the timed region" % (len(ws), PER_WINDOW)}
            if ranks.world == 1 and not args.kernel_only:
                # the host-pointer entry point (pageable buffers in, pageable out): PCIe-inclusive,
                # reported beside `value`, never as it
                # (the C entry point on the caller's own, reused buffers: a fresh numpy array per call would add
                #  its page faults -- 0.3 ms for 4.5 MB of posteriors -- to every call)
                reps = 20
                h_post = np.empty((args.batch, ctx.n_out), np.float32)
                for _ in range(10):
                    ctx._check(ctx.L.lcrc_posteriors(ctx.h, mel, args.batch, h_post))
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    ctx.L.lcrc_posteriors(ctx.h, mel, args.batch, h_post)
                    ts.append(time.perf_counter() - t0)
                dt = float(np.median(ts))
                line["host_path"] = {"value": round(args.batch / dt, 1), "unit": "frames/s",
                                     "ms_per_call": round(dt * 1e3, 4),
                                     "what": "lcrc_posteriors() on reused pageable buffers: memcpy to pinned + H2D + kernel + "
                                             "D2H + memcpy, synchronous"}
                # the zero-copy form (lcrc_stage_buffers / lcrc_stage_run): the caller fills and reads the context's
                # pinned buffers, so what is left is H2D + kernel + D2H -- the PCIe-inclusive floor of this launch size
                import ctypes as C
                pm, pp = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
                ctx._check(ctx.L.lcrc_stage_buffers(ctx.h, args.batch, C.byref(pm), C.byref(pp)))
                C.memmove(pm, mel.ctypes.data, mel.nbytes)
                one = np.array([0, args.batch], np.int32)
                for _ in range(10):
                    ctx._check(ctx.L.lcrc_stage_run(ctx.h, one, 1))
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    ctx.L.lcrc_stage_run(ctx.h, one, 1)
                    ts.append(time.perf_counter() - t0)
                dt = float(np.median(ts))
                line["host_path_zero_copy"] = {"value": round(args.batch / dt, 1), "unit": "frames/s",
                                               "ms_per_call": round(dt * 1e3, 4),
                                               "what": "lcrc_stage_run() on the context's pinned buffers: the kernel reads the features and stores the "
                                                       "posteriors in place (mapped, over PCIe while it runs: no copy commands), synchronous"}
            if ranks.world == 1 and not args.no_extras:
                en_dir = os.path.join(ROOT, "tests", "golden", "models", "PHN_EN_TIMIT_LCRC_N500")
                for key, leg in (("small_launches", lambda: small_launch_legs(capi, modelgen, dev, stream)),
                                 ("wave_path", lambda: wave_path_leg(capi, mdir, nb, gpu)),
                                 ("wave_path_en", lambda: wave_path_leg(
                                     capi, en_dir, 23, gpu, raw=config1_lin16_signal(), wave_format="lin16", sent_mean_norm=False,
                                     what="configs[1] input per SURVEY 8(d) cfg2 (EN, 16 kHz lin16, 5 sines + noise, seed 1234, "
                                          "4096 frames, posterior-only): lcrc_wave_to_posteriors(), host bytes in, host posteriors out "
                                          "(reused buffers), synchronous") if os.path.isdir(en_dir) else None),
                                 ("single_file", lambda: pre["single_file"] if "single_file" in pre else
                                  None if pre else single_file_leg(mdir, gpu)),
                                 ("dropin_reference_cli", lambda: pre["dropin_reference_cli"] if "dropin_reference_cli" in pre else
                                  None if pre else reference_cli_leg(mdir, gpu))):
                    try:
                        val = leg()
                    except Exception as e:      # side legs are reported when they can be measured, never fatal
                        val = {"error": repr(e)}
                    if val is not None:
                        line[key] = val
                if isinstance(line.get("small_launches"), dict):
                    for k in ("push_bunch5", "push_bunch512"):
                        if k in line["small_launches"]:
                            line[k] = line["small_launches"].pop(k)
                    # the other shipped systems at the headline's launch size: same measurement, their own leg
                    sys_keys = [k for k in line["small_launches"] if k.endswith("_%d" % BATCH)]
                    if sys_keys:
                        line["systems"] = {k: line["small_launches"].pop(k) for k in sys_keys}
                        line["systems"]["what"] = ("roofline fraction (algorithmic FLOP of the system's three nets / f32 MFMA "
                                                   "peak) of %d-frame launches of the other shipped systems, each behind its own "
                                                   "time-based pre-heat, median of %d windows with min / max; CZ at this size is "
                                                   "`roofline`" % (BATCH, WINDOWS))
            line["config"]["cli_processes_timed"] = ("before this process touched the GPU" if pre else
                                                     "behind the headline, this process holding the GPU")
            if args.list_files > 0 and pre:
                line["sharded_list"], line["four_systems"] = pre["sharded_list"], pre["four_systems"]
            elif args.list_files > 0:
                # (N ranks: the other ranks idle at the barrier below; their contexts hold no work)
                legs = cli_process_legs(argparse.Namespace(no_extras=True, list_files=args.list_files), dmap, mdir)
                line["sharded_list"], line["four_systems"] = legs["sharded_list"], legs["four_systems"]
            split_post = None
            gpu_post = d_post.cpu().numpy() if ranks.world == 1 else None
            if ranks.world == 1 and not args.no_extras:
                try:
                    line["split_f16"], sp = split_f16_leg(capi, ctx, stream, d_mel, d_post, args.batch, flops_frame,
                                                          torch.from_numpy(gpu_post).to(dev), kernel_ms)
                    split_post = sp.cpu().numpy()
                except Exception as e:
                    line["split_f16"] = {"error": repr(e)}
            if ranks.world == 1 and not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline(mdir, nb, mel, gpu_post, args.cpu_seconds, split_post)
        ctx.close()
    ranks.host_barrier()     # rank 0's side legs are over (a CPU-side wait: the other ranks' GPUs stay idle meanwhile)
    ranks.finish()
    if line is not None:
        emit(line, args.detail_out)


if __name__ == "__main__":
    main()
