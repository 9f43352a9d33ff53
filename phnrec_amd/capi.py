"""ctypes view of include/lcrc.h (libphnrec_lcrc.so) for tests and bench.py.

This is plumbing, not a second implementation: every call goes through the C
ABI into the HIP kernel.  If the library has not been built, or no GPU is
usable, it raises -- there is no fallback of any kind.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libphnrec_lcrc.so")

# every symbol include/lcrc.h declares (tests check the library exports them all)
SYMBOLS = [
    "lcrc_create", "lcrc_clone", "lcrc_device_warmup", "lcrc_device_preload", "lcrc_device_pci_bus_id", "lcrc_create_system", "lcrc_model_outputs", "lcrc_destroy", "lcrc_last_error", "lcrc_abi_version", "lcrc_model_info",
    "lcrc_num_outputs", "lcrc_num_banks", "lcrc_trap_shift", "lcrc_device", "lcrc_net_dims",
    "lcrc_posteriors", "lcrc_posteriors_batch", "lcrc_posteriors_device", "lcrc_posteriors_probe", "lcrc_posteriors_rows",
    "lcrc_stage_buffers", "lcrc_stage_run",
    "lcrc_frontend_configure", "lcrc_frontend_set_ln", "lcrc_device_ln", "lcrc_set_mean_order", "lcrc_frontend_frames", "lcrc_wave_to_mel", "lcrc_wave_to_posteriors",
    "lcrc_reserve", "lcrc_wave_stage_buffer", "lcrc_wave_stage_run", "lcrc_wave_stage_energies", "lcrc_staged_posteriors",
    "lcrc_output_configure", "lcrc_decoder_configure", "lcrc_set_posterior_readback", "lcrc_last_labels", "lcrc_set_decoder_overlap", "lcrc_prev_labels", "lcrc_set_launch_order",
    "lcrc_reset", "lcrc_push", "lcrc_delay",
    "lcrc_last_kernel_ms", "lcrc_set_timing", "lcrc_set_wait_mode", "lcrc_set_kernel_done_callback", "lcrc_set_tile_frames", "lcrc_set_hidden_split", "lcrc_set_arithmetic", "lcrc_debug_fail_alloc", "lcrc_debug_fail_launch", "lcrc_kernel_name",
]

LCRC_OK, LCRC_E_ARG, LCRC_E_IO, LCRC_E_MODEL, LCRC_E_DEVICE, LCRC_E_NOMEM, LCRC_E_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6

ARITH_F32, ARITH_SPLIT_F16 = 0, 1

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class Softening(C.Structure):
    """struct lcrc_softening (include/lcrc.h)"""
    _fields_ = [("func", C.c_int), ("arg1", C.c_float), ("arg2", C.c_float), ("arg3", C.c_float)]


SOFT_FUNCS = {"none": 0, "log": 1, "igor": 2, "gmm_bypass": 3}


class Label(C.Structure):
    """struct lcrc_label (include/lcrc.h)"""
    _fields_ = [("start", C.c_int), ("end", C.c_int), ("phn", C.c_int), ("score", C.c_float)]


class Frontend(C.Structure):
    """struct lcrc_frontend (include/lcrc.h)"""
    _fields_ = [("wave_format", C.c_int), ("sample_freq", C.c_int), ("vector_size", C.c_int),
                ("vector_step", C.c_int), ("nbanks_full", C.c_int), ("lower_freq", C.c_float),
                ("higher_freq", C.c_float), ("preem_coef", C.c_float), ("scale", C.c_float),
                ("dc_shift", C.c_float), ("z_mean_source", C.c_int), ("sent_mean_norm", C.c_int)]


class LcrcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lcrc error %d: %s" % (code, msg))
        self.code = code


_lib = None
_hip = None


def _load_hip_runtime():
    """libphnrec_lcrc.so carries no DT_NEEDED on libamdhip64 (see csrc/Makefile): a
    process may hold only one HIP runtime, and next to PyTorch it must be torch's
    bundled copy.  Import torch FIRST when it is installed (so a later `import torch`
    in the same process finds its own runtime already in place), then promote that
    runtime to the global symbol scope for our library to bind against."""
    global _hip
    if _hip is not None:
        return _hip
    candidates = []
    if os.environ.get("PHNREC_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
            candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        except Exception:
            pass
    candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for p in candidates:
        if os.path.isabs(p) and not os.path.exists(p):
            continue
        try:
            _hip = C.CDLL(p, mode=C.RTLD_GLOBAL)
            return _hip
        except OSError as e:   # try the next one
            last = e
    raise OSError("no HIP runtime (libamdhip64) could be loaded: %s" % last)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C phnrec_amd/csrc). There is no CPU fallback." % LIB_PATH)
    _load_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lcrc_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lcrc_clone.argtypes = [C.POINTER(vp), vp]
    L.lcrc_device_warmup.argtypes = [C.c_int]
    L.lcrc_device_preload.argtypes = [C.c_int, C.c_int]
    L.lcrc_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.lcrc_model_outputs.argtypes = [C.c_char_p, C.c_char_p]
    L.lcrc_create_system.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.lcrc_destroy.argtypes = [vp]
    L.lcrc_destroy.restype = None
    L.lcrc_last_error.argtypes = [vp]
    L.lcrc_last_error.restype = C.c_char_p
    L.lcrc_kernel_name.argtypes = [vp]
    L.lcrc_kernel_name.restype = C.c_char_p
    for name in ("lcrc_num_outputs", "lcrc_num_banks", "lcrc_trap_shift", "lcrc_device", "lcrc_delay",
                 "lcrc_reset"):
        getattr(L, name).argtypes = [vp]
    L.lcrc_net_dims.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 3
    L.lcrc_posteriors.argtypes = [vp, _f32p, C.c_int, _f32p]
    L.lcrc_posteriors_batch.argtypes = [vp, _f32p, _i32p, C.c_int, _f32p]
    L.lcrc_posteriors_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lcrc_posteriors_probe.argtypes = [vp, _f32p, C.c_int, _f32p] + [vp] * 5
    L.lcrc_push.argtypes = [vp, _f32p, C.c_int, vp, C.c_int]
    L.lcrc_stage_buffers.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float))]
    L.lcrc_stage_run.argtypes = [vp, _i32p, C.c_int]
    L.lcrc_frontend_configure.argtypes = [vp, C.POINTER(Frontend)]
    L.lcrc_output_configure.argtypes = [vp, C.POINTER(Softening), C.c_int, C.c_int]
    L.lcrc_decoder_configure.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float]
    L.lcrc_set_posterior_readback.argtypes = [vp, C.c_int]
    L.lcrc_last_labels.argtypes = [vp, C.POINTER(C.POINTER(Label)), C.POINTER(C.POINTER(C.c_int)),
                                   C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
    L.lcrc_prev_labels.argtypes = L.lcrc_last_labels.argtypes
    L.lcrc_set_decoder_overlap.argtypes = [vp, C.c_int]
    L.lcrc_set_launch_order.argtypes = [vp, C.c_int]
    L.lcrc_frontend_set_ln.argtypes = [vp, C.c_int]
    L.lcrc_frontend_frames.argtypes = [vp, C.c_longlong]
    _i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
    _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    L.lcrc_wave_to_mel.argtypes = [vp, _u8p, _i64p, C.c_int, _f32p, _i32p]
    L.lcrc_wave_to_posteriors.argtypes = [vp, _u8p, _i64p, C.c_int, _f32p, _i32p]
    L.lcrc_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_longlong]
    L.lcrc_wave_stage_buffer.argtypes = [vp, C.c_longlong, C.POINTER(C.POINTER(C.c_ubyte))]
    L.lcrc_wave_stage_run.argtypes = [vp, _i64p, _i64p, C.c_int, _f32p, _i32p]
    L.lcrc_model_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_size_t,
                                  C.POINTER(C.c_uint)]
    L.lcrc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.lcrc_set_timing.argtypes = [vp, C.c_int]
    L.lcrc_set_wait_mode.argtypes = [vp, C.c_int]
    L.lcrc_set_tile_frames.argtypes = [vp, C.c_int]
    L.lcrc_set_hidden_split.argtypes = [vp, C.c_int]
    L.lcrc_set_mean_order.argtypes = [vp, C.c_int]
    L.lcrc_set_arithmetic.argtypes = [vp, C.c_int]
    L.lcrc_debug_fail_alloc.argtypes = [C.c_int]
    L.lcrc_debug_fail_launch.argtypes = [C.c_int]
    L.lcrc_posteriors_rows.argtypes = [vp, _f32p, C.c_int, C.c_int, C.c_int, _f32p]
    _lib = L
    return L


def device_ln(x, form, device=0):
    """the GPU front-end's ln() of an array in the form named (0: double log rounded once, 1 / 2: glibc's logf sequence with /
    without fused multiply-adds), computed on the device (lcrc_device_ln)"""
    L = load()
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    y = np.empty_like(x)
    L.lcrc_device_ln.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_longlong]
    rc = L.lcrc_device_ln(int(device), int(form), x, y, x.size)
    if rc != 0:
        raise LcrcError(rc, L.lcrc_last_error(None).decode())
    return y


def model_info(model_dir, nbanks):
    """Host-only pre-flight: dims of the three nets, kernel variant, LDS bytes (no GPU needed)."""
    L = load()
    dims = (C.c_int * 9)()
    name = C.create_string_buffer(64)
    lds = C.c_uint()
    rc = L.lcrc_model_info(os.fsencode(model_dir), nbanks, dims, name, 64, C.byref(lds))
    if rc != 0:
        raise LcrcError(rc, L.lcrc_last_error(None).decode())
    d = list(dims)
    return {"dims": [tuple(d[0:3]), tuple(d[3:6]), tuple(d[6:9])], "kernel": name.value.decode(),
            "lds_bytes": lds.value}


class Lcrc:
    """One estimator context on one GPU (mirrors class Traps' public surface)."""

    def __init__(self, model_dir, nbanks, device=0, trap_len=31, add_c0=True, system="LCRC", hamming=False, clone_of=None):
        self.L = load()
        self.h = C.c_void_p()
        if clone_of is not None:         # lcrc_clone: a second context that shares clone_of's weights on the device
            rc = self.L.lcrc_clone(C.byref(self.h), clone_of.h)
        elif system == "LCRC" and not hamming:
            rc = self.L.lcrc_create(C.byref(self.h), os.fsencode(model_dir), nbanks, trap_len,
                                    int(add_c0), device)
        else:
            rc = self.L.lcrc_create_system(C.byref(self.h), os.fsencode(model_dir), system.encode(), nbanks,
                                           trap_len, int(add_c0), int(hamming), device)
        if rc != 0:
            self.h = None
            raise LcrcError(rc, self.L.lcrc_last_error(None).decode())
        self.nbanks = nbanks
        self.n_out = self.L.lcrc_num_outputs(self.h)

    def clone(self):
        return Lcrc(None, self.nbanks, clone_of=self)

    def close(self):
        if getattr(self, "h", None):
            self.L.lcrc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise LcrcError(rc, self.L.lcrc_last_error(self.h).decode())

    @property
    def kernel_name(self):
        return self.L.lcrc_kernel_name(self.h).decode()

    def net_dims(self, which):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.lcrc_net_dims(self.h, which, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- whole-utterance form (ProcessOffline's prime/main/flush) --
    def posteriors(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        post = np.empty((mel.shape[0], self.n_out), np.float32)
        self._check(self.L.lcrc_posteriors(self.h, mel, mel.shape[0], post))
        return post

    def posteriors_rows(self, mel, row_first, row_count):
        """rows [row_first, row_first + row_count) of the strip `mel`; the other rows are context only"""
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        post = np.empty((row_count, self.n_out), np.float32)
        self._check(self.L.lcrc_posteriors_rows(self.h, mel, mel.shape[0], row_first, row_count, post))
        return post

    def posteriors_batch(self, mel, off):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        off = np.ascontiguousarray(off, dtype=np.int32)
        post = np.zeros((mel.shape[0], self.n_out), np.float32)
        self._check(self.L.lcrc_posteriors_batch(self.h, mel, off, len(off) - 1, post))
        return post

    def posteriors_staged(self, mel, off):
        """lcrc_stage_buffers + lcrc_stage_run: the caller writes into the context's pinned buffers"""
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        off = np.ascontiguousarray(off, dtype=np.int32)
        n = mel.shape[0]
        pm, pp = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        self._check(self.L.lcrc_stage_buffers(self.h, n, C.byref(pm), C.byref(pp)))
        if n:
            np.ctypeslib.as_array(pm, shape=(n, self.nbanks))[:] = mel
        self._check(self.L.lcrc_stage_run(self.h, off, len(off) - 1))
        return np.ctypeslib.as_array(pp, shape=(n, self.n_out)).copy() if n else np.zeros((0, self.n_out), np.float32)

    # -- posterior writer path: softening functions and HTK byte order on the device --
    def configure_output(self, stages=(), big_endian=False):
        """stages: up to two of "none" | "log" | "gmm_bypass" | ("igor", middle, right_base, left_base)"""
        arr = (Softening * max(1, len(stages)))()
        for i, st in enumerate(stages):
            name, args = (st, ()) if isinstance(st, str) else (st[0], st[1:])
            args = list(args) + [0.0] * (3 - len(args))
            arr[i] = Softening(SOFT_FUNCS[name], *args)
        self._check(self.L.lcrc_output_configure(self.h, arr, len(stages), int(big_endian)))

    # -- decoder on the device (PhnDec) --
    def configure_decoder(self, n_phonemes, states=3, time_pruning=40, wpenalty=0.0):
        self._check(self.L.lcrc_decoder_configure(self.h, n_phonemes, states, time_pruning, wpenalty))

    def set_posterior_readback(self, on):
        self._check(self.L.lcrc_set_posterior_readback(self.h, int(on)))

    def set_decoder_overlap(self, on):
        """staged calls return when their posterior kernels are done; their labels come from prev_labels() after the next
        call, or last_labels() after the last one (lcrc_set_decoder_overlap)"""
        self._check(self.L.lcrc_set_decoder_overlap(self.h, int(on)))

    def set_launch_order(self, on):
        """posterior kernels of this device's ordered contexts run one after the other, in queueing order (lcrc_set_launch_order)"""
        self._check(self.L.lcrc_set_launch_order(self.h, int(on)))

    def prev_labels(self):
        """labels of the staged call BEFORE the most recent one (lcrc_prev_labels)"""
        return self.last_labels(fn=self.L.lcrc_prev_labels)

    def last_labels(self, fn=None):
        """labels of the most recent host-synchronous call: one list of (start, end, phn, score) per utterance"""
        lab, first, cnt = C.POINTER(Label)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        n = C.c_int()
        self._check((fn or self.L.lcrc_last_labels)(self.h, C.byref(lab), C.byref(first), C.byref(cnt), C.byref(n)))
        out = []
        for u in range(n.value):
            out.append([(lab[first[u] + k].start, lab[first[u] + k].end, lab[first[u] + k].phn, lab[first[u] + k].score)
                        for k in range(cnt[u])])
        return out

    # -- waveform entry (GPU mel-bank front-end) --
    def configure_frontend(self, wave_format="lin16", sample_freq=8000, vector_size=200, vector_step=80,
                           nbanks_full=-1, lower_freq=64.0, higher_freq=4000.0, preem_coef=0.0, scale=1.0,
                           dc_shift=0.0, z_mean_source=False, sent_mean_norm=True):
        fe = Frontend({"lin16": 1, "alaw": 2}[wave_format], sample_freq, vector_size, vector_step, nbanks_full,
                      lower_freq, higher_freq, preem_coef, scale, dc_shift, int(z_mean_source), int(sent_mean_norm))
        self._check(self.L.lcrc_frontend_configure(self.h, C.byref(fe)))

    def set_frontend_ln(self, form):
        """0: log() in double rounded once; 1 / 2: glibc's logf sequence with / without fused multiply-adds (lcrc_frontend_set_ln)"""
        self._check(self.L.lcrc_frontend_set_ln(self.h, int(form)))

    def set_mean_order(self, sequential):
        """True (default): the reference's sequential column sums; False: fixed-shape tree sums (opt-in)"""
        self._check(self.L.lcrc_set_mean_order(self.h, int(sequential)))

    def frontend_frames(self, n_bytes):
        n = self.L.lcrc_frontend_frames(self.h, n_bytes)
        if n < 0:
            raise LcrcError(n, "front-end not configured (lcrc_frontend_configure)")
        return n

    def _wave(self, fn, blobs, width):
        blobs = [bytes(b) for b in blobs]
        off = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
        raw = np.frombuffer(b"".join(blobs) + b"\0", dtype=np.uint8).copy()
        rows = sum(self.frontend_frames(len(b)) for b in blobs)
        out = np.zeros((rows, width), np.float32)
        foff = np.zeros(len(blobs) + 1, np.int32)
        self._check(fn(self.h, raw, off, len(blobs), out, foff))
        return out, foff

    def wave_to_mel(self, blobs):
        """raw files (bytes objects) -> (mel BEFORE sentence normalisation, frame offsets)"""
        return self._wave(self.L.lcrc_wave_to_mel, blobs, self.nbanks)

    def wave_to_posteriors(self, blobs):
        return self._wave(self.L.lcrc_wave_to_posteriors, blobs, self.n_out)

    def reserve(self, max_rows, max_utts, max_wave_bytes=0):
        """lcrc_reserve: the buffers later calls of up to this size would allocate on demand, allocated now"""
        self._check(self.L.lcrc_reserve(self.h, int(max_rows), int(max_utts), int(max_wave_bytes)))

    def wave_to_posteriors_staged(self, blobs):
        """lcrc_wave_stage_buffer / lcrc_wave_stage_run: the files are written straight into the pinned buffer"""
        blobs = [bytes(b) for b in blobs]
        start, pos = [], 0
        for b in blobs:
            start.append(pos)
            pos += len(b) + (len(b) & 1)
        buf = C.POINTER(C.c_ubyte)()
        self._check(self.L.lcrc_wave_stage_buffer(self.h, pos, C.byref(buf)))
        for s, b in zip(start, blobs):
            C.memmove(C.addressof(buf.contents) + s, b, len(b))
        rows = sum(self.frontend_frames(len(b)) for b in blobs)
        out = np.zeros((rows, self.n_out), np.float32)
        foff = np.zeros(len(blobs) + 1, np.int32)
        self._check(self.L.lcrc_wave_stage_run(self.h, np.array(start, np.int64), np.array([len(b) for b in blobs], np.int64),
                                               len(blobs), out, foff))
        return out, foff

    def wave_decode_staged(self, blobs):
        """lcrc_wave_stage_run with post = NULL: the posteriors stay on the device (decoder configured, read-back off);
        returns the frame offsets.  Labels: last_labels(), or prev_labels() under set_decoder_overlap"""
        blobs = [bytes(b) for b in blobs]
        start, pos = [], 0
        for b in blobs:
            start.append(pos)
            pos += len(b) + (len(b) & 1)
        buf = C.POINTER(C.c_ubyte)()
        self._check(self.L.lcrc_wave_stage_buffer(self.h, max(pos, 16), C.byref(buf)))
        for s, b in zip(start, blobs):
            C.memmove(C.addressof(buf.contents) + s, b, len(b))
        foff = np.zeros(len(blobs) + 1, np.int32)
        i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
        fn = self.L.lcrc_wave_stage_run
        keep = fn.argtypes
        fn.argtypes = [C.c_void_p, i64p, i64p, C.c_int, C.c_void_p, _i32p]
        try:
            self._check(fn(self.h, np.array(start, np.int64).reshape(-1), np.array([len(b) for b in blobs], np.int64).reshape(-1),
                           len(blobs), None, foff))
        finally:
            fn.argtypes = keep
        return foff

    def wave_energies_staged(self, blobs):
        """lcrc_wave_stage_buffer / lcrc_wave_stage_energies: mel-bank energies (before ln) of raw files, as a COPY of the
        pinned feature buffer the call hands out; returns (energies [rows][nbanks], frame offsets)"""
        blobs = [bytes(b) for b in blobs]
        start, pos = [], 0
        for b in blobs:
            start.append(pos)
            pos += len(b) + (len(b) & 1)
        buf = C.POINTER(C.c_ubyte)()
        self._check(self.L.lcrc_wave_stage_buffer(self.h, pos, C.byref(buf)))
        for s, b in zip(start, blobs):
            C.memmove(C.addressof(buf.contents) + s, b, len(b))
        foff = np.zeros(len(blobs) + 1, np.int32)
        en = C.POINTER(C.c_float)()
        i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
        self.L.lcrc_wave_stage_energies.argtypes = [C.c_void_p, i64p, i64p, C.c_int, C.POINTER(C.POINTER(C.c_float)), _i32p]
        self._check(self.L.lcrc_wave_stage_energies(self.h, np.array(start, np.int64), np.array([len(b) for b in blobs], np.int64),
                                                    len(blobs), C.byref(en), foff))
        rows = int(foff[-1])
        out = np.ctypeslib.as_array(en, shape=(rows, self.nbanks)).copy() if rows else np.zeros((0, self.nbanks), np.float32)
        return out, foff

    def posteriors_probe(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        n = mel.shape[0]
        k, _, ob = self.net_dims(0)
        out = {"post": np.empty((n, self.n_out), np.float32),
               "in0": np.empty((n, k), np.float32), "in1": np.empty((n, k), np.float32),
               "p0": np.empty((n, ob), np.float32), "p1": np.empty((n, ob), np.float32),
               "g": np.empty((n, 2 * ob), np.float32)}
        self._check(self.L.lcrc_posteriors_probe(
            self.h, mel, n, out["post"], *[out[k_].ctypes.data for k_ in ("in0", "in1", "p0", "p1", "g")]))
        return out

    def posteriors_device(self, d_mel_ptr, n_rows, d_post_ptr, d_off_ptr=None, n_utts=1, stream=None):
        """Raw device pointers (ints); asynchronous on `stream` (int handle or None)."""
        self._check(self.L.lcrc_posteriors_device(self.h, d_mel_ptr, d_off_ptr, n_utts, n_rows,
                                                  d_post_ptr, stream))

    # -- streaming form (Traps::Reset / CalcFeaturesBunched / GetDelay) --
    def reset(self):
        self._check(self.L.lcrc_reset(self.h))

    def push(self, mel, needed=True):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        post = np.empty((mel.shape[0], self.n_out), np.float32) if needed else None
        self._check(self.L.lcrc_push(self.h, mel, mel.shape[0],
                                     post.ctypes.data if needed else None, int(needed)))
        return post

    def delay(self):
        return self.L.lcrc_delay(self.h)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.L.lcrc_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def set_tile_frames(self, frames):
        """0 = per launch (default), 16 or 32 = forced frames per workgroup"""
        self._check(self.L.lcrc_set_tile_frames(self.h, frames))

    def set_hidden_split(self, workgroups_per_tile):
        """0 = automatic (default), 1 = never split (bit-identical however frames are batched), k = at most k"""
        self._check(self.L.lcrc_set_hidden_split(self.h, workgroups_per_tile))

    def set_arithmetic(self, arithmetic):
        """ARITH_F32 (default) or ARITH_SPLIT_F16 (f32 products as three exact f16 MFMA products; shipped LCRC shapes)"""
        self._check(self.L.lcrc_set_arithmetic(self.h, arithmetic))

    def set_wait_mode(self, poll_interval_us):
        self._check(self.L.lcrc_set_wait_mode(self.h, int(poll_interval_us)))

    def set_timing(self, on):
        self._check(self.L.lcrc_set_timing(self.h, int(on)))
