"""ctypes bindings for the CHECKERS under oracle/ (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product (phnrec_amd/) never does.

  Oracle      -- liblcrc_oracle.so, this repo's C restatement (lcrc_oracle.c)
  RefTraps    -- oracle/_ref/libphnrec_ref[_blas].so, the REAL reference's
                 Traps class behind ref_shim.cpp (present when oracle/Makefile
                 was run with /root/reference available; the built .so travels
                 to the GPU box, the sources do not)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblcrc_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile the checkers (oracle always; _ref only if /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all" if ref else "oracle"])


class _Net(C.Structure):
    _fields_ = [("nInp", C.c_int), ("nHid", C.c_int), ("nOut", C.c_int),
                ("nInp16", C.c_int), ("nHid16", C.c_int), ("nOut16", C.c_int),
                ("W1", C.POINTER(C.c_float)), ("W2", C.POINTER(C.c_float)),
                ("b1", C.POINTER(C.c_float)), ("b2", C.POINTER(C.c_float)),
                ("mean", C.POINTER(C.c_float)), ("dev", C.POINTER(C.c_float))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.orc_fexp.restype = C.c_float
        L.orc_fexp.argtypes = [C.c_float]
        L.orc_fexp_sigmoid.restype = C.c_float
        L.orc_fexp_sigmoid.argtypes = [C.c_float]
        L.orc_fexp_softmax.argtypes = [C.c_int, _f32p]
        L.orc_net_load_nbin.argtypes = [C.POINTER(_Net), C.c_char_p]
        L.orc_net_load_ascii.argtypes = [C.POINTER(_Net), C.c_char_p, C.c_char_p]
        L.orc_net_load.argtypes = [C.POINTER(_Net), C.c_char_p, C.c_char_p, C.c_int]
        L.orc_net_save_nbin.argtypes = [C.POINTER(_Net), C.c_char_p]
        L.orc_net_free.argtypes = [C.POINTER(_Net)]
        L.orc_net_forward.argtypes = [C.POINTER(_Net), _f32p, _f32p, C.c_int]
        L.orc_net_forward_probe.argtypes = [C.POINTER(_Net), _f32p, _f32p, _f32p, C.c_int]
        L.orc_lcrc_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.orc_lcrc_destroy.argtypes = [C.c_void_p]
        L.orc_lcrc_num_outputs.argtypes = [C.c_void_p]
        L.orc_lcrc_num_inputs.argtypes = [C.c_void_p]
        L.orc_lcrc_net.argtypes = [C.c_void_p, C.c_int]
        L.orc_lcrc_net.restype = C.POINTER(_Net)
        L.orc_lcrc_project.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        L.orc_lcrc_posteriors.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p]
        L.orc_lcrc_posteriors_mt.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int]
        L.orc_lcrc_posteriors_probe.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p] + [C.c_void_p] * 5
        L.orc_lcrc_posteriors_batch.argtypes = [C.c_void_p, _f32p, _i32p, C.c_int, _f32p]
        L.orc_lcrc_reset.argtypes = [C.c_void_p]
        L.orc_lcrc_push.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_lcrc_delay.argtypes = [C.c_void_p]
        L.orc_lcrc_process_offline.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int]
        L.orc_sentence_mean_norm.argtypes = [_f32p, C.c_int, C.c_int]
        _lib = L
    return _lib


def fexp(y):
    return float(lib().orc_fexp(float(np.float32(y))))


def fexp_sigmoid(x):
    return float(lib().orc_fexp_sigmoid(float(np.float32(x))))


def fexp_softmax(v):
    v = np.ascontiguousarray(v, dtype=np.float32).copy()
    lib().orc_fexp_softmax(len(v), v)
    return v


def sentence_mean_norm(mel):
    mel = np.ascontiguousarray(mel, dtype=np.float32).copy()
    lib().orc_sentence_mean_norm(mel, mel.shape[0], mel.shape[1])
    return mel


class Net:
    """One MLP loaded by the oracle's own loaders."""

    def __init__(self, nbin=None, weights=None, norms=None):
        self.n = _Net()
        if nbin is not None:
            rc = lib().orc_net_load_nbin(C.byref(self.n), nbin.encode())
        else:
            rc = lib().orc_net_load_ascii(C.byref(self.n), weights.encode(),
                                          norms.encode() if norms else None)
        if rc != 0:
            raise IOError("oracle net load failed rc=%d" % rc)

    def __del__(self):
        try:
            lib().orc_net_free(C.byref(self.n))
        except Exception:
            pass

    @property
    def dims(self):
        return self.n.nInp, self.n.nHid, self.n.nOut

    def array(self, name):
        n = self.n
        shape = {"W1": (n.nHid16, n.nInp16), "W2": (n.nOut16, n.nHid16), "b1": (n.nHid16,),
                 "b2": (n.nOut16,), "mean": (n.nInp16,), "dev": (n.nInp16,)}[name]
        return np.ctypeslib.as_array(getattr(n, name), shape=shape).copy()

    def forward(self, x, probe=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((x.shape[0], self.n.nOut), np.float32)
        if probe:
            hid = np.empty((x.shape[0], self.n.nHid), np.float32)
            lib().orc_net_forward_probe(C.byref(self.n), x, out, hid, x.shape[0])
            return out, hid
        lib().orc_net_forward(C.byref(self.n), x, out, x.shape[0])
        return out

    def save_nbin(self, path):
        rc = lib().orc_net_save_nbin(C.byref(self.n), path.encode())
        if rc != 0:
            raise IOError("save_nbin rc=%d" % rc)


class Oracle:
    """The LCRC estimator as restated in lcrc_oracle.c."""

    def __init__(self, model_dir, nbanks):
        self.h = C.c_void_p()
        rc = lib().orc_lcrc_create(C.byref(self.h), model_dir.encode(), nbanks)
        if rc != 0:
            raise IOError("orc_lcrc_create(%s) rc=%d" % (model_dir, rc))
        self.nbanks = nbanks
        self.n_out = lib().orc_lcrc_num_outputs(self.h)
        self.n_in = lib().orc_lcrc_num_inputs(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().orc_lcrc_destroy(self.h)
        except Exception:
            pass

    def net_dims(self, which):
        n = lib().orc_lcrc_net(self.h, which).contents
        return n.nInp, n.nHid, n.nOut

    def project(self, ctx):
        ctx = np.ascontiguousarray(ctx, dtype=np.float32)
        a = np.empty(self.n_in, np.float32)
        b = np.empty(self.n_in, np.float32)
        lib().orc_lcrc_project(self.h, ctx, a, b)
        return a, b

    def posteriors(self, mel, threads=1):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        post = np.empty((mel.shape[0], self.n_out), np.float32)
        if threads > 1:
            lib().orc_lcrc_posteriors_mt(self.h, mel, mel.shape[0], post, threads)
        else:
            lib().orc_lcrc_posteriors(self.h, mel, mel.shape[0], post)
        return post

    def posteriors_probe(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        n = mel.shape[0]
        ob = self.net_dims(0)[2]
        out = {"post": np.empty((n, self.n_out), np.float32),
               "in0": np.empty((n, self.n_in), np.float32),
               "in1": np.empty((n, self.n_in), np.float32),
               "p0": np.empty((n, ob), np.float32),
               "p1": np.empty((n, ob), np.float32),
               "g": np.empty((n, 2 * ob), np.float32)}
        lib().orc_lcrc_posteriors_probe(self.h, mel, n, out["post"],
                                        *[out[k].ctypes.data for k in ("in0", "in1", "p0", "p1", "g")])
        return out

    def posteriors_batch(self, mel, off):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        post = np.zeros((mel.shape[0], self.n_out), np.float32)
        lib().orc_lcrc_posteriors_batch(self.h, mel, off, len(off) - 1, post)
        return post

    def reset(self):
        lib().orc_lcrc_reset(self.h)

    def push(self, mel, needed=True, bunch=5):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        post = np.empty((mel.shape[0], self.n_out), np.float32) if needed else None
        lib().orc_lcrc_push(self.h, mel, mel.shape[0],
                            post.ctypes.data if needed else None, int(needed), bunch)
        return post

    def delay(self):
        return lib().orc_lcrc_delay(self.h)

    def process_offline(self, mel, bunch=5):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        post = np.empty((mel.shape[0], self.n_out), np.float32)
        lib().orc_lcrc_process_offline(self.h, mel, mel.shape[0], post, bunch)
        return post


def phndec(logpost, n_phonemes, states=3, time_pruning=40, wpenalty=0.0):
    """phndec_oracle.c: labels [(start, end, phn, score)] of one utterance of log-posteriors"""
    L = lib()
    L.orc_phndec.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                             _i32p, _i32p, _i32p, _f32p]
    lp = np.ascontiguousarray(logpost, dtype=np.float32)
    T = lp.shape[0]
    cols = lp.shape[1] if lp.ndim == 2 else 0
    st, en, ph = (np.zeros(max(T, 1), np.int32) for _ in range(3))
    sc = np.zeros(max(T, 1), np.float32)
    n = L.orc_phndec(lp.reshape(-1) if T else np.zeros(1, np.float32), T, cols, n_phonemes, states,
                     time_pruning, wpenalty, st, en, ph, sc)
    return [(int(st[i]), int(en[i]), int(ph[i]), float(sc[i])) for i in range(n)]


class TrapsOracle:
    """traps_oracle.c: the 1BT_DCT / 1BT / 3BT variants of Traps at any posteriors/length, and LCRC with run-time
    geometry (any length, add_c0 on or off, any number of coefficients per band) -- stateless whole-utterance form."""

    def __init__(self, model_dir, system, nbanks, add_c0=True, hamming=False, trap_len=31):
        L = lib()
        L.orc_traps_create_geometry.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_traps_destroy.argtypes = [C.c_void_p]
        L.orc_traps_num_outputs.argtypes = [C.c_void_p]
        L.orc_traps_num_band_nets.argtypes = [C.c_void_p]
        L.orc_traps_posteriors_batch.argtypes = [C.c_void_p, _f32p, _i32p, C.c_int, _f32p, C.c_void_p]
        self.L = L
        self.h = C.c_void_p()
        rc = L.orc_traps_create_geometry(C.byref(self.h), model_dir.encode(), system.encode(), nbanks, int(add_c0),
                                         int(hamming), int(trap_len))
        if rc:
            raise OSError("orc_traps_create_geometry(%s, %s, length %d) -> %d" % (model_dir, system, trap_len, rc))
        self.nbanks = nbanks
        self.n_out = L.orc_traps_num_outputs(self.h)
        self.n_band_nets = L.orc_traps_num_band_nets(self.h)

    def __del__(self):
        try:
            self.L.orc_traps_destroy(self.h)
        except Exception:
            pass

    def posteriors_batch(self, mel, off):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        off = np.ascontiguousarray(off, dtype=np.int32)
        post = np.zeros((mel.shape[0], self.n_out), np.float32)
        self.L.orc_traps_posteriors_batch(self.h, mel, off, len(off) - 1, post, None)
        return post

    def posteriors(self, mel):
        return self.posteriors_batch(mel, np.array([0, len(mel)], np.int32))


def ref_lib_path(blas=False):
    p = os.path.join(REF_DIR, "libphnrec_ref_blas.so" if blas else "libphnrec_ref.so")
    return p if os.path.exists(p) else None


def ref_cli_path(blas=False):
    p = os.path.join(REF_DIR, "phnrec_ref_blas" if blas else "phnrec_ref")
    return p if os.path.exists(p) else None


_ref_libs = {}


def _ref(blas):
    if blas not in _ref_libs:
        p = ref_lib_path(blas)
        if p is None:
            raise FileNotFoundError("oracle/_ref not built (needs /root/reference at build time)")
        if blas:
            # libmkl_rt picks its threading layer at first use; next to PyTorch's OpenMP
            # runtime the default (Intel OpenMP) layer returns NaNs.  The reference is a
            # single-threaded program, so sequential MKL is also the faithful baseline.
            os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
            os.environ.setdefault("MKL_NUM_THREADS", "1")
        L = C.CDLL(p)
        L.refshim_traps_create.restype = C.c_void_p
        L.refshim_traps_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.refshim_traps_create_system.restype = C.c_void_p
        L.refshim_traps_create_system.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refshim_traps_create_geometry.restype = C.c_void_p
        L.refshim_traps_create_geometry.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refshim_traps_destroy.argtypes = [C.c_void_p]
        L.refshim_traps_reset.argtypes = [C.c_void_p]
        L.refshim_traps_num_outs.argtypes = [C.c_void_p]
        L.refshim_traps_delay.argtypes = [C.c_void_p]
        L.refshim_traps_calc_bunched.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int]
        L.refshim_traps_probe.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int]
        L.refshim_traps_process_offline.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p]
        L.refshim_nn_load.restype = C.c_void_p
        L.refshim_nn_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.refshim_nn_destroy.argtypes = [C.c_void_p]
        L.refshim_nn_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.refshim_nn_forward.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int]
        _ref_libs[blas] = L
    return _ref_libs[blas]


class RefTraps:
    """The reference's own Traps object (traps.h:59-75) via ref_shim.cpp."""

    def __init__(self, model_dir, nbanks, bunch=5, blas=False, system="LCRC", add_c0=True, hamming=False, trap_len=31):
        self.L = _ref(blas)
        self.nbanks = nbanks
        self.bunch = bunch
        d = model_dir if model_dir.endswith("/") else model_dir + "/"     # Traps::Init concatenates dir + "weights/..."
        self.h = self.L.refshim_traps_create_geometry(d.encode(), system.encode(), nbanks, bunch,
                                                      int(add_c0), int(hamming), int(trap_len))
        if not self.h:
            raise ValueError("unknown posteriors/system " + system)
        self.n_out = self.L.refshim_traps_num_outs(self.h)

    def __del__(self):
        try:
            self.L.refshim_traps_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        self.L.refshim_traps_reset(self.h)

    def delay(self):
        return self.L.refshim_traps_delay(self.h)

    def calc_bunched(self, mel, needed=True):
        mel = np.ascontiguousarray(mel, dtype=np.float32).reshape(-1, self.nbanks)
        post = np.zeros((mel.shape[0], self.n_out), np.float32)
        self.L.refshim_traps_calc_bunched(self.h, mel, post, mel.shape[0], int(needed))
        return post

    def probe(self, which, nframes, width):
        dst = np.zeros((nframes, width), np.float32)
        w = self.L.refshim_traps_probe(self.h, which, dst, nframes)
        assert w == width, (w, width)
        return dst

    def process_offline(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        post = np.zeros((mel.shape[0], self.n_out), np.float32)
        self.L.refshim_traps_process_offline(self.h, mel, mel.shape[0], self.nbanks, post)
        return post


class RefNet:
    """The reference's own NeuralNet (nn.h:48-56) via ref_shim.cpp."""

    def __init__(self, weights, norms=None, bunch=5, blas=False):
        self.L = _ref(blas)
        rc = C.c_int(0)
        self.h = self.L.refshim_nn_load(weights.encode(), norms.encode() if norms else None,
                                        bunch, C.byref(rc))
        self.rc = rc.value
        if not self.h:
            raise IOError("NeuralNet::Load rc=%d" % rc.value)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.refshim_nn_dims(self.h, C.byref(a), C.byref(b), C.byref(c))
        self.dims = (a.value, b.value, c.value)

    def __del__(self):
        try:
            self.L.refshim_nn_destroy(self.h)
        except Exception:
            pass

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros((x.shape[0], self.dims[2]), np.float32)
        self.L.refshim_nn_forward(self.h, x, out, x.shape[0])
        return out
