"""Synthetic PhnRec model directories (reference on-disk layout, seeded weights).

The shipped PhnRec systems are research-licensed data that does not travel
with every checkout, and the benchmark contract allows random-init weights of
the named architecture.  This module writes a model directory that the
reference itself can load (and that oracle/_ref DOES load when goldens are
made): `config` (INI, srec.cpp:34-110 schema), `weights/{band0,band1,merger}.nbin`
(nn.cpp:464-531 layout), `windows/band{0,1}.window` (traps.cpp:549-570),
`dicts/phonemes`, optionally ASCII `.weights` / `.norms` (nn.cpp:116-412).

Shapes of the four shipped systems (read from their .nbin headers, SURVEY.md
Appendix A.2) are in SYSTEMS.
"""
import hashlib
import os

import numpy as np

# name -> dims of the LCRC system
SYSTEMS = {
    "PHN_CZ_SPDAT_LCRC_N1500": dict(nbanks=15, hidden=1500, n_out=138, sample_freq=8000,
                                    vector_size=200, vector_step=80, lower=64, higher=4000,
                                    sent_mean_norm=True, wpenalty=-4.6875, fmt="lin16", suffix="mel"),
    "PHN_HU_SPDAT_LCRC_N1500": dict(nbanks=15, hidden=1500, n_out=186, sample_freq=8000,
                                    vector_size=200, vector_step=80, lower=64, higher=4000,
                                    sent_mean_norm=True, wpenalty=-2.8125, fmt="lin16", suffix="mel"),
    "PHN_RU_SPDAT_LCRC_N1500": dict(nbanks=15, hidden=1400, n_out=159, sample_freq=8000,
                                    vector_size=200, vector_step=80, lower=64, higher=4000,
                                    sent_mean_norm=True, wpenalty=-0.9375, fmt="lin16", suffix="mel"),
    "PHN_EN_TIMIT_LCRC_N500": dict(nbanks=23, hidden=500, n_out=120, sample_freq=16000,
                                   vector_size=400, vector_step=160, lower=0, higher=8000,
                                   sent_mean_norm=False, wpenalty=-2.03125, fmt="lin16", suffix="fea"),
}

N_COEF = 11      # C0 + 10 DCT coefficients per band and half-context
HALF = 16
TRAP_LEN = 31


def pad4(n):
    return (n + 3) & ~3


def half_windows(half=HALF):
    """The two half-Hamming windows (16 taps in every shipped system; `half` taps for other posteriors/length values)."""
    j = np.arange(half, dtype=np.float64)
    w0 = 0.54 - 0.46 * np.cos(2.0 * np.pi * j / (2.0 * max(half - 1, 1)))
    return w0.astype(np.float32), w0[::-1].astype(np.float32).copy()


def random_net(rng, n_inp, n_hid, n_out, kind):
    """One MLP with activations in a realistic range (spread sigmoids, peaky softmax)."""
    w1 = rng.standard_normal((n_hid, n_inp)).astype(np.float32) * np.float32(1.5 / np.sqrt(n_inp))
    b1 = rng.standard_normal(n_hid).astype(np.float32) * np.float32(0.5)
    w2 = rng.standard_normal((n_out, n_hid)).astype(np.float32) * np.float32(4.0 / np.sqrt(n_hid))
    b2 = rng.standard_normal(n_out).astype(np.float32) * np.float32(0.5)
    if kind == "band":
        mean = rng.standard_normal(n_inp).astype(np.float32) * np.float32(0.5)
        dev = rng.uniform(0.3, 1.5, n_inp).astype(np.float32)
    elif kind == "dct31":   # C0 / DCT of a 31-point trajectory: a few times larger than the taps
        mean = rng.standard_normal(n_inp).astype(np.float32)
        dev = rng.uniform(0.05, 0.25, n_inp).astype(np.float32)
    elif kind == "neglog":  # 1BT / 3BT merger input = -ln(band posteriors)
        mean = rng.uniform(3.0, 9.0, n_inp).astype(np.float32)
        dev = rng.uniform(0.2, 0.5, n_inp).astype(np.float32)
    else:  # merger input = log posteriors
        mean = rng.uniform(-9.0, -3.0, n_inp).astype(np.float32)
        dev = rng.uniform(0.2, 0.5, n_inp).astype(np.float32)
    return dict(w1=w1, b1=b1, w2=w2, b2=b2, mean=mean, dev=dev)


def write_nbin(path, net):
    """nn.cpp:533-592: int32 2,nInp,nHid,nOut then padded W1,W2,b1,b2,mean,dev."""
    n_hid, n_inp = net["w1"].shape
    n_out = net["w2"].shape[0]
    i16, h16, o16 = pad4(n_inp), pad4(n_hid), pad4(n_out)
    w1 = np.zeros((h16, i16), np.float32)
    w1[:n_hid, :n_inp] = net["w1"]
    w2 = np.zeros((o16, h16), np.float32)
    w2[:n_out, :n_hid] = net["w2"]
    b1 = np.zeros(h16, np.float32)
    b1[:n_hid] = net["b1"]
    b2 = np.zeros(o16, np.float32)
    b2[:n_out] = net["b2"]
    mean = np.zeros(i16, np.float32)
    mean[:n_inp] = net["mean"]
    dev = np.ones(i16, np.float32)
    dev[:n_inp] = net["dev"]
    with open(path, "wb") as f:
        np.array([2, n_inp, n_hid, n_out], "<i4").tofile(f)
        for a in (w1, w2, b1, b2, mean, dev):
            a.astype("<f4").tofile(f)


def write_ascii(weights_path, norms_path, net):
    """nn.cpp:116-412 text form: weigvec/weigvec/biasvec/biasvec and vec/vec."""
    with open(weights_path, "w") as f:
        for key, tag in (("w1", "weigvec"), ("w2", "weigvec"), ("b1", "biasvec"), ("b2", "biasvec")):
            a = net[key].reshape(-1)
            f.write("%s %d\n" % (tag, a.size))
            f.write("\n".join("%.9e" % v for v in a))
            f.write("\n")
    with open(norms_path, "w") as f:
        for key in ("mean", "dev"):
            a = net[key]
            f.write("vec %d\n" % a.size)
            f.write("\n".join("%.9e" % v for v in a))
            f.write("\n")


def config_text(nbanks, sample_freq=8000, vector_size=200, vector_step=80, lower=64, higher=4000,
                sent_mean_norm=True, wpenalty=-4.6875, fmt="lin16", suffix="mel", bunch_size=5,
                system="LCRC", add_c0=True, hamming=False, trap_len=TRAP_LEN, sent_max_norm=False, sent_chmax_norm=False,
                **_unused):
    b = "true" if sent_mean_norm else "false"
    mx, cmx = ("true" if sent_max_norm else "false"), ("true" if sent_chmax_norm else "false")
    c0, hm = ("true" if add_c0 else "false"), ("true" if hamming else "false")
    return f"""[source]
format={fmt}
sample_freq={sample_freq}

[posteriors]
system={system}
length={trap_len}
add_c0={c0}
hamming={hm}
suffix=lop
bunch_size={bunch_size}
softening_func=none 0 0 0

[params]
kind=fbanks
suffix={suffix}

[melbanks]
nbanks={nbanks}
lower_freq={lower}
higher_freq={higher}
vector_size={vector_size}
vector_step={vector_step}
preem_coef=0.0

[decoder]
type=phndec
num_states_per_phn=3
softening_func=log 0 0 0
wpenalty={wpenalty}
lm_scale=1
time_pruning=40
mode=decode

[offlinenorm]
sent_mean_norm={b}
sent_var_norm=false
sent_max_norm={mx}
sent_chmax_norm={cmx}

[dirs]
tmp=$C/tmp

[models]
hmm_defs=$T/models
nstates=3
gen_from_phn_list=true

[dicts]
phoneme_list=$C/dicts/phonemes
lexicon1=none
lexicon1_save_bin=false
lexicon2=none
lexicon2_save_bin=false
keyword_list=none

[networks]
default=$C/net/network
gen_phn_loop=false
gen_kws_net=false
omit_phn=oth

[labels]
suffix=rec
remove_path=true

[kws]
default_thr=-15
thresholds_file=none
"""


def write_model_dir(path, nbanks, hidden, n_out, seed=0, ascii_too=False, nbin=True,
                    hidden_merger=None, coefs=N_COEF, **cfg):
    """Write a loadable LCRC model directory; returns the dict of nets.  The shipped geometry unless told otherwise:
    `coefs` inputs per band of the band nets (C0 included when add_c0), trap_len / add_c0 through the config."""
    rng = np.random.default_rng(seed)
    k = nbanks * coefs
    nets = {
        "band0": random_net(rng, k, hidden, n_out, "band"),
        "band1": random_net(rng, k, hidden, n_out, "band"),
        "merger": random_net(rng, 2 * n_out, hidden_merger or hidden, n_out, "merger"),
    }
    for sub in ("weights", "norms", "windows", "dicts", "tmp", "net"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    for name, net in nets.items():
        if nbin:
            write_nbin(os.path.join(path, "weights", name + ".nbin"), net)
        if ascii_too or not nbin:
            write_ascii(os.path.join(path, "weights", name + ".weights"),
                        os.path.join(path, "norms", name + ".norms"), net)
    for i, w in enumerate(half_windows((cfg.get("trap_len", TRAP_LEN) - 1) // 2 + 1)):
        with open(os.path.join(path, "windows", "band%d.window" % i), "w") as f:
            f.write(" ".join("%.7e" % v for v in w) + "\n")
    n_phn = n_out // 3 - 1
    with open(os.path.join(path, "dicts", "phonemes"), "w") as f:
        f.write("".join("p%02d\n" % i for i in range(n_phn)))
    with open(os.path.join(path, "config"), "w") as f:
        f.write(config_text(nbanks, **cfg))
    return nets


def write_traps_dir(path, system, nbanks, hidden, n_out, seed=0, band_out=12, band_hidden=40, coefs=6,
                    add_c0=True, hamming=False, trap_len=TRAP_LEN, **cfg):
    """Write a loadable model directory for the non-LCRC `posteriors/system` variants (traps.cpp:88-171):
    "1BT_DCT": merger over nbanks * coefs DCT features (coefs counts C0 when add_c0);
    "1BT" / "3BT": nbanks (nbanks - 2) band nets 31 -> band_hidden -> band_out and a merger over their outputs."""
    rng = np.random.default_rng(seed)
    nets = {}
    if system == "1BT_DCT":
        nets["merger"] = random_net(rng, nbanks * coefs, hidden, n_out, "dct31")
    else:
        tb = nbanks - 2 if system == "3BT" else nbanks
        for i in range(tb):
            nets["band%d" % i] = random_net(rng, trap_len, band_hidden + i, band_out, "band")
        nets["merger"] = random_net(rng, tb * band_out, hidden, n_out, "neglog")
    for sub in ("weights", "norms", "windows", "dicts", "tmp", "net"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    for name, net in nets.items():
        write_nbin(os.path.join(path, "weights", name + ".nbin"), net)
    n_phn = n_out // 3 - 1
    with open(os.path.join(path, "dicts", "phonemes"), "w") as f:
        f.write("".join("p%02d\n" % i for i in range(n_phn)))
    with open(os.path.join(path, "config"), "w") as f:
        f.write(config_text(nbanks, system=system, add_c0=add_c0, hamming=hamming, trap_len=trap_len, **cfg))
    return nets


def write_system(path, system, seed=0, **kw):
    """Write a synthetic model dir with the dims of one of the shipped systems."""
    s = dict(SYSTEMS[system])
    s.update(kw)
    return write_model_dir(path, s.pop("nbanks"), s.pop("hidden"), s.pop("n_out"), seed=seed, **s)


def nets_digest(nets):
    h = hashlib.sha256()
    for name in sorted(nets):
        for key in sorted(nets[name]):
            h.update(np.ascontiguousarray(nets[name][key]).tobytes())
    return h.hexdigest()


def synth_mel(n, nbanks, seed=0, mean_norm=True):
    """Log-mel-like frames: smooth trajectories around 8..15 (cf. real test.raw dumps)."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(9.0, 14.0, nbanks)
    x = rng.standard_normal((n + 16, nbanks))
    kern = np.hanning(9)
    kern /= kern.sum()
    sm = np.stack([np.convolve(x[:, b], kern, mode="valid") for b in range(nbanks)], axis=1)[:n]
    mel = (base + 4.0 * sm + 0.3 * rng.standard_normal((n, nbanks))).astype(np.float32)
    if mean_norm:
        mel = mel - mel.mean(axis=0, dtype=np.float32)
    return np.ascontiguousarray(mel, dtype=np.float32)
