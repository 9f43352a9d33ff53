"""CPU-only checks of the product side: the C-ABI library loads, exports every symbol
include/lcrc.h declares, reads the reference's model formats, and refuses to compute
without a GPU (no fallback).  No kernel is launched here."""
import os
import re
import subprocess

import numpy as np
import pytest

from phnrec_amd import capi, modelgen
from tests.util import ROOT, model_dir


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "lcrc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lcrc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), "libphnrec_lcrc.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS out of sync with include/lcrc.h"
    assert L.lcrc_abi_version() == 4      # include/lcrc.h LCRC_ABI_VERSION
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (lcrc_[a-z_0-9]+)", out))
    assert set(declared) <= exported


def test_library_has_gfx950_code_object_and_no_torch_types():
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    out = subprocess.check_output(["nm", "-D", "-C", capi.LIB_PATH]).decode()
    assert "at::" not in out and "c10::" not in out, "torch types leaked into the C ABI library"
    needed = subprocess.check_output(["readelf", "-d", capi.LIB_PATH]).decode()
    assert "libamdhip64" not in needed, "the HIP runtime must come from the host process (see csrc/Makefile)"


@pytest.mark.parametrize("system,kernel", [("PHN_CZ_SPDAT_LCRC_N1500", "cz_42_69_9"),
                                           ("PHN_HU_SPDAT_LCRC_N1500", "hu_42_93_12"),
                                           ("PHN_RU_SPDAT_LCRC_N1500", "ru_42_80_10"),
                                           ("PHN_EN_TIMIT_LCRC_N500", "en_64_60_8")])
def test_model_info_on_shipped_models(system, kernel):
    spec = modelgen.SYSTEMS[system]
    info = capi.model_info(model_dir(system), spec["nbanks"])
    k = spec["nbanks"] * 11
    assert info["dims"] == [(k, spec["hidden"], spec["n_out"])] * 2 + \
        [(2 * spec["n_out"], spec["hidden"], spec["n_out"])]
    assert info["kernel"] == kernel
    assert info["lds_bytes"] <= 160 * 1024


def test_kernel_variants_for_all_shipped_shapes(tmp_path):
    want = {"PHN_CZ_SPDAT_LCRC_N1500": "cz_42_69_9", "PHN_HU_SPDAT_LCRC_N1500": "hu_42_93_12",
            "PHN_RU_SPDAT_LCRC_N1500": "ru_42_80_10", "PHN_EN_TIMIT_LCRC_N500": "en_64_60_8"}
    for system, kernel in want.items():
        d = str(tmp_path / system)
        modelgen.write_system(d, system, seed=1)
        info = capi.model_info(d, modelgen.SYSTEMS[system]["nbanks"])
        assert info["kernel"] == kernel and info["lds_bytes"] <= 160 * 1024, (system, info)
    d = str(tmp_path / "odd")
    modelgen.write_model_dir(d, 20, 77, 50, seed=2)
    assert capi.model_info(d, 20)["kernel"].startswith("generic_")


def test_ascii_and_nbin_models_load_identically(tmp_path, oracle_mod):
    """ASCII .weights/.norms (nn.cpp:116-412) vs .nbin (nn.cpp:464-531), and the .nbin cache
    NeuralNet::Load writes next to ASCII weights (nn.cpp:613-618)"""
    a, b = str(tmp_path / "ascii"), str(tmp_path / "bin")
    modelgen.write_model_dir(a, 15, 40, 18, seed=3, nbin=False)
    modelgen.write_model_dir(b, 15, 40, 18, seed=3)
    assert not os.path.exists(os.path.join(a, "weights", "band0.nbin"))
    ia, ib = capi.model_info(a, 15), capi.model_info(b, 15)
    assert ia == ib
    for name in ("band0", "band1", "merger"):
        cached = os.path.join(a, "weights", name + ".nbin")
        assert os.path.exists(cached), "ASCII load should leave the .nbin cache like the reference"
        # "%.9e" text round-trips float32 exactly, so the cache equals the direct binary
        assert open(cached, "rb").read() == open(os.path.join(b, "weights", name + ".nbin"), "rb").read()
        # and the oracle's own reader sees the same net
        n = oracle_mod.Net(nbin=cached)
        assert n.dims == tuple(ia["dims"][("band0", "band1", "merger").index(name)])


def test_model_errors_without_gpu(tmp_path):
    with pytest.raises(capi.LcrcError) as e:
        capi.model_info(str(tmp_path / "missing"), 15)
    assert e.value.code == capi.LCRC_E_IO
    d = str(tmp_path / "m")
    modelgen.write_model_dir(d, 15, 32, 12, seed=1)
    with pytest.raises(capi.LcrcError) as e:
        capi.model_info(d, 23)
    assert e.value.code == capi.LCRC_E_MODEL
    os.remove(os.path.join(d, "windows", "band1.window"))
    with pytest.raises(capi.LcrcError) as e:
        capi.model_info(d, 15)
    assert e.value.code == capi.LCRC_E_IO and "window" in str(e.value)
    # truncated .nbin
    d2 = str(tmp_path / "m2")
    modelgen.write_model_dir(d2, 15, 32, 12, seed=1)
    p = os.path.join(d2, "weights", "merger.nbin")
    blob = open(p, "rb").read()
    open(p, "wb").write(blob[:len(blob) // 2])
    with pytest.raises(capi.LcrcError) as e:
        capi.model_info(d2, 15)
    assert e.value.code == capi.LCRC_E_MODEL
    # too many outputs for the kernel (> 208)
    d3 = str(tmp_path / "m3")
    modelgen.write_model_dir(d3, 15, 32, 240, seed=1)
    with pytest.raises(capi.LcrcError) as e:
        capi.model_info(d3, 15)
    assert e.value.code == capi.LCRC_E_UNSUPPORTED


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.LcrcError) as e:
        capi.Lcrc(model_dir("PHN_CZ_SPDAT_LCRC_N1500"), 15)
    assert e.value.code == capi.LCRC_E_DEVICE
    assert capi.load().lcrc_device_warmup(0) == capi.LCRC_E_DEVICE          # the start-up helper says so too
    import ctypes
    buf = ctypes.create_string_buffer(64)
    assert capi.load().lcrc_device_pci_bus_id(0, buf, 64) == capi.LCRC_E_DEVICE and buf.value == b""
    # and nothing in the product imports, links or loads anything under oracle/
    pat = re.compile(r"from\s+oracle|import\s+oracle|lcrc_oracle|liblcrc_oracle|libphnrec_ref|orc_[a-z_]+\(")
    for root, _, files in os.walk(os.path.join(ROOT, "phnrec_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert not pat.search(txt), "%s reaches into oracle/: the product may not depend on it" % f


def test_header_is_plain_c(tmp_path):
    """include/lcrc.h is the boundary a C / cgo / JNI binding would include: it must compile as C99 on its
    own, and a C program that only takes the addresses of all entry points must link against the library"""
    import re
    import subprocess
    hdr = os.path.join(ROOT, "include", "lcrc.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                   check=True)
    names = sorted(set(re.findall(r"\b(lcrc_[a-z_0-9]+)\s*\(", open(hdr).read())) - {"lcrc_ctx"})
    src = tmp_path / "link.c"
    src.write_text('#include "lcrc.h"\n#include <stdio.h>\nint main(void) {\n  const void *p[] = {%s};\n'
                   '  printf("%%d %%d\\n", (int)(sizeof p / sizeof p[0]), lcrc_abi_version());\n  return 0;\n}\n'
                   % ", ".join("(const void *)%s" % n for n in names))
    exe = tmp_path / "link"
    lib = os.path.join(ROOT, "phnrec_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib,
                    "-lphnrec_lcrc", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == len(names) >= 30 and int(out[1]) >= 1


def test_device_ln_argument_checks_need_no_gpu():
    """lcrc_device_ln validates its arguments before it touches a device: an unknown form, a negative count or missing arrays
    are LCRC_E_ARG, an empty array is nothing to do; with values to compute it needs the GPU and says so (no CPU fallback)"""
    x = np.array([1.0, 2.0, 0.0, -1.0], np.float32)
    with pytest.raises(capi.LcrcError) as e:
        capi.device_ln(x, 7)
    assert e.value.code == capi.LCRC_E_ARG
    assert capi.device_ln(np.zeros(0, np.float32), 1).size == 0
    L = capi.load()
    import ctypes as C
    L.lcrc_device_ln.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong]
    assert L.lcrc_device_ln(0, 1, None, None, 4) == capi.LCRC_E_ARG and L.lcrc_device_ln(0, 1, None, None, -1) == capi.LCRC_E_ARG
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(capi.LcrcError) as e:
            capi.device_ln(x, 1)
        assert e.value.code == capi.LCRC_E_DEVICE
