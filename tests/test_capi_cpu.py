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


HEADERS = ("lcrc.h", "lcrc_pipeline.h", "lcrc_experimental.h")      # the Traps seam; the "next" rows + list helpers; hooks


def _declared_functions(headers=HEADERS):
    names = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(lcrc_[a-z_0-9]+)\s*\(", txt))
    return sorted(names - {"lcrc_kernel_done_fn"})


def test_library_exports_every_declared_symbol():
    L = capi.load()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), "libphnrec_lcrc.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS out of sync with include/*.h"
    assert L.lcrc_abi_version() == 5      # include/lcrc.h LCRC_ABI_VERSION
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == sorted(HEADERS)
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


def test_headers_are_plain_c(tmp_path):
    """include/*.h is the boundary a C / cgo / JNI binding would include: each header must compile as C99 on its
    own, and a C program that only takes the addresses of all entry points must link against the library"""
    import subprocess
    for h in HEADERS:
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", h)], check=True)
    names = _declared_functions()
    src = tmp_path / "link.c"
    src.write_text('#include "lcrc_experimental.h"\n#include <stdio.h>\nint main(void) {\n  const void *p[] = {%s};\n'
                   '  printf("%%d %%d\\n", (int)(sizeof p / sizeof p[0]), lcrc_abi_version());\n  return 0;\n}\n'
                   % ", ".join("(const void *)%s" % n for n in names))
    exe = tmp_path / "link"
    lib = os.path.join(ROOT, "phnrec_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib,
                    "-lphnrec_lcrc", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == len(names) >= 30 and int(out[1]) >= 1


def test_core_header_is_the_traps_seam_and_small():
    """lcrc.h alone is what INTEGRATION.md's two adoption routes call (and the reference-side binding compiles against it
    alone): at most 250 lines, none of the tuning / test hooks, none of the list pipeline's helpers"""
    core = _declared_functions(("lcrc.h",))
    txt = open(os.path.join(ROOT, "include", "lcrc.h")).read()
    assert len(txt.splitlines()) <= 250
    for must in ("lcrc_create", "lcrc_create_system", "lcrc_destroy", "lcrc_reset", "lcrc_push", "lcrc_delay", "lcrc_num_outputs",
                 "lcrc_posteriors", "lcrc_posteriors_batch", "lcrc_last_error", "lcrc_set_arithmetic"):
        assert must in core, must
    for moved in ("lcrc_set_tile_frames", "lcrc_set_hidden_split", "lcrc_debug_fail_alloc", "lcrc_debug_fail_launch",
                  "lcrc_posteriors_probe", "lcrc_set_wait_mode", "lcrc_set_mean_order", "lcrc_set_decoder_overlap",
                  "lcrc_set_launch_order", "lcrc_wave_stage_run", "lcrc_reserve"):
        assert moved not in core, moved
    hooks = _declared_functions(("lcrc_experimental.h",))
    assert set(hooks) >= {"lcrc_set_tile_frames", "lcrc_set_hidden_split", "lcrc_debug_fail_alloc", "lcrc_posteriors_probe",
                          "lcrc_set_mean_order"}
    # the binding of INTEGRATION.md includes lcrc.h and nothing else of this repository's headers
    binding = open(os.path.join(ROOT, "tests", "integration", "traps_lcrc.cpp")).read()
    assert '#include "lcrc.h"' in binding and "lcrc_pipeline.h" not in binding and "lcrc_experimental.h" not in binding
