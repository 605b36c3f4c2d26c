"""CPU tests of the drop-in boundary: libfo1hip.so loads, exports every symbol
include/fo1.h declares, and rejects bad arguments with the documented error codes
(no compute without a GPU)."""
import os
import re

import pytest

from vlm_fo1_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="fo1.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fo1_[a-z0-9_]+)\s*\(", txt)))


def exported(so):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "vlm_fo1_amd", so)], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("fo1_") and " T " in l})


def all_defined_dynamic_symbols(so):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "vlm_fo1_amd", so)], capture_output=True, text=True, check=True).stdout
    return sorted({l.split()[-1] for l in out.splitlines() if l.strip()})


def test_export_table_is_exactly_the_headers():
    """nothing but the declared C entry points leaves the libraries — no un-prefixed helper, no mangled fo1:: internal,
    no kernel stub or kernel-handle object (-fvisibility=hidden + the headers' visibility pragma + csrc/exports.map)."""
    prod, ab = declared_symbols(), declared_symbols("fo1_ab.h")
    assert all_defined_dynamic_symbols("libfo1hip.so") == prod
    assert all_defined_dynamic_symbols("libfo1hip_ab.so") == sorted(prod + ab)


def test_library_exports_every_declared_symbol():
    lib = L.load()              # the test session's library: the PRODUCT build (tests/conftest.py)
    assert lib._name.endswith("libfo1hip.so")
    syms = declared_symbols()
    assert "fo1_hfre_region_pool" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/fo1.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(L.SIGNATURES) == syms
    assert sorted(L.SIGNATURES_AB) == declared_symbols("fo1_ab.h")
    with L.use_ab() as ab:      # the test / bench build exports both headers; the switch routes L.load() and restores it afterwards
        assert L.load() is ab and L.ab_build() and ab._name.endswith("libfo1hip_ab.so")
        for sname in syms + sorted(L.SIGNATURES_AB):
            assert hasattr(ab, sname)
    assert L.load() is lib and not L.ab_build()


def test_product_library_has_no_switches_and_the_ab_build_has_exactly_the_two_headers():
    """the product library's export table IS include/fo1.h — none of the process-global A/B, ablation or
    tuning switches (include/fo1_ab.h) exist in it; the test / bench build exports both headers and nothing else."""
    prod, ab = declared_symbols(), declared_symbols("fo1_ab.h")
    assert not set(prod) & set(ab)
    assert exported("libfo1hip.so") == prod
    assert exported("libfo1hip_ab.so") == sorted(prod + ab)
    # fo1_ab.h = the process-global switches (`_set_`) + the measured no-gain kernel forms and instruments moved out of the product ABI in round 5
    extra = {"fo1_gemm_bf16_wtiled", "fo1_splitk_swiglu_bf16", "fo1_mfma_clock_probe", "fo1_gemm_profile_shapes", "fo1_traffic_probe"}
    assert all("_set_" in s or s in extra for s in ab) and extra <= set(ab) and not any("_set_" in s for s in prod)
    # the product build also leaves the measured-slower kernel forms out
    import subprocess
    for so, want in (("libfo1hip.so", False), ("libfo1hip_ab.so", True)):
        blob = open(os.path.join(ROOT, "vlm_fo1_amd", so), "rb").read()
        for kernel in (b"gemm_bt_p8_kernel", b"gemm_bt_p4p_kernel", b"attn_decode_wg_kernel", b"gemv_batch_kernel", b"hfre_pool_kernel", b"hfre_pool_bands_kernel"):
            assert (kernel in blob) == want, f"{kernel.decode()} {'missing from' if want else 'present in'} {so}"


def test_abi_version():
    assert L.load().fo1_abi_version() == 9


def test_hfre_argument_errors():
    lib = L.load()
    S = L.HfreSource
    ok = S(16, 120, 160, 256, 256, 120, 160, 0.25, 0, 0)
    arr = (S * 1)(ok)
    assert lib.fo1_hfre_workspace_bytes(arr, 1, 100) > 0
    # NULL boxes
    assert lib.fo1_hfre_region_pool(arr, 1, None, 3, None, 1.0, 1.0, 7, 0, 1.0, 1.0, None, 256, 256, None, 0, None) == -1
    assert b"NULL" in lib.fo1_last_error()
    # channel count not a multiple of 64
    bad = (S * 1)(S(16, 120, 160, 100, 104, 120, 160, 0.25, 0, 0))
    assert lib.fo1_hfre_region_pool(bad, 1, 16, 3, None, 1.0, 1.0, 7, 0, 1.0, 1.0, 16, 256, 256, None, 0, None) == -1
    # map larger than the LDS weight tables
    big = (S * 1)(S(16, 2000, 160, 256, 256, 2000, 160, 0.25, 0, 0))
    assert lib.fo1_hfre_workspace_bytes(big, 1, 1) == 0
    # zero boxes is a no-op, not an error
    assert lib.fo1_hfre_region_pool(arr, 1, None, 0, None, 1.0, 1.0, 7, 0, 1.0, 1.0, None, 256, 256, None, 0, None) == 0
    # workspace too small -> -2
    assert lib.fo1_hfre_region_pool(arr, 1, 16, 3, None, 1.0, 1.0, 7, 0, 1.0, 1.0, 16, 256, 256, 16, 8, None) == -2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(L, "LIB_PATH_AB", str(tmp_path / "nope_ab.so"))
    with pytest.raises(L.Fo1Error):
        L.load()


def test_plain_c_host_builds_against_the_header_and_drives_the_library(tmp_path):
    """include/fo1.h is valid C99 (a non-Python integrator's compiler sees it), and a C program with no Python / torch in the process
    loads the library, sizes a workspace and gets an argument error back as (rc < 0, fo1_last_error() text) — tests/host_emul/abi_host.c."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    hdr = tmp_path / "hdr.c"
    hdr.write_text('#include "fo1.h"\n#include "fo1_ab.h"\nint main(void) { return 0; }\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(hdr)], check=True)
    exe = tmp_path / "abi_host"
    subprocess.run([gcc, "-std=c99", "-Wall", "-I", inc, os.path.join(root, "tests", "host_emul", "abi_host.c"), "-o", str(exe), "-ldl"], check=True)
    so = os.path.join(root, "vlm_fo1_amd", "libfo1hip.so")
    r = subprocess.run([str(exe), so], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rc=-1" in r.stdout and "NULL boxes" in r.stdout
