"""CPU-side checks of the product: the C-ABI library builds for gfx950, loads, exports every symbol the header
declares, and refuses to work without a GPU (no CPU fallback).  No compute here."""
import ctypes
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from pecaller_amd import build
    return build.build()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "pemap_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pe(?:map|call)_dev_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib_path):
    from pecaller_amd import pemap
    syms = header_symbols()
    assert len(syms) >= 30
    assert sorted(pemap.SYMBOLS) == syms, "pecaller_amd.pemap.SYMBOLS is out of sync with include/pemap_hip.h"
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and nothing torch-shaped in the signatures: the header compiles as plain C
    src = os.path.join(ROOT, "tests", "_hdr_check.c")
    with open(src, "w") as f:
        f.write('#include "../include/pemap_hip.h"\nint main(void){return PEMAP_MAX_HITS == 200 ? 0 : 1;}\n')
    try:
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", src])
    finally:
        os.remove(src)


def test_device_code_is_gfx950_only(lib_path):
    """the fat binary carries exactly one device code object, for gfx950"""
    blob = open(lib_path, "rb").read()
    ids = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert ids == {b"gfx950"}, ids


def test_no_cpu_fallback(lib_path):
    """without a GPU pemap_dev_create must fail loudly; with one this test is skipped"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pecaller_amd import PemapDev, PemapError
    with pytest.raises(PemapError) as e:
        PemapDev(0)
    assert "no HIP device" in str(e.value) or "CPU" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """nothing under pecaller_amd/ may import, link or call oracle/ (it is the checker, not the product)"""
    bad = []
    for dp, dn, fn in os.walk(os.path.join(ROOT, "pecaller_amd")):
        for f in fn:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if "oracle" in t.lower() and f not in ("__init__.py",):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_program_builds():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "pecaller_amd", "csrc")])
    exe = os.path.join(ROOT, "pecaller_amd", "pemapper_hip")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "o", "x.sdx", "q", "f"], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage" in r.stdout      # pemapper.c:226-232
    exe = os.path.join(ROOT, "pecaller_amd", "pecaller_hip")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "pileup", "x.sdx"], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage" in r.stdout      # pecaller.c:251-257


def test_shard_ranges_cover_without_overlap():
    from pecaller_amd import dist as pd
    for n in (0, 1, 7, 1000, 10**7 + 3):
        for w in (1, 2, 3, 8):
            r = [pd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
