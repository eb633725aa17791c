"""The C-ABI library loads and exports every symbol include/orbfe.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from orb_slam2_ssd_semantic_amd import _ffi
    L = _ffi.lib()
    names = header_symbols()
    assert len(names) >= 25
    assert set(names) == set(_ffi.SYMBOLS), set(names) ^ set(_ffi.SYMBOLS)
    for n in names:
        assert getattr(L, n) is not None
    assert L.orbfe_version() == 100
    assert L.orbfe_strerror(0) == b"ok"
    assert b"no usable HIP device" in L.orbfe_strerror(-6)


def test_struct_layouts():
    from orb_slam2_ssd_semantic_amd import _ffi
    assert _ffi.KP_DTYPE.itemsize == 28  # cv::KeyPoint
    assert C.sizeof(_ffi.OrbfeParams) == 40
    assert [n for n in _ffi.KP_DTYPE.names] == ["x", "y", "size", "angle", "response", "octave", "class_id"]


def test_host_hamming_without_device(oracle):
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    rng = np.random.default_rng(0)
    for _ in range(100):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b)


def test_no_cpu_fallback(have_gpu):
    """Without a GPU the product must fail loudly (ORBFE_ERR_NODEVICE), never compute on the CPU."""
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher, OrbfeError, _ffi
    if have_gpu:
        e = ORBextractor()
        assert e.capacity() >= 1000 + 16
        return
    for ctor in (ORBextractor, ORBmatcher):
        try:
            ctor()
        except OrbfeError as err:
            assert err.status == _ffi.ORBFE_ERR_NODEVICE
        else:
            raise AssertionError("constructed without a HIP device")


def test_bad_params_rejected():
    from orb_slam2_ssd_semantic_amd import _ffi
    L = _ffi.lib()
    h = C.c_void_p()
    bad = _ffi.OrbfeParams(1000, 1.2, 0, 20, 7, 640, 480, 1, -1, 0)  # nlevels = 0
    assert L.orbfe_create(C.byref(bad), C.byref(h)) == _ffi.ORBFE_ERR_ARG
    assert L.orbfe_create(None, C.byref(h)) == _ffi.ORBFE_ERR_ARG
    assert L.orbfe_get_scales(None, None, None, None, None) == _ffi.ORBFE_ERR_ARG
    assert L.orbfe_extract(None, None, 0, 0, 0, None, None, 0, None) == _ffi.ORBFE_ERR_ARG


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file of the product package may reference it."""
    pkg = os.path.join(ROOT, "orb_slam2_ssd_semantic_amd")
    for dp, _, files in os.walk(pkg):
        if os.path.basename(dp) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc", ".inc")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_ffi" not in txt and "orb_oracle" not in txt and "liborb_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
