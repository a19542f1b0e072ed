"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, and refuses to
run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest


def _lib():
    from rovinasemanticsegmentation_amd import _capi as capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return capi


def test_library_exports_every_symbol_of_the_header():
    capi = _lib()
    L = capi.lib()
    hdr = open(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "include", "rvseg.h")).read()
    declared = set(re.findall(r"\b(rvseg_[a-z_]+)\s*\(", hdr))
    declared -= {"rvseg_ctx"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(capi.SYMBOLS)


def test_params_default_match_reference_config():
    capi = _lib()
    p = capi.default_params()
    assert (p.width, p.height, p.stride) == (640, 480, 2)           # config.json:87
    assert (p.patch_size, p.patch_size_reduce) == (77, 11)          # config.json:32,34
    assert (p.depth_min, p.depth_max) == (0.5, 15.0)                # config.json:89-90
    assert (p.dcrf_xyz_kernel, p.dcrf_rgb_kernel, p.dcrf_kernel_weight, p.dcrf_iterations) == (0.5, 4.0, 10.0, 10)
    assert p.use_dense_crf == 0                                      # config.json:81


def test_create_without_gpu_fails_loudly():
    capi = _lib()
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    p = capi.default_params()
    h = C.c_void_p()
    st = capi.lib().rvseg_create(C.byref(p), C.byref(h))
    assert st == capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in capi.lib().rvseg_last_error(None)
    assert not h.value


def test_create_rejects_bad_params_before_touching_the_device():
    capi = _lib()
    h = C.c_void_p()
    for kw in ({"stride": 0}, {"width": 2}, {"depth_min": 0.1}, {"max_batch": 0}, {"label_mode": 9}):
        p = capi.default_params(**kw)
        assert capi.lib().rvseg_create(C.byref(p), C.byref(h)) == capi.ERR_INVALID_ARG, kw


def test_cpp_facade_header_compiles():
    """include/rvseg_segmenter.hpp (the C++ mirror of `class Segmenter`) is plain C++17 over the C ABI."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "segmenter_facade_test.cpp")])
