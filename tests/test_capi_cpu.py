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


def test_schedule_struct_mirrors_the_header_field_by_field():
    """rvseg_schedule / rvseg_schedule_info in _capi.py against include/rvseg.h (names, order, count), and the defaults the
    header documents -- rvseg_schedule_default is host-only, so this runs without a GPU."""
    import ctypes as C
    import re
    from rovinasemanticsegmentation_amd import _capi as capi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rvseg.h")).read()

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in re.findall(r"int32_t\s+([^;]+);", body):
            names += [n.strip() for n in decl.split(",")]
        return names
    assert [f[0] for f in capi.RvsegSchedule._fields_] == fields("rvseg_schedule")
    assert [f[0] for f in capi.RvsegScheduleInfo._fields_] == fields("rvseg_schedule_info")
    sc = capi.RvsegSchedule()
    for f, _ in capi.RvsegSchedule._fields_:
        setattr(sc, f, 77)
    capi.lib().rvseg_schedule_default(C.byref(sc))
    got = {f: getattr(sc, f) for f, _ in capi.RvsegSchedule._fields_}
    want = dict.fromkeys(got, 0)
    want.update(resident_band=16, resident_chunk=128, resident_window=-1, overlap_build=1, overlap_layers=1)
    assert got == want


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


# ---- host-only forest entry points (no GPU needed): rvseg_forest_check / rvseg_forest_rewrite -------
def _golden(name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", name), "rb") as fh:
        return fh.read()


@pytest.mark.parametrize("name", ["forest_multi.dat", "forest_single.dat", "forest_tiny.dat"])
def test_forest_writer_round_trips_reference_written_files_byte_for_byte(name):
    """The golden files were written by the reference's own RandomForest::write (oracle/gen_golden.py):
    parsing them and writing them again must give the same bytes (classifier.cpp:144-152,210-220)."""
    capi = _lib()
    blob = _golden(name)
    assert capi.forest_rewrite(blob) == blob


def test_forest_check_accepts_the_goldens_and_reports_their_shape():
    capi = _lib()
    st, msg, info = capi.forest_check(_golden("forest_multi.dat"), 366)
    assert st == capi.OK, msg
    assert info["n_trees"] >= 1 and info["n_nodes"] > info["n_trees"] and info["max_depth"] >= 1


def test_forest_check_refuses_more_than_64_trees():
    """libforest has no tree limit; the device evaluator keeps 16 leaf rows in each of a point's 4
    lanes.  A 65-tree ensemble must be refused, not mis-evaluated."""
    capi = _lib()
    from rovinasemanticsegmentation_amd import synthetic
    ok = synthetic.make_forest_bytes(seed=3, n_trees=64, leaves_per_tree=4, max_depth=4)
    st, msg, info = capi.forest_check(ok, 366)
    assert st == capi.OK and info["n_trees"] == 64, msg
    big = synthetic.make_forest_bytes(seed=3, n_trees=65, leaves_per_tree=4, max_depth=4)
    st, msg, _ = capi.forest_check(big, 366)
    assert st == capi.ERR_CAPACITY and "65 trees" in msg


def test_forest_check_refuses_corrupt_streams_and_model_config_mismatch():
    capi = _lib()
    blob = _golden("forest_tiny.dat")
    assert capi.forest_check(blob[:len(blob) // 2])[0] == capi.ERR_FORMAT       # truncated
    assert capi.forest_check(b"\x00\x00\x00\x00")[0] == capi.ERR_NO_FOREST       # T = 0
    st, msg, _ = capi.forest_check(_golden("forest_multi.dat"), 3)               # D = 3: split features out of range
    assert st == capi.ERR_FORMAT and "mismatch" in msg
