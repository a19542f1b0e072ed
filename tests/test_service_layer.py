"""SURVEY.md 8(f) rank 3 -- the service / wire layer around the hot path, host only (no GPU):
the (map id -> labels) store with srvStoredSemanticsIds / srvGetLocalMapSegmentation
(src/segmenter.cpp:711-774), the debug cloud dumps (:684-706) and the DenseCRF2D feature builders
(third-party/densecrf/src/densecrf.cpp:61-81)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _capi():
    from rovinasemanticsegmentation_amd import _capi as capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return capi


def test_cpp_local_map_store_and_cloud_dumps(tmp_path):
    """tests/cpp/service_layer_test.cpp drives rvseg::LocalMapStore and rvseg::dump_clouds of the C++ facade."""
    _capi()
    exe = str(tmp_path / "svc")
    lib_dir = os.path.join(ROOT, "rovinasemanticsegmentation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "service_layer_test.cpp"), "-o", exe,
                           "-L", lib_dir, "-lrvseg", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "service layer ok" in r.stdout
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".cld")) == ["cloud42_layer_0.cld", "cloud42_layer_1.cld", "cloud42_rgb.cld"]


def test_python_local_map_store_follows_the_reference_services():
    _capi()
    from rovinasemanticsegmentation_amd.segmenter import LocalMapStore
    st = LocalMapStore(["material", "object"])
    assert st.srvStoredSemanticsIds() == []
    st.store(7, [[1, 2, 3], [4, 5, 6]])
    st.store(-3, [[0, 0], [8, 8]])
    assert st.srvStoredSemanticsIds() == [7, -3]                       # arrival order (:722-729)
    mid, labels = st.srvGetLocalMapSegmentation(7, ["object", "material"])
    assert mid == 7 and labels.dtype == np.uint8 and labels.tolist() == [4, 5, 6, 1, 2, 3]   # layers concatenated (:757-768)
    assert st.srvGetLocalMapSegmentation(7, ["material", "texture"]) is False     # unknown layer (:744-746)
    assert st.srvGetLocalMapSegmentation(8, ["material"]) is False                # unknown id (:773)


def test_densecrf2d_feature_builders_match_the_reference_expressions():
    """feature(0, j*W+i) = i / sx, feature(1, ..) = j / sy, colour channels im[..] / s (densecrf.cpp:61-81),
    all fp32 divisions; compared with the same expressions in numpy float32."""
    capi = _capi()
    W, H = 37, 23
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    g = capi.crf_features_gaussian(W, H, 3.0, 7.0)
    want = np.stack([xs.ravel().astype(np.float32) / np.float32(3.0), ys.ravel().astype(np.float32) / np.float32(7.0)], 1)
    assert g.dtype == np.float32 and np.array_equal(g, want)
    b = capi.crf_features_bilateral(W, H, 80.0, 60.0, 13.0, 11.0, 9.0, im)
    px = im.reshape(-1, 3).astype(np.float32)
    want = np.stack([xs.ravel().astype(np.float32) / np.float32(80.0), ys.ravel().astype(np.float32) / np.float32(60.0),
                     px[:, 0] / np.float32(13.0), px[:, 1] / np.float32(11.0), px[:, 2] / np.float32(9.0)], 1)
    assert np.array_equal(b, want)
