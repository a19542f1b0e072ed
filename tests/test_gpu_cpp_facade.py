"""The C++ Segmenter facade (include/rvseg_segmenter.hpp) compiled with g++ against librvseg.so and
checked against the oracle: per-frame posteriors bit-exact, cloud labels with / without CRF."""
import os
import subprocess

import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_segmenter_facade(tmp_path, oracle, golden_dir):
    exe = str(tmp_path / "facade")
    lib_dir = os.path.join(ROOT, "rovinasemanticsegmentation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "segmenter_facade_test.cpp"), "-o", exe,
                           "-L", lib_dir, "-lrvseg", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    W, H = 160, 120
    rgb, depth = synthetic.make_batch(1, W, H, holes=True)
    (tmp_path / "rgb.u8").write_bytes(rgb[0].tobytes())
    (tmp_path / "depth.u16").write_bytes(depth[0].tobytes())
    forest_path = os.path.join(golden_dir, "forest_multi.dat")
    out_path = str(tmp_path / "out.bin")
    r = subprocess.run([exe, forest_path, str(tmp_path / "rgb.u8"), str(tmp_path / "depth.u16"), out_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    N = W * H
    raw = np.fromfile(out_path, np.uint8)
    post = raw[: 17 * N * 4].view(np.float32)
    pairwise = raw[17 * N * 4: 17 * N * 4 + 6 * N * 4].view(np.float32).reshape(N, 6)
    o1 = 17 * N * 4 + 6 * N * 4
    labels = raw[o1:o1 + 4 * N].reshape(2, 2, N)   # [layer][crf, plain][N]
    fused = raw[o1 + 4 * N:o1 + 4 * N + 17 * N * 4].view(np.float32)
    fused_labels = raw[o1 + 4 * N + 17 * N * 4:].reshape(2, N)

    forest = oracle.Forest(forest_path)
    p = oracle.default_params(width=W, height=H)
    want_post, _ = oracle.rf_frame(p, forest, 1, rgb[0], depth[0], synthetic.make_calib(W, H))
    assert np.array_equal(post, want_post)
    off = 0
    for l, C in enumerate((8, 9)):
        un = want_post[off:off + N * C].reshape(N, C)
        Q = oracle.crf_inference(-un, pairwise, 10.0, 3)
        assert np.array_equal(labels[l, 0].view(np.int8), oracle.labels(Q, C, 1, unknown=C - 1))
        assert np.array_equal(labels[l, 1].view(np.int8), oracle.labels(un, C, 2, unknown=C - 1))
        off += N * C
    # fusion of the frame seen twice (segmenter.cpp:561-616) and the no-CRF cloud labels (:660-681)
    idx = np.empty((2, N), np.int32)
    idx[0] = np.arange(N)
    idx[1] = np.where(np.arange(N) % 3 == 0, N - 1 - np.arange(N), -1)
    want = oracle.fuse_posteriors(idx.reshape(2, H, W), np.stack([want_post, want_post]), [8, 9], N)
    assert np.array_equal(fused, want)
    off = 0
    for l, C in enumerate((8, 9)):
        assert np.array_equal(fused_labels[l].view(np.int8), oracle.labels(want[off:off + N * C].reshape(N, C), C, 2, unknown=C - 1))
        off += N * C


def test_cpp_segmenter_queue_layer(tmp_path, golden_dir):
    """enqueueFrame / processFramesFromQueueInternalRF / onNewLocalMap / processMapFromQueue / start-stop / commInit +
    gatherLabels (include/rvseg_segmenter.hpp; src/segmenter.cpp:245-304, 334-346, 434, 518-621): two cameras, 20 key
    frames enqueued out of order, batched draining compared bit for bit with per-frame calls, skipped and postponed
    maps, the worker threads, the RCCL gather at world size 1 (tests/cpp/segmenter_queue_test.cpp)."""
    exe = str(tmp_path / "queue")
    lib_dir = os.path.join(ROOT, "rovinasemanticsegmentation_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "segmenter_queue_test.cpp"), "-o", exe,
                           "-L", lib_dir, "-lrvseg", "-lpthread", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    W, H = 160, 120
    rgb, depth = synthetic.make_batch(1, W, H, holes=True)
    (tmp_path / "rgb.u8").write_bytes(rgb[0].tobytes())
    (tmp_path / "depth.u16").write_bytes(depth[0].tobytes())
    r = subprocess.run([exe, os.path.join(golden_dir, "forest_multi.dat"), str(tmp_path / "rgb.u8"), str(tmp_path / "depth.u16")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "queue ok" in r.stdout
