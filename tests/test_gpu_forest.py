"""GPU parity of the forest path: C-ABI rvseg_forest_eval vs the reference goldens and the oracle."""
import os

import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vec(golden_dir):
    return np.load(os.path.join(golden_dir, "forest_vectors.npz"))


def test_multi_layer_bit_exact_vs_reference_golden(gpu_ctx_factory, golden_dir, vec):
    ctx = gpu_ctx_factory(multi_layer=1)
    ctx.forest_load(os.path.join(golden_dir, "forest_multi.dat"))
    info = ctx.forest_info()
    assert info["n_trees"] == 4 and info["class_counts"] == [8, 9]
    out = ctx.forest_eval(vec["points"])
    assert np.array_equal(out.ravel(), vec["forest_multi_multi"])


def test_single_layer_bit_exact_vs_reference_golden(gpu_ctx_factory, golden_dir, vec):
    ctx = gpu_ctx_factory(multi_layer=0)
    ctx.forest_load(os.path.join(golden_dir, "forest_single.dat"))
    assert ctx.forest_info()["class_counts"] == [9]
    assert np.array_equal(ctx.forest_eval(vec["points"]).ravel(), vec["forest_single_single"])
    ctx.forest_load(os.path.join(golden_dir, "forest_multi.dat"))
    assert np.array_equal(ctx.forest_eval(vec["points"]).ravel(), vec["forest_multi_single"])


def test_big_forest_vs_oracle(gpu_ctx_factory, oracle):
    from rovinasemanticsegmentation_amd import synthetic
    blob = synthetic.make_forest_bytes(seed=5, n_trees=4, leaves_per_tree=4096, max_depth=30)
    X = synthetic.random_points(9, 20000)
    want = oracle.Forest(blob).eval(X, multi=True)
    ctx = gpu_ctx_factory(multi_layer=1)
    ctx.forest_load(blob)
    assert ctx.forest_info()["max_depth"] <= 30
    got = ctx.forest_eval(X)
    assert np.array_equal(got, want)


def test_many_trees_and_empty_input(gpu_ctx_factory, oracle):
    from rovinasemanticsegmentation_amd import synthetic
    blob = synthetic.make_forest_bytes(seed=6, n_trees=11, leaves_per_tree=64, max_depth=10)
    X = synthetic.random_points(2, 777)
    ctx = gpu_ctx_factory(multi_layer=1)
    ctx.forest_load(blob)
    assert np.array_equal(ctx.forest_eval(X), oracle.Forest(blob).eval(X, multi=True))
    assert ctx.forest_eval(np.zeros((0, 366), np.float32)).shape == (0, 17)


def test_loader_errors(gpu_ctx_factory, golden_dir):
    import rovinasemanticsegmentation_amd as rv
    ctx = gpu_ctx_factory(multi_layer=1)
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.forest_load("/nonexistent/forest.dat")
    assert e.value.status == rv.capi.ERR_IO and "Could not open file" in str(e.value)
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.forest_eval(np.zeros((1, 366), np.float32))
    assert e.value.status == rv.capi.ERR_NO_FOREST
    data = open(os.path.join(golden_dir, "forest_multi.dat"), "rb").read()
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.forest_load(data[: len(data) // 2])
    assert e.value.status == rv.capi.ERR_FORMAT
    # model/config mismatch: forest_tiny splits on features < 4 but carries (3,2) layers; a
    # single-layer forest in multi mode must be refused
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.forest_load(os.path.join(golden_dir, "forest_single.dat"))
    assert e.value.status == rv.capi.ERR_FORMAT
    # feature index out of range for a smaller feature vector
    small = gpu_ctx_factory(multi_layer=1, feature_color_patch=0)
    with pytest.raises(rv.capi.RvsegError) as e:
        small.forest_load(data)
    assert e.value.status == rv.capi.ERR_FORMAT and "mismatch" in str(e.value)


def test_forest_eval_equals_the_reference_evaluator_run_on_this_box(gpu_ctx_factory, tmp_path):
    """oracle/_ref/libforest_ref is the REFERENCE's own classifier.cpp compiled in the build container (make -C oracle
    ref; nothing of the reference is in the repository, the binary travels with the snapshot).  It is run here, on the GPU
    box, on a bench-sized forest (4 trees x 2^14 leaves, depth <= 30, D = 366) and on a forest written by the GPU trainer:
    RandomForest::read parses both files, multiClassLogPosterior / classLogPosterior evaluate 20 000 points, and the HIP
    evaluator must return the same floats bit for bit -- reference-pinned parity at a size the committed vectors
    (1 024 points, 96 leaves per tree) do not reach."""
    import os
    import subprocess
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libforest_ref")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/libforest_ref not built (needs /root/reference in the build container)")
    X = synthetic.random_points(77, 20000)
    (tmp_path / "x.f32").write_bytes(X.tobytes())
    big = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=9, layer_classes=(8, 9))
    # a model written by the GPU trainer (two layers; shared forests carry no single-label histograms)
    tctx = gpu_ctx_factory()
    lab = np.stack([(X[:, 3] // 32).astype(np.int32), (X[:, 363] > 7).astype(np.int32) + 2 * (X[:, 100] > 128).astype(np.int32)], 1)
    trained = tctx.forest_train(X, lab, [8, 4], num_trees=3, max_depth=12, min_split_examples=20, seed=3)
    for name, blob, modes in (("bench", big, ("single", "multi")), ("trained", trained, ("multi",))):
        path = tmp_path / (name + ".dat")
        path.write_bytes(blob)
        for mode in modes:
            out = tmp_path / (name + "_" + mode + ".f32")
            subprocess.check_call([ref, "eval", str(path), str(tmp_path / "x.f32"), "366", mode, str(out)], timeout=300)
            want = np.fromfile(out, np.float32)
            ctx = gpu_ctx_factory(multi_layer=1 if mode == "multi" else 0)
            ctx.forest_load(blob)
            got = ctx.forest_eval(X)
            assert got.size == want.size, (name, mode)
            assert np.array_equal(got.ravel(), want), (name, mode)
