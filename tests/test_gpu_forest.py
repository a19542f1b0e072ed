"""GPU parity of the forest path: C-ABI rvseg_forest_eval vs the reference goldens and the oracle."""
import os

import numpy as np
import pytest

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
