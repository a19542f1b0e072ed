"""GPU parity of the frame path (features -> forest -> up-sample -> pack -> labels) through the
C ABI against the CPU oracle: bit-exact features, log-posteriors and labels."""
import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic

pytestmark = pytest.mark.gpu


def _small_case(seed, W=96, H=64, lo=500, hi=15000):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    depth = (lo + (hi - lo) * ((xs + 2 * ys) % 97) / 96.0).astype(np.uint16)
    depth[rng.random((H, W)) < 0.08] = 0
    depth[5:9, 5:40] = 400          # below depth_min
    depth[20:24, 50:60] = 15001     # above depth_max
    return rgb, depth


def test_extract_features_full_frame_with_holes(gpu_ctx_factory, oracle):
    rgb, depth = synthetic.make_frame(3, holes=True)
    calib = synthetic.make_calib()
    want, wx, wy = oracle.extract(oracle.default_params(), rgb, depth, calib)
    ctx = gpu_ctx_factory()
    got, gx, gy = ctx.extract_features(rgb, depth, calib)
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy)
    assert got.shape == want.shape
    for name, sl in (("patch", slice(0, 363)), ("depth", slice(363, 364)), ("height", slice(364, 365)), ("normal", slice(365, 366))):
        assert np.array_equal(got[:, sl], want[:, sl]), name
    assert (want[:, 365] == -2).any() and (want[:, 365] > 0).any()


@pytest.mark.parametrize("density,seed", [(0.002, 1), (0.004, 2), (0.01, 3), (0.0005, 4)])
def test_smoothing_window_map_with_isolated_depth_edges(gpu_ctx_factory, oracle, density, seed):
    """The smoothing-window map comes from a two-pass raster chamfer distance over the whole image; the kernel re-runs
    both passes per 80 x 44 tile with a 10-pixel apron (kernels_features.hip: window_map_kernel), which is exact because
    a distance below 10 is a path of at most 9 steps.  Isolated invalid pixels on a smooth surface put most of the image
    at distances of 3 .. 15 from the nearest depth edge -- every window size, and sources just inside and just outside
    the apron, at every tile seam -- and the normal feature (window sums over that window) has to equal the oracle's
    bit for bit at stride 1, i.e. at every pixel."""
    W, H = 640, 480
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    depth = (2000 + 3 * xs + 2 * ys + 40 * np.sin(xs / 37.0) * np.cos(ys / 23.0)).astype(np.uint16)
    depth[rng.random((H, W)) < density] = 0
    kw = dict(width=W, height=H, stride=1, feature_color_patch=0, feature_depth=0, feature_height=0)
    calib = synthetic.make_calib(W, H)
    want, wx, wy = oracle.extract(oracle.default_params(**kw), rgb, depth, calib)
    ctx = gpu_ctx_factory(**kw)
    got, gx, gy = ctx.extract_features(rgb, depth, calib)
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy)
    assert got.shape == want.shape and want.shape[1] == 1
    assert np.array_equal(got, want)
    valid = want[:, 0] > -2
    assert 0.2 < valid.mean() < 0.999


@pytest.mark.parametrize("stride,patch,r", [(1, 7, 3), (2, 9, 4), (4, 5, 2)])
def test_extract_features_small_configs(gpu_ctx_factory, oracle, stride, patch, r):
    W, H = 96, 64
    rgb, depth = _small_case(stride)
    kw = dict(width=W, height=H, stride=stride, patch_size=patch, patch_size_reduce=r)
    calib = synthetic.make_calib(W, H)
    want, wx, wy = oracle.extract(oracle.default_params(**kw), rgb, depth, calib)
    ctx = gpu_ctx_factory(**kw)
    got, gx, gy = ctx.extract_features(rgb, depth, calib)
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy)
    assert np.array_equal(got, want)


def test_extract_feature_toggles(gpu_ctx_factory, oracle):
    W, H = 96, 64
    rgb, depth = _small_case(9)
    calib = synthetic.make_calib(W, H)
    for kw in (dict(feature_normal=0), dict(feature_color_patch=0), dict(feature_depth=0, feature_height=0)):
        kw = dict(width=W, height=H, patch_size=9, patch_size_reduce=3, **kw)
        want, wx, wy = oracle.extract(oracle.default_params(**kw), rgb, depth, calib)
        ctx = gpu_ctx_factory(**kw)
        got, gx, gy = ctx.extract_features(rgb, depth, calib)
        assert got.shape == want.shape and np.array_equal(got, want), kw


def test_near_plane_uses_full_reflected_border(gpu_ctx_factory, oracle):
    # depth 0.5 m -> half = 77 = the whole reflected border (feature_extractor.h:37,140)
    rgb, _ = synthetic.make_frame(0)
    depth = np.full((480, 640), 500, np.uint16)
    depth[:, 320:] = 15000       # half = 2 -> 5x5 ROI (up-sampling branch of the resize)
    calib = synthetic.make_calib()
    p = oracle.default_params(feature_normal=0)
    want, wx, wy = oracle.extract(p, rgb, depth, calib)
    ctx = gpu_ctx_factory(feature_normal=0)
    got, gx, gy = ctx.extract_features(rgb, depth, calib)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("multi,fill", [(1, 0.0), (0, -1000.0)])
def test_rf_frames_bit_exact(gpu_ctx_factory, oracle, multi, fill):
    blob = synthetic.make_forest_bytes(seed=11, n_trees=4, leaves_per_tree=2048, max_depth=24)
    rgb, depth = synthetic.make_batch(3, holes=True)
    calib = synthetic.make_calib()
    forest = oracle.Forest(blob)
    ctx = gpu_ctx_factory(multi_layer=multi, fill_value=fill, max_batch=2, label_mode=0 if multi == 0 else 2)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib)
    p = oracle.default_params(fill_value=fill)
    cc = forest.classes(multi)
    N = 640 * 480
    for i in range(3):
        want, P = oracle.rf_frame(p, forest, multi, rgb[i], depth[i], calib)
        assert 0 < P < 76800
        assert np.array_equal(out["posteriors"][i], want), i
        off = 0
        for l, C in enumerate(cc):
            wl = oracle.labels(want[off:off + N * C], C, 0 if multi == 0 else 2, unknown=[7, 8][l])
            assert np.array_equal(out["labels"][i, l].ravel(), wl)
            off += N * C
    # the eval-tool rule yields -1 only where nothing beats -1000
    if multi == 0:
        assert out["labels"].min() >= -1


def test_rf_frames_small_odd_sizes(gpu_ctx_factory, oracle):
    W, H = 100, 52   # not multiples of the distance-map tile; stride 2
    rgb, depth = _small_case(21, W, H)
    kw = dict(width=W, height=H, patch_size=9, patch_size_reduce=3)
    blob = synthetic.make_forest_bytes(seed=4, n_trees=3, leaves_per_tree=128, max_depth=10, D=30)
    forest = oracle.Forest(blob)
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(**kw)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb[None], depth[None], calib)
    want, _ = oracle.rf_frame(oracle.default_params(**kw), forest, 1, rgb, depth, calib)
    assert np.array_equal(out["posteriors"][0], want)


def test_stride_must_divide_image(gpu_ctx_factory):
    import rovinasemanticsegmentation_amd as rv
    ctx = gpu_ctx_factory(width=101, height=52, patch_size=9, patch_size_reduce=3, feature_color_patch=0)
    with pytest.raises(rv.capi.RvsegError):
        ctx.extract_features(np.zeros((52, 101, 3), np.uint8), np.zeros((52, 101), np.uint16), synthetic.make_calib(101, 52))


def test_frame_without_any_valid_depth(gpu_ctx_factory, oracle):
    """No sample point passes the mask: the low-res image keeps the fill value everywhere and the
    labels follow the rule for it (eval rule: -1 with fill -1000; node rule: unknown with fill 0)."""
    blob = synthetic.make_forest_bytes(seed=4, n_trees=2, leaves_per_tree=32, max_depth=6)
    rgb, _ = synthetic.make_batch(1)
    depth = np.zeros((1, 480, 640), np.uint16)
    calib = synthetic.make_calib()
    ctx = gpu_ctx_factory(multi_layer=0, fill_value=-1000.0, label_mode=0)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib)
    assert (out["posteriors"] == -1000.0).all() and (out["labels"] == -1).all()
    feat, xv, yv = ctx.extract_features(rgb[0], depth[0], calib)
    assert feat.shape[0] == 0
    ctx2 = gpu_ctx_factory(multi_layer=1, fill_value=0.0, label_mode=2)
    ctx2.forest_load(blob)
    out = ctx2.segment_frames(rgb, depth, calib)
    assert (out["posteriors"] == 0).all()
    assert (out["labels"][0, 0] == 7).all() and (out["labels"][0, 1] == 8).all()
    # zero frames is a no-op
    ctx2.L.rvseg_segment_frames(ctx2.h, 0, None, None, None, None, None, None)


@pytest.mark.parametrize("stride,toggles", [(1, dict()), (4, dict(feature_color_patch=0)), (2, dict(feature_normal=0, feature_height=0))])
def test_rf_frames_strides_and_feature_toggles(gpu_ctx_factory, oracle, stride, toggles):
    """The on-demand feature walk with other sample grids and without the colour patch / scalar features."""
    W, H = 96, 64
    rgb, depth = _small_case(5, W, H)
    kw = dict(width=W, height=H, stride=stride, patch_size=9, patch_size_reduce=3, **toggles)
    D = oracle.feature_length(oracle.default_params(**kw))
    blob = synthetic.make_forest_bytes(seed=9, n_trees=5, leaves_per_tree=64, max_depth=9, D=D)
    forest = oracle.Forest(blob)
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(**kw)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb[None], depth[None], calib)
    want, _ = oracle.rf_frame(oracle.default_params(**kw), forest, 1, rgb, depth, calib)
    assert np.array_equal(out["posteriors"][0], want)


@pytest.mark.parametrize("stride,W,H", [(1, 128, 64), (2, 192, 68), (4, 256, 128), (2, 64, 32)])
def test_upsampling_through_lds_tiles_other_scales(gpu_ctx_factory, oracle, stride, W, H):
    """Images whose width is a multiple of 64 and height a multiple of 4 are up-sampled by the tiled kernel
    (kernels_rf.hip: upsample_pack_tiled_kernel: 64 x 4 output pixels per block through LDS); strides 1, 2 and 4 give it
    source windows of 66, 34 and 18 columns, with the clamped first / last rows and the `tail` column of cv::resize at
    the borders.  Posteriors of both label layers (8 and 9 classes) against the oracle, bit for bit."""
    rgb, depth = _small_case(stride + W, W, H)
    kw = dict(width=W, height=H, stride=stride, patch_size=9, patch_size_reduce=3)
    D = oracle.feature_length(oracle.default_params(**kw))
    blob = synthetic.make_forest_bytes(seed=13, n_trees=4, leaves_per_tree=64, max_depth=9, D=D)
    forest = oracle.Forest(blob)
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(**kw)
    ctx.forest_load(blob)
    out = ctx.segment_frames(np.stack([rgb, rgb[::-1].copy()]), np.stack([depth, depth[::-1].copy()]), calib)
    for i, (r, d) in enumerate(((rgb, depth), (rgb[::-1].copy(), depth[::-1].copy()))):
        want, _ = oracle.rf_frame(oracle.default_params(**kw), forest, 1, r, d, calib)
        assert np.array_equal(out["posteriors"][i], want), i


def test_lazy_walk_touches_every_feature(gpu_ctx_factory, oracle):
    """The production kernel (rf_frames_lazy_kernel) computes a patch value only when a node tests it, so
    a small forest exercises a small part of the 363 cells.  64 trees x 512 leaves = 32 704 inner nodes:
    every one of the 366 features is split on (asserted below), ~90 times each with thresholds spread
    over the value range, and the posteriors of every sample point must still equal the oracle's."""
    import struct
    W, H = 160, 120
    blob = synthetic.make_forest_bytes(seed=77, n_trees=64, leaves_per_tree=512, max_depth=14)
    # features that inner nodes split on (stream layout: int32 T, then per tree vec<int32> features, ...)
    pos, used = 4, set()
    for _ in range(64):
        n = struct.unpack_from("<i", blob, pos)[0]
        feat = np.frombuffer(blob, np.int32, n, pos + 4)
        left = np.frombuffer(blob, np.int32, n, pos + 4 + 4 * n + 4 + 4 * n + 4)
        used |= set(feat[left != 0].tolist())
        pos += 3 * (4 + 4 * n)
        for _vec in range(2):   # histograms, multi_histograms: skip by walking the length prefixes
            cnt = struct.unpack_from("<i", blob, pos)[0]
            pos += 4
            for _i in range(cnt):
                m = struct.unpack_from("<i", blob, pos)[0]
                pos += 4
                if _vec == 0:
                    pos += 4 * m
                else:
                    for _l in range(m):
                        c = struct.unpack_from("<i", blob, pos)[0]
                        pos += 4 + 4 * c
    assert used == set(range(366))
    forest = oracle.Forest(blob)
    rgb, depth = synthetic.make_batch(2, W, H, holes=True, start=3)
    depth[1, :, : W // 2] = 520       # near plane: the widest ROIs (half = 74), reflected border in play
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(width=W, height=H, max_batch=2)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib, want_labels=False)
    p = oracle.default_params(width=W, height=H)
    for i in range(2):
        want, P = oracle.rf_frame(p, forest, 1, rgb[i], depth[i], calib)
        assert P > 1000
        assert np.array_equal(out["posteriors"][i], want), i


def test_host_entry_with_page_locked_buffers_equals_the_staged_path(gpu_ctx_factory):
    """rvseg_host_register'ed in / out buffers (the copy engines write the caller's memory directly) against ordinary
    pageable numpy buffers (pinned staging ring inside the library): identical outputs, three chunks per call."""
    W, H = 160, 120
    blob = synthetic.make_forest_bytes(seed=12, n_trees=4, leaves_per_tree=256, max_depth=12)
    rgb, depth = synthetic.make_batch(20, W, H, holes=True)
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(width=W, height=H, use_dense_crf=1, dcrf_iterations=2, label_mode=1, max_batch=8)
    ctx.forest_load(blob)
    want = ctx.segment_frames(rgb, depth, calib)
    bufs = ctx.host_buffers(20)
    try:
        bufs["rgb"][...] = rgb
        bufs["depth"][...] = depth
        for k in ("posteriors", "marginals", "labels"):
            bufs[k].fill(0)
        got = ctx.segment_frames(bufs["rgb"], bufs["depth"], calib, out=bufs)
        for k in ("posteriors", "marginals", "labels"):
            assert np.array_equal(got[k], want[k]), k
        # mixed: page-locked outputs, pageable inputs
        for k in ("posteriors", "marginals", "labels"):
            bufs[k].fill(0)
        got = ctx.segment_frames(rgb, depth, calib, out=bufs)
        for k in ("posteriors", "marginals", "labels"):
            assert np.array_equal(got[k], want[k]), k
    finally:
        ctx.release_host_buffers(bufs)
