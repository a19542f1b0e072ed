"""Known-answer and property tests of the oracle's feature path (rows A-F, K-M).

OpenCV / PCL are not vendored and not installed: these rows are PARITY UNPINNED at the
third-party level; the tests pin the reference-owned rules by hand-computed values and the
third-party restatements by independent float formulas.
"""
import numpy as np
import pytest


# ------------------------------------------------------------------------------------------- Lab
def _lab_float(img):
    """Float CIE Lab of an image whose channel 0 is treated as B (CV_BGR2Lab), scaled to 8 bit."""
    x = img.astype(np.float64) / 255.0
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    B, G, R = lin[..., 0], lin[..., 1], lin[..., 2]
    X = (0.412453 * R + 0.357580 * G + 0.180423 * B) / 0.950456
    Y = 0.212671 * R + 0.715160 * G + 0.072169 * B
    Z = (0.019334 * R + 0.119193 * G + 0.950227 * B) / 1.088754
    f = lambda t: np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
    L = np.where(Y > 0.008856, 116 * np.cbrt(Y) - 16, 903.3 * Y)
    a = 500 * (f(X) - f(Y))
    b = 200 * (f(Y) - f(Z))
    return np.stack([L * 255 / 100, a + 128, b + 128], -1)


def test_lab_known_colours(oracle):
    img = np.array([[[0, 0, 0], [255, 255, 255], [128, 128, 128]]], np.uint8)
    lab = oracle.bgr2lab(img)
    assert lab[0, 0].tolist() == [0, 128, 128]
    assert lab[0, 1].tolist() == [255, 128, 128]
    assert lab[0, 2, 1] == 128 and lab[0, 2, 2] == 128 and abs(int(lab[0, 2, 0]) - 137) <= 1


def test_lab_close_to_float_formula_with_rb_swap(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    lab = oracle.bgr2lab(img).astype(np.float64)
    ref = _lab_float(img)
    assert np.abs(lab - ref).max() <= 3.0  # 11-bit LUT index quantisation in the dark range
    assert np.abs(lab - ref).mean() < 0.5
    # channel 0 really is treated as blue: swapping channels changes the answer
    swapped = oracle.bgr2lab(img[..., ::-1].copy()).astype(np.float64)
    assert np.abs(swapped - ref).mean() > 5


# ------------------------------------------------------------------------------------- 8-bit resize
def test_patch_resize_identity_and_constant(oracle):
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    # ROI side == output side -> plain copy
    out = oracle.resize_patch(lab, 7, 9, 11, 11)
    assert np.array_equal(out, lab[9:20, 7:18])
    const = np.full((30, 30, 3), 77, np.uint8)
    for size in (5, 11, 23, 29):
        assert (oracle.resize_patch(const, 0, 0, size, 11) == 77).all()


def test_patch_resize_matches_float_bilinear(oracle):
    rng = np.random.default_rng(2)
    # smooth image so that +-1 quantisation is the only difference
    ys, xs = np.mgrid[0:80, 0:90]
    lab = np.stack([xs * 2 + ys, 255 - xs - ys, (xs * ys) % 200], -1).clip(0, 255).astype(np.uint8)
    for size in (5, 7, 21, 45, 77):
        x0, y0 = 3, 2
        out = oracle.resize_patch(lab, x0, y0, size, 11).astype(np.float64)
        scale = size / 11.0
        want = np.zeros((11, 11, 3))
        for dy in range(11):
            fy = (dy + 0.5) * scale - 0.5
            sy = int(np.floor(fy)); fy -= sy
            r0, r1 = min(max(sy, 0), size - 1), min(max(sy + 1, 0), size - 1)
            for dx in range(11):
                fx = (dx + 0.5) * scale - 0.5
                sx = int(np.floor(fx)); fx -= sx
                if sx < 0: sx, fx = 0, 0.0
                if sx >= size - 1: sx, fx = size - 1, 0.0
                c1 = min(sx + 1, size - 1)
                roi = lab[y0:y0 + size, x0:x0 + size].astype(np.float64)
                top = roi[r0, sx] * (1 - fx) + roi[r0, c1] * fx
                bot = roi[r1, sx] * (1 - fx) + roi[r1, c1] * fx
                want[dy, dx] = top * (1 - fy) + bot * fy
        assert np.abs(out - want).max() <= 1.0, size


def test_patch_resize_reflect_border(oracle):
    # BORDER_REFLECT = fedcba|abcdefgh|hgfedcb: index -1 -> 0, -2 -> 1, W -> W-1
    lab = np.zeros((12, 12, 3), np.uint8)
    lab[..., 0] = np.arange(12)[None, :] * 10
    lab[..., 1] = np.arange(12)[:, None] * 10
    out = oracle.resize_patch(lab, -3, -2, 11, 11)  # identity size: pure gather
    cols = [2, 1, 0] + list(range(8))
    rows = [1, 0] + list(range(9))
    assert out[0, :, 0].tolist() == [c * 10 for c in cols]
    assert out[:, 0, 1].tolist() == [r * 10 for r in rows]
    out = oracle.resize_patch(lab, 5, 5, 11, 11)
    assert out[0, :, 0].tolist() == [50, 60, 70, 80, 90, 100, 110, 110, 100, 90, 80]


# ------------------------------------------------------------------------------------ float resize
def test_upsample_2x_weights_and_edges(oracle):
    rng = np.random.default_rng(3)
    src = rng.standard_normal((6, 5, 3)).astype(np.float32)
    out = oracle.resize_linear(src, 10, 12)
    f32 = np.float32
    # first / last column copy the edge sample in x; interior x weights are .75/.25
    def hrow(r):
        h = np.empty((10, 3), f32)
        for dx in range(10):
            if dx == 0:
                h[dx] = src[r, 0] * f32(1) + src[r, 1] * f32(0)
            elif dx == 9:
                h[dx] = src[r, 4] * f32(1)
            elif dx % 2 == 1:
                k = dx // 2
                h[dx] = src[r, k] * f32(0.75) + src[r, k + 1] * f32(0.25)
            else:
                k = dx // 2 - 1
                h[dx] = src[r, k] * f32(0.25) + src[r, k + 1] * f32(0.75)
        return h
    # y keeps its weights and clips rows: dy=0 -> rows (0,0) with (.25,.75)
    want0 = hrow(0) * f32(0.25) + hrow(0) * f32(0.75)
    assert np.array_equal(out[0], want0)
    want1 = hrow(0) * f32(0.75) + hrow(1) * f32(0.25)
    assert np.array_equal(out[1], want1)
    want_last = hrow(5) * f32(0.75) + hrow(5) * f32(0.25)
    assert np.array_equal(out[11], want_last)


def test_upsample_bleeds_fill_value(oracle):
    src = np.full((4, 4, 2), -1000, np.float32)
    src[1, 1] = (-1.0, -2.0)
    out = oracle.resize_linear(src, 8, 8)
    # the valid cell is blended with the sentinel in its neighbourhood
    assert out[2, 2, 0] == np.float32(np.float32(np.float32(-1000 * 0.25) + np.float32(-1.0 * 0.75)) * np.float32(0.75)
                                       + np.float32(-1000 * 0.25 + -1000 * 0.75) * np.float32(0.25)) or out[2, 2, 0] < -1
    assert (out[5:, 5:] == -1000).all()


# ------------------------------------------------------------------------------------------- cloud
def test_cloud_known_answer(oracle):
    p = oracle.default_params(width=8, height=6, depth_min=0.5, depth_max=15.0)
    depth = np.full((6, 8), 2000, np.uint16)
    depth[0, 0] = 0        # invalid -> NaN
    depth[0, 1] = 499      # 0.499 < 0.5 -> NaN
    depth[0, 2] = 500      # valid
    depth[0, 3] = 15001    # > 15 -> NaN
    calib = np.concatenate([np.eye(3).ravel(), np.eye(3).ravel(), [1, 2, 3]]).astype(np.float32)
    cl = oracle.cloud(p, depth, calib)
    assert np.isnan(cl[0, 0]).all() and np.isnan(cl[0, 1]).all() and np.isnan(cl[0, 3]).all()
    assert cl[0, 2].tolist() == [0.5 * 2 + 1, 0.0 + 2, 0.5 + 3]
    assert cl[4, 5].tolist() == [2.0 * 5 + 1, 2.0 * 4 + 2, 2.0 + 3]
    # rotation is applied as (R*Kinv)*m + t
    R = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float32)
    calib = np.concatenate([np.eye(3).ravel(), R.ravel(), [0, 0, 0.5]]).astype(np.float32)
    cl = oracle.cloud(p, depth, calib)
    assert cl[4, 5].tolist() == [2.0, -10.0, -8.0 + 0.5]


# ------------------------------------------------------------------------------------------ normals
def _plane_cloud(W, H, nx=0.0, ny=0.0):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    z = 2.0 + nx * xs * 0.01 + ny * ys * 0.01
    return np.stack([xs * 0.01, ys * 0.01, z], -1).astype(np.float32)


def test_normals_plane_and_border(oracle):
    cl = _plane_cloud(48, 40)
    nz, dist = oracle.normals_nz(cl)
    assert np.isnan(nz[:10]).all() and np.isnan(nz[-10:]).all() and np.isnan(nz[:, :10]).all() and np.isnan(nz[:, -10:]).all()
    inner = nz[10:-10, 10:-10]
    assert np.allclose(np.abs(inner), 1.0, atol=1e-6)
    # tilted plane z = 2 + x: normal ~ (-1,0,1)/sqrt2 -> |nz| = 0.7071
    cl = _plane_cloud(48, 40, nx=1.0)
    nz, _ = oracle.normals_nz(cl)
    assert np.allclose(np.abs(nz[10:-10, 10:-10]), np.sqrt(0.5), atol=1e-5)


def test_normals_distance_map_and_depth_edge(oracle):
    W, H = 64, 48
    cl = _plane_cloud(W, H)
    cl[:, 32:, 2] += 1.0   # depth jump between columns 31 and 32: |dz| = 1 > 0.02*(2+1)*2
    nz, dist = oracle.normals_nz(cl)
    assert (dist[:-1, 31] == 0).all() and (dist[:-1, 32] == 0).all()  # the last row is never a pair origin
    assert dist[20, 30] == 1.0 and dist[20, 29] == 2.0 and dist[20, 33] == 1.0
    # smoothing <= 2 -> NaN next to the edge, valid from distance 3 on
    assert np.isnan(nz[20, 30]) and np.isnan(nz[20, 29]) and np.isnan(nz[20, 31])
    assert not np.isnan(nz[20, 28]) and not np.isnan(nz[20, 35])
    # far from the edge the window is the full 10
    assert dist[20, 15] >= 10
    # NaN points: marked as change, excluded
    cl2 = _plane_cloud(W, H)
    cl2[24, 30] = np.nan
    nz2, dist2 = oracle.normals_nz(cl2)
    assert dist2[24, 30] == 0 and np.isnan(nz2[24, 30])
    assert dist2[24, 31] == 0  # the right neighbour pair is marked too
    assert abs(abs(nz2[24, 36]) - 1) < 1e-6


def test_acos_matches_libm_to_float_rounding(oracle):
    rng = np.random.default_rng(4)
    xs = np.concatenate([rng.random(4000).astype(np.float32), np.array([0, 0.5, 1, 0.49999997, 0.99999994], np.float32)])
    bad = 0
    for x in xs:
        got = np.float32(oracle.acos_f32(x))
        want = np.float32(np.arccos(np.float64(x)))
        if got != want:
            bad += 1
            assert abs(float(got) - float(want)) <= np.spacing(want)
    assert bad <= 2


# ------------------------------------------------------------------------------------------ extract
def test_extract_reference_owned_rules(oracle):
    """mask rule, stride grid, half = int(patch/(2.0*depth)), layout 3*r*r + depth, height, normal."""
    W, H = 64, 48
    p = oracle.default_params(width=W, height=H, stride=2, patch_size=7, patch_size_reduce=3,
                              depth_min=0.5, depth_max=15.0)
    assert oracle.feature_length(p) == 3 * 3 * 3 + 3
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    depth = np.full((H, W), 1000, np.uint16)
    depth[0, 0] = 0
    depth[0, 2] = 499
    depth[0, 4] = 500
    depth[0, 6] = 15000
    depth[0, 8] = 15001
    depth[20, 20] = 3500   # half = int(7/(2*3.5)) = 1 -> 3x3 ROI = copy
    depth[20, 22] = 700    # half = int(7/1.4) = 5 (7/1.4 = 5.000000000000001 in double? -> int)
    calib = np.concatenate([np.eye(3).ravel() / 50, np.eye(3).ravel(), [0, 0, 0.25]]).astype(np.float32)
    feat, xv, yv = oracle.extract(p, rgb, depth, calib)
    pts = list(zip(xv.tolist(), yv.tolist()))
    assert (0, 0) not in pts and (2, 0) not in pts and (8, 0) not in pts
    assert (4, 0) in pts and (6, 0) in pts
    assert all(x % 2 == 0 and y % 2 == 0 for x, y in pts)
    assert pts == sorted(pts, key=lambda t: (t[1], t[0]))  # row-major order of the stride grid
    lab = oracle.bgr2lab(rgb)
    i = pts.index((20, 20))
    assert np.array_equal(feat[i, :27], lab[19:22, 19:22].reshape(-1).astype(np.float32))
    assert feat[i, 27] == np.float32(3500) / np.float32(1000)
    cl = oracle.cloud(p, depth, calib)
    assert feat[i, 28] == cl[20, 20, 2]
    half = int(7 / (2.0 * float(np.float32(700) / np.float32(1000))))
    j = pts.index((22, 20))
    assert np.array_equal(feat[j, :27], oracle.resize_patch(lab, 22 - half, 20 - half, 2 * half + 1, 3).reshape(-1).astype(np.float32))
    # border points have NaN normals -> -2; interior of the constant plane -> acos(1) = 0 ... or
    # NaN next to the depth spikes
    k = pts.index((4, 0))
    assert feat[k, 29] == -2.0
    m = pts.index((40, 30))
    assert feat[m, 29] == oracle.acos_f32(1.0) or abs(feat[m, 29]) < 1e-3


def test_labels_rules(oracle):
    v = np.array([[-1000, -1000, -1000], [-3, -1, -2], [-1, -1, -5], [0, 0, 0]], np.float32)
    assert oracle.labels(v, 3, 0).tolist() == [-1, 1, 0, 0]            # test.cpp:160-175
    assert oracle.labels(v, 3, 2, unknown=2).tolist() == [2, 1, 0, 2]  # sum != 0 guard -> unknown
    q = np.array([[0.5, 0.3, 0.2], [0.7, 0.2, 0.1], [0.1, 0.2, 0.7], [2 / 3, 1 / 3, 0]], np.float32)
    # strict '>' from 2.0/C = 0.6667
    assert oracle.labels(q, 3, 1, unknown=9).tolist() == [9, 0, 2, 9 if np.float32(2 / 3) <= np.float32(2.0 / 3) else 0]
    assert oracle.labels(q, 3, 3).tolist() == [0, 0, 2, 0]
