"""CPU: the oracle's local-map fusion (src/segmenter.cpp:561-616) against hand-computed cases.
The reference holds no fixture for this loop (fps_mapper's projector is not in the tree), so the
cases below are derived from the cited lines directly."""
import numpy as np

from rovinasemanticsegmentation_amd import synthetic


def test_fusion_known_answer_two_layers(oracle):
    H, W, cc = 2, 3, [2, 3]
    idx = np.array([[[0, -1, 2], [1, 0, -1]],
                    [[2, 2, -1], [-1, 1, 0]]], np.int32)
    rng = np.random.default_rng(0)
    post = rng.integers(-8, 8, (2, sum(cc) * H * W)).astype(np.float32)
    got = oracle.fuse_posteriors(idx, post, cc, 4)
    want = [np.zeros((4, c), np.float32) for c in cc]
    for m in range(2):
        off = 0
        for l, c in enumerate(cc):
            for pix in range(H * W):
                k = idx[m].ravel()[pix]
                if k >= 0:
                    want[l][k] += post[m, off + pix * c: off + (pix + 1) * c]
            off += H * W * c
    assert np.array_equal(got, np.concatenate([w.ravel() for w in want]))
    assert np.all(got.reshape(-1)[3 * 2:4 * 2] == 0)          # point 3 is never seen: stays 0 (:566)


def test_fusion_order_is_image_then_raster(oracle):
    # fp32: (1e8 + 1) - 1e8 = 0 but (1e8 - 1e8) + 1 = 1 -- the sum must run in (image, pixel) order
    H, W = 1, 3
    idx = np.array([[[0, 0, -1]], [[-1, 0, -1]]], np.int32)     # point 0: image 0 pixels 0,1 then image 1 pixel 1
    post = np.array([[1e8, 1.0, 5.0], [7.0, -1e8, 7.0]], np.float32)
    got = oracle.fuse_posteriors(idx, post, [1], 1)
    assert got[0] == np.float32(np.float32(np.float32(0.0 + 1e8) + np.float32(1.0)) + np.float32(-1e8)) == 0.0
    post2 = np.array([[1e8, -1e8, 5.0], [7.0, 1.0, 7.0]], np.float32)
    assert oracle.fuse_posteriors(idx, post2, [1], 1)[0] == 1.0


def test_projector_stand_in_is_a_zbuffer():
    calib = synthetic.make_calib(64, 48)
    pts = np.array([[2.0, 0.0, 0.6], [4.0, 0.0, 0.6], [2.0, 0.5, 0.6]], np.float32)   # base frame: x forward
    idx = synthetic.project_cloud(pts, calib, 64, 48)
    assert idx[24, 32] == 0                      # the nearer of the two points on the optical axis
    assert (idx == 1).sum() == 0
    assert (idx == 2).sum() == 1 and np.argwhere(idx == 2)[0][1] < 32   # +y (left of the camera) lands left of centre
