"""Seeded synthetic inputs for tests and bench.py.

The trained model and the ROVINA data set are not obtainable offline (resources/get_rf_model.sh:2,
README.md:4 of the reference), so every forest and frame is synthetic:
  * RGB    piece-wise constant 80-px blocks + linear ramps + uniform +-6 noise
  * depth  planes 1500 + 2000*x/W mm with 500-mm checkerboard steps every 160 px, +-5 mm noise;
           optional 10 % zero-depth holes
  * calib  K = [525,0,320; 0,525,240; 0,0,1] (scaled with the image), camera-to-base rotation,
           t = (0,0,0.6)
  * forest T trees grown top-down by random splits, written in the libforest stream format
           (third-party/libforest/src/classifier.cpp:144-152,210-220; io.h:84-96)
  * local map: a cloud back-projected from frame 0 and pin-hole z-buffer index images -- a stand-in
           for fps_mapper::MultiProjector::project (src/segmenter.cpp:578), which is not in the tree
"""
import struct

import numpy as np


def make_calib(W=640, H=480):
    s = W / 640.0
    fx = fy = 525.0 * s
    cx, cy = W / 2.0, H / 2.0
    Kinv = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], np.float64)
    R = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)  # camera (x right, z fwd) -> base (z up)
    t = np.array([0.0, 0.0, 0.6])
    return np.concatenate([Kinv.ravel(), R.ravel(), t]).astype(np.float32)


def make_deep_frame(index=0, W=640, H=480, holes=False, seed=1234):
    """A scene with a deep range and textured colour: a ground plane receding from 1 m to 10 m, box-shaped
    obstacles, slowly varying saturated colours.  The Segmenter's CRF kernel (xyz * 0.5, rgb * 4) gives such a frame
    1 500+ lattice vertices instead of the ~300 of `make_frame` -- the regime of a real photo (1 660-2 240
    on the reference's im2.ppm with a 1-10 m depth)."""
    rng = np.random.default_rng(seed + 7919 * (index + 1))
    ys, xs = np.mgrid[0:H, 0:W]
    fy = ys / float(H)
    depth = 10000.0 - 9000.0 * fy ** 0.7
    for k in range(6):   # obstacles: columns of the image 0.6 - 2.4 m in front of the ground behind them
        x0 = int((0.08 + 0.15 * k + 0.02 * ((index + k) % 3)) * W)
        w = int(0.07 * W)
        top = int((0.15 + 0.1 * ((k + index) % 4)) * H)
        sel = (xs >= x0) & (xs < x0 + w) & (ys >= top)
        depth = np.where(sel, np.maximum(900.0, depth - (600.0 + 300.0 * k)), depth)
    depth = depth + rng.integers(-5, 6, size=(H, W))
    ph = 0.7 * index
    r = 128 + 100 * np.sin(xs * (50.0 / W) * 0.25 + ph)
    g = 128 + 100 * np.sin(ys * (37.0 / H) * 0.35 + 1.3 + ph)
    b = 128 + 100 * np.sin((xs / float(W) + ys / float(H)) * 9.0 + 2.1 - ph)
    rgb = np.stack([r, g, b], -1) + rng.integers(-8, 9, size=(H, W, 3))
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    depth = np.clip(depth, 500, 15000).astype(np.uint16)
    if holes:
        hole = rng.random((H // 8 + 1, W // 8 + 1)) < 0.10
        hole = np.kron(hole, np.ones((8, 8), bool))[:H, :W]
        depth[hole] = 0
    return rgb, depth


def make_frame(index=0, W=640, H=480, holes=False, seed=1234, scene="flat"):
    if scene == "deep":
        return make_deep_frame(index, W, H, holes, seed)
    rng = np.random.default_rng(seed + index)
    ys, xs = np.mgrid[0:H, 0:W]
    base = np.array([[180, 60, 50], [60, 170, 70], [50, 80, 190]], np.float64)
    blk = ((xs // 80) + 2 * (ys // 80) + index) % 3
    rgb = base[blk]
    rgb = rgb + 40.0 * xs[..., None] / W + 25.0 * ys[..., None] / H
    rgb = rgb + rng.integers(-6, 7, size=(H, W, 3))
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    checker = (((xs // 160) + (ys // 160)) % 2) * 500
    depth = 1500 + 2000.0 * xs / W + checker + rng.integers(-5, 6, size=(H, W)) + 37 * (index % 5)
    depth = np.clip(depth, 500, 15000).astype(np.uint16)
    if holes:
        hole = rng.random((H // 8 + 1, W // 8 + 1)) < 0.10
        hole = np.kron(hole, np.ones((8, 8), bool))[:H, :W]
        depth[hole] = 0
    return rgb, depth


def make_batch(n, W=640, H=480, holes=False, seed=1234, start=0, scene="flat"):
    rgb = np.empty((n, H, W, 3), np.uint8)
    depth = np.empty((n, H, W), np.uint16)
    for i in range(n):
        rgb[i], depth[i] = make_frame(start + i, W, H, holes, seed, scene)
    return rgb, depth


def _feature_range(f, D):
    if D == 366:
        if f == 363:
            return 0.5, 15.0
        if f == 364:
            return -1.0, 3.0
        if f == 365:
            return 0.0, np.pi / 2
    return 0.0, 255.0


def make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, D=366,
                      single_classes=9, layer_classes=(8, 9)):
    """Random forest in the reference's forest.dat format.  Children are appended pair-wise at
    split time from a LIFO/random open list, so node ids are NOT breadth-first (like the learner,
    learning.cpp:650-651)."""
    rng = np.random.default_rng(seed)
    out = [struct.pack("<i", n_trees)]
    for _ in range(n_trees):
        feat, thr, left, depth = [0], [0.0], [0], [0]
        open_nodes = [0]
        n_leaves = 1
        while open_nodes and n_leaves < leaves_per_tree:
            k = len(open_nodes) - 1 if rng.random() < 0.5 else int(rng.integers(len(open_nodes)))
            node = open_nodes.pop(k)
            if depth[node] >= max_depth:
                continue
            f = int(rng.integers(D))
            lo, hi = _feature_range(f, D)
            th = lo + (hi - lo) * rng.random()
            if hi == 255.0 and rng.random() < 0.34:
                th = np.floor(th)  # ties on byte-valued features exercise the strict '<'
            feat[node], thr[node] = f, float(np.float32(th))
            l = len(feat)
            left[node] = l
            for _c in range(2):
                feat.append(0); thr.append(0.0); left.append(0); depth.append(depth[node] + 1)
            open_nodes += [l, l + 1]
            n_leaves += 1
        n = len(feat)
        left_a = np.asarray(left, np.int32)
        is_leaf = left_a == 0
        out.append(struct.pack("<i", n) + np.asarray(feat, np.int32).tobytes())
        out.append(struct.pack("<i", n) + np.asarray(thr, np.float32).tobytes())
        out.append(struct.pack("<i", n) + left_a.tobytes())

        def log_hist(C):
            p = 1e-3 + (1 - 1e-3) * rng.random((n, C))
            return np.log(p / p.sum(1, keepdims=True)).astype(np.float32)

        # histograms: per node int32 count + floats (inner nodes: count 0)
        rec = [struct.pack("<i", n)]
        if single_classes:
            h = log_hist(single_classes)
            for i in range(n):
                rec.append(struct.pack("<i", single_classes) + h[i].tobytes() if is_leaf[i] else b"\0\0\0\0")
        else:
            rec.append(b"\0\0\0\0" * n)
        out.append(b"".join(rec))
        rec = [struct.pack("<i", n)]
        if layer_classes:
            hs = [log_hist(c) for c in layer_classes]
            for i in range(n):
                if is_leaf[i]:
                    rec.append(struct.pack("<i", len(layer_classes)) + b"".join(
                        struct.pack("<i", c) + hs[l][i].tobytes() for l, c in enumerate(layer_classes)))
                else:
                    rec.append(b"\0\0\0\0")
        else:
            rec.append(b"\0\0\0\0" * n)
        out.append(b"".join(rec))
    return b"".join(out)


def random_points(seed, P, D=366):
    rng = np.random.default_rng(seed)
    X = rng.integers(0, 256, (P, D)).astype(np.float32)
    if D == 366:
        X[:, 363] = rng.uniform(0.5, 15.0, P)
        X[:, 364] = rng.uniform(-1.0, 3.0, P)
        X[:, 365] = np.where(rng.random(P) < 0.1, -2.0, rng.uniform(0, np.pi / 2, P))
    return X


def back_project(depth, calib, W=640, H=480):
    """(R K^-1) (d x, d y, d) + t for every pixel with depth > 0 (feature_extractor.h:209-223), float64."""
    c = np.asarray(calib, np.float64)
    Kinv, R, t = c[:9].reshape(3, 3), c[9:18].reshape(3, 3), c[18:21]
    ys, xs = np.mgrid[0:H, 0:W]
    d = depth.astype(np.float64) / 1000.0
    m = np.stack([d * xs, d * ys, d], -1).reshape(-1, 3)
    pts = m @ (R @ Kinv).T + t
    return pts, (depth.ravel() > 0)


def project_cloud(cloud_xyz, calib, W=640, H=480, shift=(0.0, 0.0, 0.0)):
    """Pin-hole z-buffer: IndexImage (int32, -1 = no point) of the cloud seen from the calibration's
    camera moved by `shift` (base frame).  The nearest point wins a pixel."""
    c = np.asarray(calib, np.float64)
    Kinv, R, t = c[:9].reshape(3, 3), c[9:18].reshape(3, 3), c[18:21]
    K = np.linalg.inv(Kinv)
    cam = (np.asarray(cloud_xyz, np.float64) - (t + np.asarray(shift, np.float64))) @ R   # R^T (p - t)
    z = cam[:, 2]
    ok = z > 0.05
    uvw = cam @ K.T
    u = np.rint(uvw[:, 0] / np.where(ok, z, 1.0)).astype(np.int64)
    v = np.rint(uvw[:, 1] / np.where(ok, z, 1.0)).astype(np.int64)
    ok &= (u >= 0) & (u < W) & (v >= 0) & (v < H)
    idx = np.full(H * W, -1, np.int32)
    order = np.argsort(-z[ok], kind="stable")          # far first, so the nearest point is written last
    pts = np.nonzero(ok)[0][order]
    idx[v[pts] * W + u[pts]] = pts.astype(np.int32)
    return idx.reshape(H, W)


def make_local_map(n_frames, W=640, H=480, step=4, seed=1234):
    """Frames + a cloud (every `step`-th pixel of frame 0, base frame, rgb in [0,1]) + one index image
    per frame (the camera drifts a few centimetres per frame)."""
    rgb, depth = make_batch(n_frames, W, H, seed=seed)
    calib = make_calib(W, H)
    pts, valid = back_project(depth[0], calib, W, H)
    sel = np.zeros((H, W), bool)
    sel[::step, ::step] = True
    sel = sel.ravel() & valid
    cloud_xyz = pts[sel].astype(np.float32)
    cloud_rgb = (rgb[0].reshape(-1, 3)[sel].astype(np.float32) / np.float32(255.0))
    index_images = np.stack([project_cloud(cloud_xyz, calib, W, H, shift=(0.01 * i, -0.02 * i, 0.005 * i)) for i in range(n_frames)])
    return rgb, depth, calib, cloud_xyz, cloud_rgb, index_images
