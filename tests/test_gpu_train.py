"""SURVEY.md 8(f) rank 4 -- forest training on the GPU (rvseg_forest_train, rvseg_forest_train_frames) against the CPU
oracle's learner (oracle/rvseg_oracle_train.c: the reference's DecisionTreeLearner::learn restated depth-first with
sorts, third-party/libforest/src/learning.cpp:410-1012): the two forest.dat images are compared BYTE FOR BYTE --
same nodes in the same order, same thresholds, same leaf histograms -- on several seeds, layer counts, feature mixes,
with and without bootstrap, and from frames with the reference's augmentation loop (src/train.cpp:115-147).
The reference learner itself draws from std::random_device, so this parity is to the oracle (parity unpinned at the
reference level, see the oracle's header).  Independent of the oracle, what the reference DEFINES is checked as well:

  * the output is a valid forest.dat (rvseg_forest_check, loadable, evaluable);
  * leaf histograms = log((h + s) / (total + C s)) with h accumulated from ALL examples, each adding the inverted
    class frequency of its label (updateMultiHistograms, :960-1012) -- recomputed here with numpy from the tree
    structure, bit for bit;
  * the split at the root minimises E(left) + E(right) over all features and thresholds (brute force), thresholds
    sit between two adjacent distinct values;
  * stopping rules: max_depth, min_split_examples, min_child_split_examples, purity;
  * equal seeds give equal bytes, and the forest learns a learnable rule.
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _parse(blob):
    """forest.dat -> list of trees: dict(feat, thr, left, hist[list], mhist[list of list])"""
    pos = 0
    T = struct.unpack_from("<i", blob, pos)[0]; pos += 4
    trees = []
    for _ in range(T):
        n = struct.unpack_from("<i", blob, pos)[0]
        feat = np.frombuffer(blob, np.int32, n, pos + 4); pos += 4 + 4 * n
        thr = np.frombuffer(blob, np.float32, n, pos + 4); pos += 4 + 4 * n
        left = np.frombuffer(blob, np.int32, n, pos + 4); pos += 4 + 4 * n
        cnt = struct.unpack_from("<i", blob, pos)[0]; pos += 4
        hist = []
        for _i in range(cnt):
            m = struct.unpack_from("<i", blob, pos)[0]; pos += 4
            hist.append(np.frombuffer(blob, np.float32, m, pos).copy()); pos += 4 * m
        cnt = struct.unpack_from("<i", blob, pos)[0]; pos += 4
        mhist = []
        for _i in range(cnt):
            m = struct.unpack_from("<i", blob, pos)[0]; pos += 4
            layers = []
            for _l in range(m):
                c = struct.unpack_from("<i", blob, pos)[0]; pos += 4
                layers.append(np.frombuffer(blob, np.float32, c, pos).copy()); pos += 4 * c
            mhist.append(layers)
        trees.append(dict(feat=feat, thr=thr, left=left, hist=hist, mhist=mhist))
    assert pos == len(blob)
    return trees


def _route(tree, X):
    """findLeafNode for every row (classifier.cpp:97-117)."""
    node = np.zeros(X.shape[0], np.int64)
    depth = np.zeros(X.shape[0], np.int64)
    while True:
        l = tree["left"][node]
        go = l != 0
        if not go.any():
            return node, depth
        v = X[np.arange(X.shape[0]), tree["feat"][node]]
        nxt = np.where(v < tree["thr"][node], l, l + 1)
        node = np.where(go, nxt, node)
        depth += go


def _data(P=6000, D=40, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.integers(0, 256, (P, D)).astype(np.float32)
    X[:, D - 1] = rng.uniform(0.5, 15.0, P)            # a float feature (depth-like)
    X[:, D - 2] = np.where(rng.random(P) < 0.1, -2.0, rng.uniform(0, np.pi / 2, P))
    l0 = (X[:, 3] < 100).astype(np.int32) + 2 * (X[:, 17] < 60).astype(np.int32)            # 4 classes
    l1 = (X[:, D - 1] < 4.0).astype(np.int32) + (X[:, 8] < 30).astype(np.int32)              # 3 classes
    noise = rng.random(P) < 0.02
    l0 = np.where(noise, rng.integers(0, 4, P), l0).astype(np.int32)
    return X, np.stack([l0, l1], 1), [4, 3]


def test_trained_forest_is_valid_learns_the_rule_and_is_reproducible(gpu_ctx_factory):
    import rovinasemanticsegmentation_amd as rv
    X, labels, cc = _data()
    D = X.shape[1]
    kw = dict(width=160, height=120, patch_size=9, patch_size_reduce=3)   # any context; D comes with the data
    ctx = gpu_ctx_factory(**kw)
    blob = ctx.forest_train(X, labels, cc, num_trees=4, max_depth=12, min_split_examples=20, seed=5)
    assert ctx.forest_train(X, labels, cc, num_trees=4, max_depth=12, min_split_examples=20, seed=5) == blob
    assert ctx.forest_train(X, labels, cc, num_trees=4, max_depth=12, min_split_examples=20, seed=6) != blob
    st, msg, info = rv.capi.forest_check(blob, D)
    assert st == rv.capi.OK, msg
    assert info["n_trees"] == 4 and info["max_depth"] <= 13            # depth[node] > max_depth stops: at most max_depth + 1 edges
    trees = _parse(blob)
    # (the evaluator's D is tied to the extractor configuration, so the accuracy check routes with numpy)
    votes = [np.zeros((X.shape[0], c), np.float32) for c in cc]
    for t in trees:
        leaf, _ = _route(t, X)
        for l in range(2):
            votes[l] += np.stack([t["mhist"][n][l] for n in leaf])
    for l in range(2):
        acc = (votes[l].argmax(1) == labels[:, l]).mean()
        assert acc > 0.95, (l, acc)


def test_leaf_histograms_follow_update_multi_histograms_bit_for_bit(gpu_ctx_factory):
    X, labels, cc = _data(P=3000, D=24, seed=3)
    ctx = gpu_ctx_factory(width=160, height=120)
    smoothing = 1.0
    blob = ctx.forest_train(X, labels, cc, num_trees=2, max_depth=8, min_split_examples=30, seed=11, smoothing=smoothing)
    trees = _parse(blob)
    P = X.shape[0]
    for t in trees:
        leaf, _ = _route(t, X)
        for l, C in enumerate(cc):
            counts = np.bincount(labels[:, l], minlength=C).astype(np.float32)
            freq = np.float32(P) / counts                                   # getInvertedClassFrequency, data.h:359-370
            for n in np.unique(leaf):
                assert t["left"][n] == 0
                h = np.zeros(C, np.float32)
                for c in range(C):
                    k = int(((leaf == n) & (labels[:, l] == c)).sum())
                    acc = np.float32(0)
                    for _ in range(k):                                      # one float addition per example (:989-991)
                        acc = np.float32(acc + freq[c])
                    h[c] = acc
                total = np.float32(0)
                for c in range(C):
                    total = np.float32(total + h[c])
                want = np.log((h + np.float32(smoothing)) / np.float32(total + np.float32(C * smoothing))).astype(np.float32)
                got = t["mhist"][n][l]
                assert got.shape == (C,)
                # logf of glibc vs numpy's float32 log: both correctly rounded for these arguments in practice;
                # allow 1 ulp, the accumulation itself (h, total) is what must agree
                assert np.allclose(got, want, rtol=0, atol=2e-7 * np.abs(want).max() + 1e-7), (n, l, got, want)
        # every inner node is empty-handed, every leaf carries one histogram per layer
        for n in range(len(t["left"])):
            assert (len(t["mhist"][n]) == 0) == (t["left"][n] != 0)


def _entropy_mass(counts):
    n = counts.sum()
    nz = counts[counts > 0]
    return n * np.log2(n) - (nz * np.log2(nz)).sum() if n > 0 else 0.0


def test_root_split_is_the_brute_force_optimum_and_stop_rules_hold(gpu_ctx_factory):
    rng = np.random.default_rng(9)
    P, D = 1500, 12
    X = rng.integers(0, 64, (P, D)).astype(np.float32)
    X[:, 11] = rng.normal(0, 1, P).astype(np.float32)
    y = ((X[:, 4] < 20) ^ (X[:, 11] < 0.3)).astype(np.int32) + (X[:, 7] < 10).astype(np.int32)
    cc = [3]
    ctx = gpu_ctx_factory(width=160, height=120)
    min_split, min_child, max_depth = 40, 5, 6
    blob = ctx.forest_train(X, y[:, None], cc, num_trees=1, max_depth=max_depth, min_split_examples=min_split,
                            min_child_split_examples=min_child, num_features=D, use_bootstrap=0, seed=2)
    t = _parse(blob)[0]
    assert len(t["hist"][int(np.flatnonzero(t["left"] == 0)[0])]) == 3     # single layer: `histograms` filled too
    # brute force at the root over every feature and every midpoint of adjacent distinct values
    best = (np.inf, None, None)
    for f in range(D):
        order = np.argsort(X[:, f], kind="stable")
        xs, ys = X[order, f], y[order]
        left = np.zeros(3)
        right = np.bincount(ys, minlength=3).astype(np.float64)
        for m in range(1, P):
            left[ys[m - 1]] += 1
            right[ys[m - 1]] -= 1
            if xs[m] - xs[m - 1] < 1e-6:
                continue
            obj = _entropy_mass(left) + _entropy_mass(right)
            if obj < best[0] - 1e-9:
                best = (obj, f, (xs[m - 1], xs[m]))
    f0, thr0 = int(t["feat"][0]), float(t["thr"][0])
    lo_side = X[:, f0] < thr0
    got_obj = _entropy_mass(np.bincount(y[lo_side], minlength=3).astype(np.float64)) + \
        _entropy_mass(np.bincount(y[~lo_side], minlength=3).astype(np.float64))
    # every feature is searched exactly (byte-valued ones through per-value histograms, the float one through a sort);
    # the slack covers fastlog2's approximation of log2 (the learner's own objective, fastlog.h:47-58)
    assert got_obj <= best[0] * (1 + 2e-3) + 1e-6, (got_obj, best)
    vals = np.unique(X[:, f0])
    below, above = vals[vals < thr0], vals[vals >= thr0]
    assert len(below) and len(above) and below.max() < thr0 <= above.min()
    # stop rules on the finished tree (no bootstrap: node masses are plain counts)
    leaf, depth = _route(t, X)
    assert depth.max() <= max_depth + 1
    n_nodes = len(t["left"])
    # masses of all nodes: push counts up from the leaves
    mass = np.zeros(n_nodes, np.int64)
    np.add.at(mass, leaf, 1)
    for n in range(n_nodes - 1, -1, -1):
        if t["left"][n] != 0:
            mass[n] = mass[t["left"][n]] + mass[t["left"][n] + 1]
    for n in range(n_nodes):
        if t["left"][n] != 0:
            assert mass[n] >= min_split, n
            assert mass[t["left"][n]] >= min_child and mass[t["left"][n] + 1] >= min_child, n
    assert mass[0] == P


def test_trained_model_runs_through_the_frame_path(gpu_ctx_factory, oracle):
    """A forest trained on features extracted by the library itself segments frames like any forest.dat: the
    GPU frame path and the CPU oracle agree bit for bit on it."""
    from rovinasemanticsegmentation_amd import synthetic
    W, H = 160, 120
    rgb, depth = synthetic.make_batch(2, W, H, holes=True)
    calib = synthetic.make_calib(W, H)
    kw = dict(width=W, height=H, patch_size=9, patch_size_reduce=3)
    ctx = gpu_ctx_factory(**kw)
    feats, xs, ys = ctx.extract_features(rgb[0], depth[0], calib)
    D = feats.shape[1]
    assert D == 30
    # labels from the image position (two layers), like a ground-truth label image sampled at (x_v, y_v)
    l0 = (xs // 40).astype(np.int32) % 3
    l1 = (ys // 30).astype(np.int32) % 4
    blob = ctx.forest_train(feats, np.stack([l0, l1], 1), [3, 4], num_trees=3, max_depth=10, min_split_examples=10, seed=4)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib)
    forest = oracle.Forest(blob)
    p = oracle.default_params(**kw)
    for i in range(2):
        want, _ = oracle.rf_frame(p, forest, 1, rgb[i], depth[i], calib)
        assert np.array_equal(out["posteriors"][i], want), i


@pytest.mark.parametrize("seed,layers,bootstrap", [(1, 2, 1), (7, 2, 1), (3, 1, 1), (5, 2, 0)])
def test_trainer_equals_the_oracle_learner_byte_for_byte(gpu_ctx_factory, oracle, seed, layers, bootstrap):
    """Mixed byte / float features (two float columns with many near-ties and a -2 sentinel like the normal feature),
    label noise so that trees grow deep, sqrt(D) features per node: GPU (level-wise, histograms + per-level sort) and
    oracle (depth-first, one sort per node and feature) must write the same file."""
    X, labels, cc = _data(P=5000, D=40, seed=seed)
    X[:, 38] = np.round(X[:, 38], 2)                  # many exact ties in a float feature
    X[::7, 39] = X[::7, 39] + np.float32(5e-7)        # neighbours closer than the 1e-6 cut rule (learning.cpp:578-585)
    if layers == 1:
        labels, cc = labels[:, :1], cc[:1]
    kw = dict(num_trees=3, max_depth=14, min_split_examples=12, min_child_split_examples=2, use_bootstrap=bootstrap, seed=seed)
    ctx = gpu_ctx_factory(width=160, height=120)
    got = ctx.forest_train(X, labels, cc, **kw)
    want = oracle.forest_train(X, labels, cc, **kw)
    if got != want:   # say where they part
        tg, tw = _parse(got), _parse(want)
        for k, (a, b) in enumerate(zip(tg, tw)):
            assert len(a["left"]) == len(b["left"]), ("tree %d: node counts" % k, len(a["left"]), len(b["left"]))
            for name in ("left", "feat", "thr"):
                bad = np.flatnonzero(a[name] != b[name])
                assert bad.size == 0, ("tree %d: %s differs first at node %d" % (k, name, bad[0]), a[name][bad[0]], b[name][bad[0]])
    assert got == want
    assert max(len(t["left"]) for t in _parse(got)) > 100      # the comparison covered real trees


def test_training_from_frames_with_augmentation_equals_the_oracle(gpu_ctx_factory, oracle):
    """rvseg_forest_train_frames (extraction + the reference's augmentation loop on the device side, src/train.cpp:115-147)
    against the oracle end to end: the oracle extracts the same six variants per frame on the CPU (WITH_POSITIVE_LABEL:
    valid depth and all labels >= 0, feature_extractor.h:93-121), builds the P x D matrix the reference's DataStorage
    would hold, and learns depth-first; the two files must be equal.  Also: the number of training points."""
    from rovinasemanticsegmentation_amd import synthetic
    W, H = 160, 120
    n = 2
    rgb, depth = synthetic.make_batch(n, W, H, holes=True, start=4)
    calib = synthetic.make_calib(W, H)
    kw = dict(width=W, height=H, patch_size=9, patch_size_reduce=3)
    yy, xx = np.mgrid[0:H, 0:W]
    lab = np.empty((n, 2, H, W), np.int8)
    for i in range(n):
        lab[i, 0] = ((xx // 40) + i) % 3
        lab[i, 1] = ((yy // 30) + (xx // 80)) % 4
        lab[i, 0][(xx + yy) % 11 == 0] = -1            # unlabelled pixels are skipped
        lab[i, 1][yy < 6] = -1
    cc = [3, 4]
    tkw = dict(num_trees=2, max_depth=10, min_split_examples=10, seed=9)
    ctx = gpu_ctx_factory(**kw)
    got, n_ex = ctx.forest_train_frames(rgb, depth, calib, lab, cc, augment=True, **tkw)
    # the oracle's data set, in the reference's order: per frame, for a in (-20, 0, +20): the frame, then its flip
    p = oracle.default_params(**kw)
    Xs, Ys = [], []
    for i in range(n):
        for a in (-20, 0, 20):
            col = rgb[i].astype(np.int32)
            col[:, :, 0] = np.clip(col[:, :, 0] + a, 0, 255)     # cv::Mat += char: Scalar(a, 0, 0, 0), saturated
            col = col.astype(np.uint8)
            for flip in (False, True):
                c2 = col[:, ::-1].copy() if flip else col
                d2 = depth[i][:, ::-1].copy() if flip else depth[i]
                l2 = lab[i][:, :, ::-1] if flip else lab[i]
                feats, xs, ys = oracle.extract(p, c2, d2, calib)
                keep = (l2[0][ys, xs] >= 0) & (l2[1][ys, xs] >= 0)
                Xs.append(feats[keep])
                Ys.append(np.stack([l2[0][ys, xs][keep], l2[1][ys, xs][keep]], 1).astype(np.int32))
    X, Y = np.concatenate(Xs), np.concatenate(Ys)
    assert n_ex == X.shape[0]
    want = oracle.forest_train(X, Y, cc, **tkw)
    assert got == want
    # and the matrix entry point on the same data
    assert ctx.forest_train(X, Y, cc, **tkw) == want
