"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/rvseg_oracle.c).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
module; the product package never does.  See rvseg_oracle.h for what is pinned and what is not.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcParams(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int),
        ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("patch_size", C.c_int), ("patch_size_reduce", C.c_int),
        ("feature_color_patch", C.c_int), ("feature_depth", C.c_int),
        ("feature_height", C.c_int), ("feature_normal", C.c_int),
        ("fill_value", C.c_float),
        ("dcrf_xyz_kernel", C.c_float), ("dcrf_rgb_kernel", C.c_float),
        ("dcrf_kernel_weight", C.c_float), ("dcrf_iterations", C.c_int),
    ]


class _Lattice(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("d", C.c_int), ("M", C.c_int),
        ("offset", C.POINTER(C.c_int)), ("barycentric", C.POINTER(C.c_float)),
        ("rank", C.POINTER(C.c_float)),
        ("blur_n1", C.POINTER(C.c_int)), ("blur_n2", C.POINTER(C.c_int)),
        ("keys", C.POINTER(C.c_short)),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("rvseg_oracle.c", "rvseg_oracle_train.c", "rvseg_oracle.h")]
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "lib"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, ip, fp = C.c_void_p, C.c_int, C.c_float
        L.orc_forest_load_mem.restype = vp
        L.orc_forest_load_mem.argtypes = [vp, C.c_size_t]
        L.orc_forest_free.argtypes = [vp]
        L.orc_forest_single_classes.argtypes = [vp]
        L.orc_forest_layers.argtypes = [vp, vp, ip]
        L.orc_forest_eval.argtypes = [vp, vp, ip, ip, ip, vp]
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        L.orc_feature_length.argtypes = [C.POINTER(OrcParams)]
        L.orc_bgr2lab_u8.argtypes = [vp, vp, ip]
        L.orc_resize_patch_u8.argtypes = [vp, ip, ip, ip, ip, ip, ip, vp]
        L.orc_resize_linear_f32.argtypes = [vp, ip, ip, ip, vp, ip, ip]
        L.orc_cloud.argtypes = [C.POINTER(OrcParams), vp, vp, vp]
        L.orc_normals_nz.argtypes = [ip, ip, vp, vp, vp]
        L.orc_acos_f32.restype = fp
        L.orc_acos_f32.argtypes = [fp]
        L.orc_exp_f32.restype = fp
        L.orc_exp_f32.argtypes = [fp]
        L.orc_extract.argtypes = [C.POINTER(OrcParams), vp, vp, vp, vp, vp, vp]
        L.orc_rf_frame.argtypes = [C.POINTER(OrcParams), vp, ip, vp, vp, vp, vp]
        L.orc_labels.argtypes = [vp, ip, ip, ip, ip, vp]
        L.orc_fuse_posteriors.argtypes = [ip, ip, ip, vp, vp, ip, vp, ip, vp]
        L.orc_lattice_init.restype = C.POINTER(_Lattice)
        L.orc_lattice_init.argtypes = [vp, ip, ip]
        L.orc_lattice_free.argtypes = [C.POINTER(_Lattice)]
        for name in ("orc_lattice_compute_seq", "orc_lattice_compute_sse", "orc_lattice_compute"):
            getattr(L, name).argtypes = [C.POINTER(_Lattice), vp, vp, ip, ip]
        L.orc_kernel_norm.argtypes = [C.POINTER(_Lattice), vp]
        L.orc_exp_and_normalize.argtypes = [vp, vp, ip, ip]
        L.orc_crf_inference.argtypes = [ip, ip, ip, vp, vp, fp, ip, vp]
        L.orc_crf_inference_multi.argtypes = [ip, ip, ip, vp, vp, vp, vp, ip, vp]
        L.orc_frame_crf_features.argtypes = [C.POINTER(OrcParams), vp, vp, vp]
        L.orc_segment_frame.argtypes = [C.POINTER(OrcParams), vp, ip, vp, vp, vp, vp, vp, vp, ip, vp]
        L.orc_forest_train.argtypes = [vp, ip, ip, vp, ip, vp, ip, ip, ip, ip, ip, ip, fp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.orc_free.argtypes = [vp]
        L.orc_lab_gamma_tab.restype = C.POINTER(C.c_ushort)
        L.orc_lab_cbrt_tab.restype = C.POINTER(C.c_ushort)
        L.orc_lab_coeffs.restype = C.POINTER(C.c_int)
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def default_params(**kw):
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def feature_length(p):
    return lib().orc_feature_length(C.byref(p))


class Forest:
    """orc_forest wrapper (RandomForest::read / classLogPosterior / multiClassLogPosterior)."""

    def __init__(self, data):
        if isinstance(data, str):
            with open(data, "rb") as fh:
                data = fh.read()
        self._buf = bytes(data)
        self.h = lib().orc_forest_load_mem(self._buf, len(self._buf))
        if not self.h:
            raise ValueError("malformed forest stream")
        self.single_classes = lib().orc_forest_single_classes(self.h)
        cc = (C.c_int * 64)()
        n = lib().orc_forest_layers(self.h, cc, 64)
        self.layers = [cc[i] for i in range(n)]

    def classes(self, multi):
        return list(self.layers) if multi else [self.single_classes]

    def eval(self, X, multi):
        X = np.ascontiguousarray(X, np.float32)
        P, D = X.shape
        S = sum(self.layers) if multi else self.single_classes
        out = np.empty((P, S), np.float32)
        lib().orc_forest_eval(self.h, _p(X), P, D, 1 if multi else 0, _p(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_forest_free(self.h)
            self.h = None


def forest_train(X, labels, class_counts, num_trees=4, max_depth=30, min_split_examples=50, min_child_split_examples=1,
                 num_features=0, use_bootstrap=1, smoothing=1.0, seed=1):
    """orc_forest_train: the CPU learner (depth-first, sorts).  Returns the forest.dat bytes."""
    X = np.ascontiguousarray(X, np.float32)
    labels = np.ascontiguousarray(np.asarray(labels, np.int32).reshape(X.shape[0], -1))
    cc = np.asarray(class_counts, np.int32)
    out = C.c_void_p()
    size = C.c_size_t()
    rc = lib().orc_forest_train(_p(X), X.shape[0], X.shape[1], _p(labels), labels.shape[1], _p(cc), num_trees, max_depth,
                                min_split_examples, min_child_split_examples, num_features, use_bootstrap, smoothing, seed,
                                C.byref(out), C.byref(size))
    if rc != 0:
        raise ValueError("orc_forest_train: bad arguments")
    try:
        return C.string_at(out, size.value)
    finally:
        lib().orc_free(out)


def bgr2lab(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().orc_bgr2lab_u8(_p(img), _p(out), img.size // 3)
    return out


def resize_patch(lab, x0, y0, size, r):
    lab = np.ascontiguousarray(lab, np.uint8)
    H, W, _ = lab.shape
    out = np.empty((r, r, 3), np.uint8)
    lib().orc_resize_patch_u8(_p(lab), W, H, x0, y0, size, r, _p(out))
    return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.float32)
    sh, sw, Cn = src.shape
    out = np.empty((dh, dw, Cn), np.float32)
    lib().orc_resize_linear_f32(_p(src), sw, sh, Cn, _p(out), dw, dh)
    return out


def cloud(p, depth, calib):
    depth = np.ascontiguousarray(depth, np.uint16)
    calib = np.ascontiguousarray(calib, np.float32)
    out = np.empty((p.height, p.width, 3), np.float32)
    lib().orc_cloud(C.byref(p), _p(depth), _p(calib), _p(out))
    return out


def normals_nz(cl):
    cl = np.ascontiguousarray(cl, np.float32)
    H, W, _ = cl.shape
    nz = np.empty((H, W), np.float32)
    dist = np.empty((H, W), np.float32)
    lib().orc_normals_nz(W, H, _p(cl), _p(nz), _p(dist))
    return nz, dist


def acos_f32(x):
    return lib().orc_acos_f32(float(x))


def exp_f32(x):
    return lib().orc_exp_f32(float(x))


def extract(p, rgb, depth, calib):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    calib = np.ascontiguousarray(calib, np.float32)
    D = feature_length(p)
    cap = (p.height // p.stride + 1) * (p.width // p.stride + 1)
    feat = np.empty((cap, D), np.float32)
    xv = np.empty(cap, np.int32)
    yv = np.empty(cap, np.int32)
    P = lib().orc_extract(C.byref(p), _p(rgb), _p(depth), _p(calib), _p(feat), _p(xv), _p(yv))
    return feat[:P].copy(), xv[:P].copy(), yv[:P].copy()


def rf_frame(p, forest, multi, rgb, depth, calib):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    calib = np.ascontiguousarray(calib, np.float32)
    S = sum(forest.classes(multi))
    post = np.empty(S * p.width * p.height, np.float32)
    P = lib().orc_rf_frame(C.byref(p), forest.h, 1 if multi else 0, _p(rgb), _p(depth), _p(calib), _p(post))
    return post, P


def labels(values, Cn, mode, unknown=0):
    values = np.ascontiguousarray(values, np.float32).reshape(-1, Cn)
    out = np.empty(values.shape[0], np.int8)
    lib().orc_labels(_p(values), values.shape[0], Cn, mode, unknown, _p(out))
    return out


def fuse_posteriors(index_images, posteriors, class_counts, cloud_size):
    idx = np.ascontiguousarray(index_images, np.int32)
    n, H, W = idx.shape
    S = int(sum(class_counts))
    post = np.ascontiguousarray(posteriors, np.float32).reshape(n, S * H * W)
    cc = (C.c_int * len(class_counts))(*class_counts)
    out = np.empty(cloud_size * S, np.float32)
    lib().orc_fuse_posteriors(n, W, H, _p(idx), _p(post), len(class_counts), cc, cloud_size, _p(out))
    return out


class Lattice:
    """Permutohedral::init (SSE branch) + compute."""

    def __init__(self, feature):
        feature = np.ascontiguousarray(feature, np.float32)
        self.N, self.d = feature.shape
        self.h = lib().orc_lattice_init(_p(feature), self.N, self.d)
        self.M = self.h.contents.M
        n = self.N * (self.d + 1)
        self.offset = np.ctypeslib.as_array(self.h.contents.offset, (n,)).reshape(self.N, self.d + 1).copy()
        self.barycentric = np.ctypeslib.as_array(self.h.contents.barycentric, (n,)).reshape(self.N, self.d + 1).copy()
        if self.M > 0:
            self.keys = np.ctypeslib.as_array(self.h.contents.keys, (self.M * self.d,)).reshape(self.M, self.d).copy()
            self.blur_n1 = np.ctypeslib.as_array(self.h.contents.blur_n1, ((self.d + 1) * self.M,)).reshape(self.d + 1, self.M).copy()
            self.blur_n2 = np.ctypeslib.as_array(self.h.contents.blur_n2, ((self.d + 1) * self.M,)).reshape(self.d + 1, self.M).copy()

    def compute(self, values, reverse=False, which="auto"):
        values = np.ascontiguousarray(values, np.float32)
        out = np.empty_like(values)
        fn = {"auto": lib().orc_lattice_compute, "seq": lib().orc_lattice_compute_seq,
              "sse": lib().orc_lattice_compute_sse}[which]
        fn(self.h, _p(out), _p(values), values.shape[1], 1 if reverse else 0)
        return out

    def norm(self):
        out = np.empty(self.N, np.float32)
        lib().orc_kernel_norm(self.h, _p(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_lattice_free(self.h)
            self.h = None


def exp_and_normalize(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_exp_and_normalize(_p(out), _p(x), x.shape[0], x.shape[1])
    return out


def crf_inference(unary_energy, feature, w, iters):
    U = np.ascontiguousarray(unary_energy, np.float32)
    F = np.ascontiguousarray(feature, np.float32)
    N, Cn = U.shape
    Q = np.empty_like(U)
    lib().orc_crf_inference(N, Cn, F.shape[1], _p(U), _p(F), C.c_float(w), iters, _p(Q))
    return Q


def crf_inference_multi(unary_energy, features, ws, iters):
    U = np.ascontiguousarray(unary_energy, np.float32)
    N, Cn = U.shape
    feats = [np.ascontiguousarray(f, np.float32) for f in features]
    ds = (C.c_int * len(feats))(*[f.shape[1] for f in feats])
    ptrs = (C.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
    wsa = (C.c_float * len(feats))(*ws)
    Q = np.empty_like(U)
    lib().orc_crf_inference_multi(N, Cn, len(feats), ds, ptrs, wsa, _p(U), iters, _p(Q))
    return Q


def frame_crf_features(p, rgb, cl):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    cl = np.ascontiguousarray(cl, np.float32)
    out = np.empty((p.width * p.height, 6), np.float32)
    lib().orc_frame_crf_features(C.byref(p), _p(rgb), _p(cl), _p(out))
    return out


def segment_frame(p, forest, multi, rgb, depth, calib, label_mode=3, unknown=None):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    calib = np.ascontiguousarray(calib, np.float32)
    cc = forest.classes(multi)
    S = sum(cc)
    N = p.width * p.height
    post = np.empty(S * N, np.float32)
    marg = np.empty(S * N, np.float32)
    lab = np.empty(len(cc) * N, np.int8)
    unk = None
    if unknown is not None:
        unk = np.asarray(unknown, np.int32)
    lib().orc_segment_frame(C.byref(p), forest.h, 1 if multi else 0, _p(rgb), _p(depth), _p(calib),
                            _p(post), _p(marg), _p(lab), label_mode, _p(unk))
    return post, marg, lab
