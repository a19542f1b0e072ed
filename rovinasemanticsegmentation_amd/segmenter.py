"""Host-side Python mirror of the reference's call shapes for the hot path, over the C ABI.

Names follow the reference so that the parity tests read like its call sites:

  RandomForest.read / classLogPosterior / multiClassLogPosterior   (libforest classifiers.h:274-344)
  FeatureExtractor.extract                                         (include/feature_extractor.h:41)
  DenseCRF(N, C).setUnaryEnergy / addPairwiseEnergy / inference / map  (densecrf.h:36-121)
  Segmenter.processFrames                                          (src/segmenter.cpp:323-443)

All compute happens in librvseg.so (HIP, gfx950).  Nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """Owns one rvseg_ctx (one per thread and device, as include/rvseg.h requires)."""

    def __init__(self, schedule=None, **params):
        """params: fields of rvseg_params; schedule: dict of rvseg_schedule fields (launch-schedule overrides for
        tests, profiling and tuning -- the library reads no environment variables)."""
        self.params = capi.default_params(**params)
        h = C.c_void_p()
        st = capi.lib().rvseg_create(C.byref(self.params), C.byref(h))
        if st != capi.OK:
            raise capi.RvsegError(st, capi.lib().rvseg_last_error(None).decode("utf-8", "replace"))
        self.h = h
        self.L = capi.lib()
        if schedule:
            self.set_schedule(**schedule)

    def set_schedule(self, **kw):
        """rvseg_set_schedule: the defaults with the given rvseg_schedule fields replaced."""
        sc = capi.RvsegSchedule()
        self.L.rvseg_schedule_default(C.byref(sc))
        known = {f[0] for f in capi.RvsegSchedule._fields_}
        for k, v in kw.items():
            if k not in known:
                raise TypeError("unknown rvseg_schedule field %r" % k)
            setattr(sc, k, int(v))
        capi.check(self.h, self.L.rvseg_set_schedule(self.h, C.byref(sc)))

    def last_schedule(self):
        """rvseg_last_schedule as a dict; `splat` as a name ("list-major" / "resident").  `vertices` and
        `planner_fallback` are -1 until poll_status(wait=True) (or a host entry point) has returned."""
        info = capi.RvsegScheduleInfo()
        capi.check(self.h, self.L.rvseg_last_schedule(self.h, C.byref(info)))
        d = {f[0]: getattr(info, f[0]) for f in capi.RvsegScheduleInfo._fields_}
        d["splat"] = capi.SPLAT_NAMES.get(d["splat"], str(d["splat"]))
        return d

    def close(self):
        if getattr(self, "h", None):
            self.L.rvseg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- forest ------------------------------------------------------------------------------
    @property
    def feature_length(self):
        return self.L.rvseg_feature_length(self.h)

    def forest_load(self, src):
        if isinstance(src, (bytes, bytearray, memoryview)):
            buf = bytes(src)
            capi.check(self.h, self.L.rvseg_forest_load_mem(self.h, buf, len(buf)))
        else:
            capi.check(self.h, self.L.rvseg_forest_load(self.h, str(src).encode()))

    def forest_write(self, path=None):
        """RandomForest::write (classifier.cpp:210-220): to `path`, or returns the bytes."""
        if path is not None:
            capi.check(self.h, self.L.rvseg_forest_write(self.h, str(path).encode()))
            return None
        size = C.c_size_t()
        capi.check(self.h, self.L.rvseg_forest_write_mem(self.h, None, 0, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        capi.check(self.h, self.L.rvseg_forest_write_mem(self.h, buf, size.value, C.byref(size)))
        return buf.raw[:size.value]

    def forest_train(self, X, labels, class_counts, **train_params):
        """rvseg_forest_train: X (P, D) float32, labels (P, L) int32 class indices.  Returns the forest.dat bytes.
        train_params: fields of rvseg_train_params (num_trees, max_depth, min_split_examples, ...)."""
        X = np.ascontiguousarray(X, np.float32)
        labels = np.ascontiguousarray(np.asarray(labels, np.int32).reshape(X.shape[0], -1))
        P, D = X.shape
        L = labels.shape[1]
        assert len(class_counts) == L
        tp = self._train_params(train_params)
        cc = (C.c_int32 * L)(*class_counts)
        size = C.c_size_t()
        # trains once (the model stays on the context), then fetches it into a buffer of the right size
        capi.check(self.h, self.L.rvseg_forest_train(self.h, _ptr(X), P, D, _ptr(labels), L, cc, C.byref(tp), None, 0, C.byref(size)))
        return self._trained_model(size.value)

    def _train_params(self, train_params):
        tp = capi.RvsegTrainParams()
        self.L.rvseg_train_params_default(C.byref(tp))
        known = {f[0] for f in capi.RvsegTrainParams._fields_}
        for k, v in train_params.items():
            if k not in known:
                raise TypeError("unknown rvseg_train_params field %r" % k)
            setattr(tp, k, v)
        return tp

    def _trained_model(self, size):
        buf = C.create_string_buffer(size)
        got = C.c_size_t()
        capi.check(self.h, self.L.rvseg_forest_train_result(self.h, buf, size, C.byref(got)))
        return buf.raw[:got.value]

    def forest_train_frames(self, rgb, depth, calib, labels, class_counts, augment=False, **train_params):
        """rvseg_forest_train_frames: rgb (n, H, W, 3) uint8, depth (n, H, W) uint16, labels (n, L, H, W) int8 (< 0 =
        unlabelled).  Returns (forest.dat bytes, number of training points)."""
        p = self.params
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        labels = np.ascontiguousarray(labels, np.int8)
        n = rgb.shape[0]
        L = labels.shape[1]
        assert rgb.shape == (n, p.height, p.width, 3) and depth.shape == (n, p.height, p.width)
        assert labels.shape == (n, L, p.height, p.width) and len(class_counts) == L
        calib = np.ascontiguousarray(np.broadcast_to(np.asarray(calib, np.float32).reshape(-1, 21), (n, 21)))
        tp = self._train_params(train_params)
        cc = (C.c_int32 * L)(*class_counts)
        size = C.c_size_t()
        n_ex = C.c_int32()
        capi.check(self.h, self.L.rvseg_forest_train_frames(self.h, n, _ptr(rgb), _ptr(depth), _ptr(calib), _ptr(labels), L, cc,
                                                            1 if augment else 0, C.byref(tp), None, 0, C.byref(size), C.byref(n_ex)))
        return self._trained_model(size.value), n_ex.value

    def poll_status(self, wait=True):
        """Status of the asynchronous part of the last segment_frames_device call: capi.OK,
        capi.NOT_READY (only with wait=False) or raises RvsegError(ERR_CAPACITY)."""
        st = self.L.rvseg_poll_status(self.h, 1 if wait else 0)
        if st == capi.NOT_READY:
            return st
        capi.check(self.h, st)
        return st

    def forest_info(self):
        nt, nn, md, nl = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        cc = (C.c_int32 * capi.RVSEG_MAX_LAYERS)()
        capi.check(self.h, self.L.rvseg_forest_info(self.h, C.byref(nt), C.byref(nn), C.byref(md), C.byref(nl), C.byref(cc)))
        return {"n_trees": nt.value, "n_nodes": nn.value, "max_depth": md.value,
                "class_counts": [cc[i] for i in range(nl.value)]}

    def forest_eval(self, X):
        X = np.ascontiguousarray(X, np.float32)
        P, D = X.shape
        S = sum(self.forest_info()["class_counts"])
        out = np.empty((P, S), np.float32)
        capi.check(self.h, self.L.rvseg_forest_eval(self.h, _ptr(X), P, D, _ptr(out)))
        return out

    # ---- features ----------------------------------------------------------------------------
    def extract_features(self, rgb, depth, calib):
        p = self.params
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        calib = np.ascontiguousarray(calib, np.float32)
        assert rgb.shape == (p.height, p.width, 3) and depth.shape == (p.height, p.width) and calib.size == 21
        cap = (p.height // p.stride + 1) * (p.width // p.stride + 1)
        D = self.feature_length
        feat = np.empty((cap, D), np.float32)
        xv = np.empty(cap, np.int32)
        yv = np.empty(cap, np.int32)
        n = C.c_int32()
        capi.check(self.h, self.L.rvseg_extract_features(self.h, _ptr(rgb), _ptr(depth), _ptr(calib), _ptr(feat), _ptr(xv), _ptr(yv), C.byref(n)))
        return feat[:n.value].copy(), xv[:n.value].copy(), yv[:n.value].copy()

    # ---- whole path --------------------------------------------------------------------------
    def host_buffers(self, n, want_posteriors=True, want_marginals=None, want_labels=True):
        """Page-locked (rvseg_host_register) in / out buffers for calls of n frames: pass them to segment_frames(out=...)
        and reuse them across calls; free with release_host_buffers."""
        p = self.params
        cc = self.forest_info()["class_counts"]
        S, Lc, N = sum(cc), len(cc), p.width * p.height
        if want_marginals is None:
            want_marginals = bool(p.use_dense_crf)
        bufs = {"rgb": np.empty((n, p.height, p.width, 3), np.uint8), "depth": np.empty((n, p.height, p.width), np.uint16),
                "posteriors": np.empty((n, S * N), np.float32) if want_posteriors else None,
                "marginals": np.empty((n, S * N), np.float32) if want_marginals else None,
                "labels": np.empty((n, Lc, p.height, p.width), np.int8) if want_labels else None, "class_counts": cc}
        for k in ("rgb", "depth", "posteriors", "marginals", "labels"):
            if bufs[k] is not None:
                st = self.L.rvseg_host_register(_ptr(bufs[k]), bufs[k].nbytes)
                if st != capi.OK:
                    raise capi.RvsegError(st, "rvseg_host_register failed")
        return bufs

    def release_host_buffers(self, bufs):
        for k in ("rgb", "depth", "posteriors", "marginals", "labels"):
            if bufs.get(k) is not None:
                self.L.rvseg_host_unregister(_ptr(bufs[k]))

    def segment_frames(self, rgb, depth, calib, want_posteriors=True, want_marginals=None, want_labels=True, out=None):
        """out: buffers from host_buffers() (page-locked): results land there without a staging copy; rgb / depth may be
        out["rgb"] / out["depth"] themselves."""
        p = self.params
        if out is not None:
            n = rgb.shape[0]
            calib = np.ascontiguousarray(np.broadcast_to(np.asarray(calib, np.float32).reshape(-1, 21), (n, 21)))
            capi.check(self.h, self.L.rvseg_segment_frames(self.h, n, _ptr(rgb), _ptr(depth), _ptr(calib), _ptr(out["posteriors"]),
                                                           _ptr(out["marginals"]), _ptr(out["labels"])))
            return out
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        n = rgb.shape[0]
        assert rgb.shape == (n, p.height, p.width, 3) and depth.shape == (n, p.height, p.width)
        calib = np.ascontiguousarray(np.broadcast_to(np.asarray(calib, np.float32).reshape(-1, 21), (n, 21)))
        cc = self.forest_info()["class_counts"]
        S, Lc, N = sum(cc), len(cc), p.width * p.height
        if want_marginals is None:
            want_marginals = bool(p.use_dense_crf)
        post = np.empty((n, S * N), np.float32) if want_posteriors else None
        marg = np.empty((n, S * N), np.float32) if want_marginals else None
        lab = np.empty((n, Lc, p.height, p.width), np.int8) if want_labels else None
        capi.check(self.h, self.L.rvseg_segment_frames(self.h, n, _ptr(rgb), _ptr(depth), _ptr(calib), _ptr(post), _ptr(marg), _ptr(lab)))
        return {"posteriors": post, "marginals": marg, "labels": lab, "class_counts": cc}

    def segment_frames_device(self, n, d_rgb, d_depth, calib, d_post=0, d_marg=0, d_labels=0, stream=0):
        """Device-pointer variant: integer addresses (e.g. torch tensor.data_ptr()) and a raw
        hipStream_t handle.  Enqueues only; the caller synchronises."""
        calib = np.ascontiguousarray(np.broadcast_to(np.asarray(calib, np.float32).reshape(-1, 21), (n, 21)))
        capi.check(self.h, self.L.rvseg_segment_frames_device(
            self.h, n, C.c_void_p(d_rgb), C.c_void_p(d_depth), _ptr(calib),
            C.c_void_p(d_post or None), C.c_void_p(d_marg or None), C.c_void_p(d_labels or None),
            C.c_void_p(stream or None)))

    # ---- CRF ---------------------------------------------------------------------------------
    def crf_infer(self, unary_energy, features, potts_w, iterations, label_mode=capi.LABEL_ARGMAX, unknown_label=0):
        U = np.ascontiguousarray(unary_energy, np.float32)
        F = np.ascontiguousarray(features, np.float32)
        N, Cn = U.shape
        assert F.shape[0] == N
        Q = np.empty_like(U)
        mp = np.empty(N, np.int8)
        capi.check(self.h, self.L.rvseg_crf_infer(self.h, N, Cn, F.shape[1], _ptr(U), _ptr(F), C.c_float(potts_w), iterations, _ptr(Q), _ptr(mp), label_mode, unknown_label))
        return Q, mp

    def crf_infer_multi(self, unary_energy, features, ws, iterations, label_mode=capi.LABEL_ARGMAX, unknown_label=0):
        U = np.ascontiguousarray(unary_energy, np.float32)
        N, Cn = U.shape
        feats = [np.ascontiguousarray(f, np.float32) for f in features]
        ds = (C.c_int32 * len(feats))(*[f.shape[1] for f in feats])
        ptrs = (C.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
        wsa = (C.c_float * len(feats))(*ws)
        Q = np.empty_like(U)
        mp = np.empty(N, np.int8)
        capi.check(self.h, self.L.rvseg_crf_infer_multi(self.h, N, Cn, len(feats), ds, ptrs, wsa, _ptr(U), iterations, _ptr(Q), _ptr(mp), label_mode, unknown_label))
        return Q, mp

    # ---- local-map fusion -------------------------------------------------------------------
    def fuse_posteriors(self, index_images, posteriors, class_counts, cloud_size):
        p = self.params
        idx = np.ascontiguousarray(index_images, np.int32)
        n = idx.shape[0]
        assert idx.shape == (n, p.height, p.width)
        S = int(sum(class_counts))
        post = np.ascontiguousarray(posteriors, np.float32).reshape(n, S * p.height * p.width)
        cc = (C.c_int32 * len(class_counts))(*class_counts)
        out = np.empty(cloud_size * S, np.float32)
        capi.check(self.h, self.L.rvseg_fuse_posteriors(self.h, n, _ptr(idx), _ptr(post), len(class_counts), cc, cloud_size, _ptr(out)))
        return out

    def process_map_device(self, n_images, d_index_images, d_posteriors, cloud_size, d_cloud_xyz, d_cloud_rgb, d_labels,
                           d_unaries=0, stream=0):
        """rvseg_process_map_device: integer device addresses, enqueues only."""
        capi.check(self.h, self.L.rvseg_process_map_device(
            self.h, n_images, C.c_void_p(d_index_images), C.c_void_p(d_posteriors), cloud_size,
            C.c_void_p(d_cloud_xyz or None), C.c_void_p(d_cloud_rgb or None), C.c_void_p(d_labels),
            C.c_void_p(d_unaries or None), C.c_void_p(stream or None)))

    def crf_infer_device(self, N, Cn, d, d_unary, unary_is_energy, d_features, potts_w, iterations, d_Q=0, d_map=0,
                         label_mode=capi.LABEL_ARGMAX, unknown_label=0, stream=0):
        capi.check(self.h, self.L.rvseg_crf_infer_device(
            self.h, N, Cn, d, C.c_void_p(d_unary), 1 if unary_is_energy else 0, C.c_void_p(d_features), C.c_float(potts_w),
            iterations, C.c_void_p(d_Q or None), C.c_void_p(d_map or None), label_mode, unknown_label, C.c_void_p(stream or None)))

    # ---- multi-GPU local-map gather over RCCL ------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        st = capi.lib().rvseg_comm_unique_id(buf)
        if st != capi.OK:
            raise capi.RvsegError(st, "rvseg_comm_unique_id failed (librccl.so missing?)")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        capi.check(self.h, self.L.rvseg_comm_init(self.h, rank, world, bytes(unique_id)))

    def gather_frames(self, d_local, bytes_per_rank, d_recv=0, root=0, stream=0):
        capi.check(self.h, self.L.rvseg_gather_frames(self.h, C.c_void_p(d_local), bytes_per_rank, C.c_void_p(d_recv or None), root,
                                                      C.c_void_p(stream or None)))

    def label_values(self, values, label_mode, unknown_label=0):
        V = np.ascontiguousarray(values, np.float32)
        N, Cn = V.shape
        out = np.empty(N, np.int8)
        capi.check(self.h, self.L.rvseg_label_values(self.h, _ptr(V), N, Cn, label_mode, unknown_label, _ptr(out)))
        return out

    def lattice_build(self, features, keys_capacity=None):
        F = np.ascontiguousarray(features, np.float32)
        N, d = F.shape
        cap = keys_capacity or N * (d + 1) + 8 * (d + 1)
        off = np.empty((N, d + 1), np.int32)
        bary = np.empty((N, d + 1), np.float32)
        keys = np.empty((cap, d), np.int16)
        M = C.c_int32()
        capi.check(self.h, self.L.rvseg_lattice_build(self.h, _ptr(F), N, d, _ptr(off), _ptr(bary), _ptr(keys), cap, C.byref(M)))
        return off, bary, keys[:M.value].copy(), M.value

    def lattice_neighbours(self, M, d, N):
        n1 = np.empty((d + 1, M), np.int32)
        n2 = np.empty((d + 1, M), np.int32)
        pix = np.empty(N * (d + 1), np.uint32)
        vs = np.empty(M, np.uint32)
        ve = np.empty(M, np.uint32)
        capi.check(self.h, self.L.rvseg_lattice_neighbours(self.h, _ptr(n1), _ptr(n2), _ptr(pix), _ptr(vs), _ptr(ve)))
        return n1, n2, pix, vs, ve

    def lattice_filter(self, values):
        V = np.ascontiguousarray(values, np.float32)
        out = np.empty_like(V)
        capi.check(self.h, self.L.rvseg_lattice_filter(self.h, _ptr(V), V.shape[1], _ptr(out)))
        return out

    def last_timing(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        n = self.L.rvseg_last_timing(self.h, names, 4096, ms, 64)
        ns = names.value.decode().split(";") if names.value else []
        return {ns[i]: ms[i] for i in range(min(n, len(ns)))}


# ---------------------------------------------------------------------------------------------
# reference-shaped facades
# ---------------------------------------------------------------------------------------------
class RandomForest:
    """libf::RandomForest as the hot path uses it (classifiers.h:274-344)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def read(self, src):  # RandomForest::read(std::istream&), classifier.cpp:222
        self.ctx.forest_load(src)

    def write(self, dst=None):  # RandomForest::write(std::ostream&), classifier.cpp:210
        return self.ctx.forest_write(dst)

    def getSize(self):
        return self.ctx.forest_info()["n_trees"]

    def classLogPosterior(self, X):  # classifier.cpp:166 -- ctx must have multi_layer=0
        return self.ctx.forest_eval(np.atleast_2d(X))

    def multiClassLogPosterior(self, X):  # classifier.cpp:187 -- ctx must have multi_layer=1
        out = self.ctx.forest_eval(np.atleast_2d(X))
        cc = self.ctx.forest_info()["class_counts"]
        offs = np.cumsum([0] + cc)
        return [out[:, offs[i]:offs[i + 1]] for i in range(len(cc))]


class FeatureExtractor:
    """Features::FeatureExtractor (include/feature_extractor.h:24-41), NO_LABEL extraction."""

    def __init__(self, ctx):
        self.ctx = ctx

    def extract(self, color, depth, calib):
        return self.ctx.extract_features(color, depth, calib)


class DenseCRF:
    """DenseCRF as Segmenter::processMapFromQueue drives it (src/segmenter.cpp:641-644)."""

    def __init__(self, ctx, N, M):
        self.ctx, self.N, self.M = ctx, N, M
        self.unary = None
        self.kernels = []

    def setUnaryEnergy(self, unary):  # densecrf.cpp:89-91; unary is N x M energy (= -log-posterior)
        unary = np.ascontiguousarray(unary, np.float32)
        assert unary.shape == (self.N, self.M)
        self.unary = unary

    def addPairwiseEnergy(self, features, potts_weight):  # densecrf.cpp:54-60 + PottsCompatibility
        features = np.ascontiguousarray(features, np.float32)
        assert features.shape[0] == self.N  # assert(features.cols() == N_), densecrf.cpp:55
        self.kernels.append((features, float(potts_weight)))

    def addPairwiseGaussian(self, W, H, sx, sy, w):  # densecrf.cpp:61-69 (features built by the C ABI)
        self.addPairwiseEnergy(capi.crf_features_gaussian(W, H, sx, sy), w)

    def addPairwiseBilateral(self, W, H, sx, sy, sr, sg, sb, im, w):  # densecrf.cpp:70-81
        self.addPairwiseEnergy(capi.crf_features_bilateral(W, H, sx, sy, sr, sg, sb, im), w)

    def inference(self, n_iterations, label_mode=capi.LABEL_ARGMAX, unknown_label=0):  # densecrf.cpp:115-131
        U = self.unary if self.unary is not None else np.zeros((self.N, self.M), np.float32)
        if len(self.kernels) == 1:
            f, w = self.kernels[0]
            return self.ctx.crf_infer(U, f, w, n_iterations, label_mode, unknown_label)
        return self.ctx.crf_infer_multi(U, [k[0] for k in self.kernels], [k[1] for k in self.kernels], n_iterations, label_mode, unknown_label)

    def map(self, n_iterations):  # densecrf.cpp:132-137
        return self.inference(n_iterations, capi.LABEL_ARGMAX)[1]


class LocalMapStore:
    """_cloud_results plus the two services that read it (src/segmenter.cpp:711-774): (map id,
    result_labels[layer]) pairs in arrival order.  Host only."""

    def __init__(self, layer_names):
        self.layer_names = list(layer_names)
        self.results = []

    def store(self, local_map_id, result_labels):  # :711-713
        self.results.append((int(local_map_id), [np.asarray(l, np.uint8).copy() for l in result_labels]))

    def srvStoredSemanticsIds(self):  # :722-729 -> IdsSrv response `int32[] local_map_ids`
        return [m[0] for m in self.results]

    def srvGetLocalMapSegmentation(self, local_map_id, segmentation_layers):
        """:731-774 -> (local_map_id, uint8[] point_labels = requested layers concatenated) or False for an
        unknown layer name (:744-746) or map id (:773)."""
        idx = []
        for name in segmentation_layers:
            if name in self.layer_names:
                idx.append(self.layer_names.index(name))
        if len(idx) != len(segmentation_layers):
            return False
        for mid, labels in self.results:
            if mid == local_map_id:
                parts = [labels[l] for l in idx]
                return mid, (np.concatenate(parts) if parts else np.zeros(0, np.uint8))
        return False


class Segmenter:
    """The per-frame inference part of `class Segmenter` (include/segmenter.h:47-69), ROS-free:
    construction loads the forest like Segmenter::Segmenter (src/segmenter.cpp:106-129),
    processFrames does what processFramesFromQueueInternalRF does per dequeued frame
    (src/segmenter.cpp:351-431) for a whole batch, plus the per-frame DenseCRF when enabled;
    processMap is processMapFromQueue for one local map; the srv* methods are the three services
    (:722-792) over plain Python values.

    layers (optional): [{"name": str, "classes": [(class name, (r, g, b)), ...]}, ...] -- the
    `color_codings` of config.json (:73-98); needed only by the services and the cloud dumps."""

    def __init__(self, forest, layers=None, **params):
        self.ctx = Context(**params)
        self.ctx.forest_load(forest)
        info = self.ctx.forest_info()
        self.layer_class_counts = info["class_counts"]
        self.layer_count = len(self.layer_class_counts)
        if layers is None:
            layers = [{"name": "layer%d" % l, "classes": [("class%d" % c, (0, 0, 0)) for c in range(n)]}
                      for l, n in enumerate(self.layer_class_counts)]
        if [len(l["classes"]) for l in layers] != list(self.layer_class_counts):
            self.ctx.close()
            raise RuntimeError("model / config mismatch: layer or class counts (README.md:30 of the reference)")
        self.layers = layers
        self.store = LocalMapStore([l["name"] for l in layers])

    def processFrames(self, color, depth, calib, **kw):
        return self.ctx.segment_frames(color, depth, calib, **kw)

    def processMap(self, index_images, posteriors, cloud_xyz, cloud_rgb, unknown_labels=None, local_map_id=None):
        """The body of processMapFromQueue for one local map (src/segmenter.cpp:561-682): fuse the frames'
        label distributions into per-point unaries through the index images, then per layer either the
        cloud DenseCRF with the thresholded argmax (:628-658) or the no-CRF rule (:660-681).
        cloud_rgb is in [0, 1] like fps_mapper's cloud (:698-700).  With local_map_id the labels are kept
        for the services (:711-713).  Returns (result_labels, unaries)."""
        p = self.ctx.params
        cc = self.layer_class_counts
        cloud_xyz = np.ascontiguousarray(cloud_xyz, np.float32)
        cloud_rgb = np.ascontiguousarray(cloud_rgb, np.float32)
        n_pts = cloud_xyz.shape[0]
        flat = self.ctx.fuse_posteriors(index_images, posteriors, cc, n_pts)
        offs = np.cumsum([0] + [c * n_pts for c in cc])
        unaries = [flat[offs[l]:offs[l + 1]].reshape(n_pts, cc[l]) for l in range(len(cc))]
        unknown = unknown_labels if unknown_labels is not None else [p.unknown_label[l] for l in range(len(cc))]
        labels = []
        if p.use_dense_crf:
            pairwise = np.concatenate([cloud_xyz * np.float32(p.dcrf_xyz_kernel), cloud_rgb * np.float32(p.dcrf_rgb_kernel)], 1)   # :629-637
            for l in range(len(cc)):
                _, mp = self.ctx.crf_infer(-unaries[l], pairwise, p.dcrf_kernel_weight, p.dcrf_iterations, capi.LABEL_CRF, unknown[l])
                labels.append(mp.astype(np.uint8))
        else:
            for l in range(len(cc)):
                labels.append(self.ctx.label_values(unaries[l], capi.LABEL_NOCRF, unknown[l]).astype(np.uint8))
        if local_map_id is not None:
            self.store.store(local_map_id, labels)
        return labels, unaries

    # ---- services (ROS request / response fields as plain values) -----------------------------------
    def srvStoredSemanticsIds(self):
        return self.store.srvStoredSemanticsIds()

    def srvGetLocalMapSegmentation(self, local_map_id, segmentation_layers):
        return self.store.srvGetLocalMapSegmentation(local_map_id, segmentation_layers)

    def srvSegmentationInformation(self):  # :776-791
        return {"layer_names": [l["name"] for l in self.layers],
                "class_counts": [len(l["classes"]) for l in self.layers],
                "class_names": [c[0] for l in self.layers for c in l["classes"]],
                "class_colors": [int(v) for l in self.layers for c in l["classes"] for v in c[1]]}

    def close(self):
        self.ctx.close()
