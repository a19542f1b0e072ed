"""ctypes binding of librvseg.so (include/rvseg.h).

The HIP library is the product: there is no CPU fallback.  Importing this module without a built
librvseg.so raises; creating a context without a GPU raises RvsegError(NO_DEVICE).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RVSEG_LIBRARY: another build of the library (A/B timing of kernel variants on one GPU box); the product default is the
# in-tree librvseg.so
LIB_PATH = os.environ.get("RVSEG_LIBRARY") or os.path.join(_HERE, "librvseg.so")

RVSEG_MAX_LAYERS = 8

OK, ERR_INVALID_ARG, ERR_IO, ERR_FORMAT, ERR_NO_FOREST, ERR_HIP, ERR_NO_DEVICE, ERR_CAPACITY, NOT_READY = range(9)
LABEL_EVAL, LABEL_CRF, LABEL_NOCRF, LABEL_ARGMAX = range(4)

# every symbol include/rvseg.h declares
SYMBOLS = [
    "rvseg_params_default", "rvseg_create", "rvseg_destroy", "rvseg_last_error",
    "rvseg_status_string", "rvseg_feature_length", "rvseg_forest_load", "rvseg_forest_load_mem",
    "rvseg_forest_info", "rvseg_forest_eval", "rvseg_extract_features", "rvseg_segment_frames",
    "rvseg_segment_frames_device", "rvseg_crf_infer", "rvseg_crf_infer_multi",
    "rvseg_lattice_build", "rvseg_lattice_filter", "rvseg_lattice_neighbours", "rvseg_last_timing",
    "rvseg_fuse_posteriors", "rvseg_label_values",
    "rvseg_forest_check", "rvseg_forest_write", "rvseg_forest_write_mem", "rvseg_forest_rewrite",
    "rvseg_poll_status",
    "rvseg_fuse_posteriors_device", "rvseg_cloud_features_device", "rvseg_crf_infer_device",
    "rvseg_label_values_device", "rvseg_process_map_device",
    "rvseg_crf_features_gaussian", "rvseg_crf_features_bilateral",
    "rvseg_train_params_default", "rvseg_forest_train",
    "rvseg_comm_unique_id", "rvseg_comm_init", "rvseg_comm_destroy", "rvseg_gather_frames",
    "rvseg_schedule_default", "rvseg_set_schedule", "rvseg_last_schedule",
    "rvseg_forest_train_result", "rvseg_forest_train_frames",
    "rvseg_host_register", "rvseg_host_unregister",
]


class RvsegParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
        ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("patch_size", C.c_int32), ("patch_size_reduce", C.c_int32),
        ("feature_color_patch", C.c_int32), ("feature_depth", C.c_int32),
        ("feature_height", C.c_int32), ("feature_normal", C.c_int32),
        ("fill_value", C.c_float),
        ("use_dense_crf", C.c_int32),
        ("dcrf_xyz_kernel", C.c_float), ("dcrf_rgb_kernel", C.c_float),
        ("dcrf_kernel_weight", C.c_float), ("dcrf_iterations", C.c_int32),
        ("multi_layer", C.c_int32), ("label_mode", C.c_int32),
        ("unknown_label", C.c_int32 * RVSEG_MAX_LAYERS),
        ("max_batch", C.c_int32), ("device", C.c_int32), ("lattice_capacity_log2", C.c_int32),
    ]


class RvsegTrainParams(C.Structure):
    _fields_ = [
        ("num_trees", C.c_int32), ("max_depth", C.c_int32), ("min_split_examples", C.c_int32),
        ("min_child_split_examples", C.c_int32), ("num_features", C.c_int32), ("use_bootstrap", C.c_int32),
        ("smoothing", C.c_float), ("seed", C.c_uint64),
    ]


class RvsegSchedule(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "splat", "resident_blocks", "resident_band", "resident_chunk", "resident_window", "resident_cap_tiles",
        "group_vertices", "overlap_build", "overlap_layers", "build_priority_high", "trace", "serial_chains", "csr_block")]


class RvsegScheduleInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "splat", "planner_fallback", "csr_path", "n_frames", "points_per_frame", "vertices", "longest_list", "resident_blocks",
        "resident_band", "resident_chunk", "capacity_log2")]


SPLAT_NAMES = {0: "none", 1: "list-major", 2: "resident"}


class RvsegError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rvseg status %d: %s" % (status, message))
        self.status = status


_lib = None


def lib():
    """Loads librvseg.so.  torch (if it is going to be used in this process) must be imported
    first so that both share one HIP runtime (same soname libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librvseg.so is not built (run `python __graft_entry__.py` or "
            "`make -C rovinasemanticsegmentation_amd/csrc`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    PP = C.POINTER(RvsegParams)
    L.rvseg_params_default.argtypes = [PP]
    L.rvseg_params_default.restype = None
    L.rvseg_create.argtypes = [PP, C.POINTER(vp)]
    L.rvseg_destroy.argtypes = [vp]
    L.rvseg_destroy.restype = None
    L.rvseg_last_error.argtypes = [vp]
    L.rvseg_last_error.restype = C.c_char_p
    L.rvseg_status_string.argtypes = [C.c_int]
    L.rvseg_status_string.restype = C.c_char_p
    L.rvseg_feature_length.argtypes = [vp]
    L.rvseg_forest_load.argtypes = [vp, C.c_char_p]
    L.rvseg_forest_load_mem.argtypes = [vp, vp, C.c_size_t]
    L.rvseg_forest_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                    C.POINTER(i32 * RVSEG_MAX_LAYERS)]
    L.rvseg_forest_eval.argtypes = [vp, vp, i32, i32, vp]
    L.rvseg_extract_features.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.rvseg_segment_frames.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.rvseg_segment_frames_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.rvseg_crf_infer.argtypes = [vp, i32, i32, i32, vp, vp, f32, i32, vp, vp, i32, i32]
    L.rvseg_crf_infer_multi.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32]
    L.rvseg_fuse_posteriors.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp]
    L.rvseg_label_values.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.rvseg_lattice_build.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, C.POINTER(i32)]
    L.rvseg_lattice_filter.argtypes = [vp, vp, i32, vp]
    L.rvseg_lattice_neighbours.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rvseg_last_timing.argtypes = [vp, C.c_char_p, C.c_size_t, vp, i32]
    L.rvseg_forest_check.argtypes = [vp, C.c_size_t, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
    L.rvseg_forest_write.argtypes = [vp, C.c_char_p]
    L.rvseg_forest_write_mem.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rvseg_forest_rewrite.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rvseg_poll_status.argtypes = [vp, i32]
    L.rvseg_fuse_posteriors_device.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp]
    L.rvseg_cloud_features_device.argtypes = [vp, i32, vp, vp, vp, vp]
    L.rvseg_crf_infer_device.argtypes = [vp, i32, i32, i32, vp, i32, vp, f32, i32, vp, vp, i32, i32, vp]
    L.rvseg_label_values_device.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.rvseg_process_map_device.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.rvseg_train_params_default.argtypes = [C.POINTER(RvsegTrainParams)]
    L.rvseg_train_params_default.restype = None
    L.rvseg_forest_train.argtypes = [vp, vp, i32, i32, vp, i32, vp, C.POINTER(RvsegTrainParams), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rvseg_forest_train_result.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rvseg_forest_train_frames.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, i32, C.POINTER(RvsegTrainParams), vp, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.POINTER(i32)]
    L.rvseg_host_register.argtypes = [vp, C.c_size_t]
    L.rvseg_host_unregister.argtypes = [vp]
    L.rvseg_comm_unique_id.argtypes = [vp]
    L.rvseg_comm_init.argtypes = [vp, i32, i32, vp]
    L.rvseg_comm_destroy.argtypes = [vp]
    L.rvseg_comm_destroy.restype = None
    L.rvseg_gather_frames.argtypes = [vp, vp, C.c_size_t, vp, i32, vp]
    L.rvseg_crf_features_gaussian.argtypes = [i32, i32, f32, f32, vp]
    L.rvseg_crf_features_bilateral.argtypes = [i32, i32, f32, f32, f32, f32, f32, vp, vp]
    L.rvseg_schedule_default.argtypes = [C.POINTER(RvsegSchedule)]
    L.rvseg_schedule_default.restype = None
    L.rvseg_set_schedule.argtypes = [vp, C.POINTER(RvsegSchedule)]
    L.rvseg_last_schedule.argtypes = [vp, C.POINTER(RvsegScheduleInfo)]
    for name in SYMBOLS:
        getattr(L, name)  # raises AttributeError if the library does not export it
    _lib = L
    return L


def default_params(**kw):
    p = RvsegParams()
    lib().rvseg_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "unknown_label":
            for i, u in enumerate(v):
                p.unknown_label[i] = int(u)
        else:
            setattr(p, k, v)
    return p


def check(ctx, status):
    if status != OK:
        msg = lib().rvseg_last_error(ctx)
        raise RvsegError(status, (msg or b"").decode("utf-8", "replace") or
                         lib().rvseg_status_string(status).decode())


def forest_check(blob, feature_length=0):
    """Host-only validation (no GPU): returns (status, message, info)."""
    blob = bytes(blob)
    nt, nn, md = C.c_int32(), C.c_int32(), C.c_int32()
    err = C.create_string_buffer(512)
    st = lib().rvseg_forest_check(blob, len(blob), feature_length, C.byref(nt), C.byref(nn), C.byref(md), err, 512)
    return st, err.value.decode("utf-8", "replace"), {"n_trees": nt.value, "n_nodes": nn.value, "max_depth": md.value}


def forest_rewrite(blob):
    """Host-only: parse a forest.dat image and serialise it again with the library's writer."""
    blob = bytes(blob)
    size = C.c_size_t()
    st = lib().rvseg_forest_rewrite(blob, len(blob), None, 0, C.byref(size))
    if st != OK:
        raise RvsegError(st, lib().rvseg_status_string(st).decode())
    out = C.create_string_buffer(size.value)
    st = lib().rvseg_forest_rewrite(blob, len(blob), out, size.value, C.byref(size))
    if st != OK:
        raise RvsegError(st, lib().rvseg_status_string(st).decode())
    return out.raw[:size.value]


def crf_features_gaussian(W, H, sx, sy):
    """DenseCRF2D::addPairwiseGaussian's feature matrix (densecrf.cpp:61-69), W*H x 2, host only."""
    import numpy as np
    out = np.empty((W * H, 2), np.float32)
    st = lib().rvseg_crf_features_gaussian(W, H, sx, sy, out.ctypes.data_as(C.c_void_p))
    if st != OK:
        raise RvsegError(st, lib().rvseg_status_string(st).decode())
    return out


def crf_features_bilateral(W, H, sx, sy, sr, sg, sb, im):
    """DenseCRF2D::addPairwiseBilateral's feature matrix (densecrf.cpp:70-81), W*H x 5, host only."""
    import numpy as np
    im = np.ascontiguousarray(im, np.uint8)
    assert im.size == W * H * 3
    out = np.empty((W * H, 5), np.float32)
    st = lib().rvseg_crf_features_bilateral(W, H, sx, sy, sr, sg, sb, im.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if st != OK:
        raise RvsegError(st, lib().rvseg_status_string(st).decode())
    return out
