"""BASELINE.json configs[2] -- the headline configuration bench.py times -- checked for correctness on
the GPU: one chunk of 64 synthetic 640x480 key frames (holes on) through rvseg_segment_frames_device
with max_batch = 64, RF (bench forest: 4 trees x 2^14 leaves, C = 9) + 5-iteration DenseCRF.

  (a) every frame bit-exact against the CPU oracle (labels, and marginals well inside the north star's
      1e-4) -- frames 0, 9, 31, 63 sit in different XCD launch groups of the splat (kernels_crf.hip),
      all 64 are compared; the oracle frames run in a thread pool (ctypes releases the GIL);
  (b) all 64 frames against single-frame GPU runs of the same context parameters.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic

pytestmark = pytest.mark.gpu

W, H, N_FRAMES, C = 640, 480, 64, 9
TOL = 1e-4   # BASELINE.json north_star: CRF class marginals within 1e-4, argmax labels bit-exact


def test_batch64_headline_config_matches_oracle_and_single_frame_runs(gpu_ctx_factory, oracle):
    torch = pytest.importorskip("torch")
    import rovinasemanticsegmentation_amd as rv
    dev = torch.device("cuda", 0)
    N = W * H
    blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30,
                                       single_classes=C, layer_classes=(8, 9))
    rgb, depth = synthetic.make_batch(N_FRAMES, W, H, holes=True)
    calib = synthetic.make_calib(W, H)
    kw = dict(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF, unknown_label=[8])
    ctx = gpu_ctx_factory(max_batch=N_FRAMES, **kw)
    ctx.forest_load(blob)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.zeros((N_FRAMES, C * N), dtype=torch.float32, device=dev)
    d_lab = torch.full((N_FRAMES, N), -99, dtype=torch.int8, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.segment_frames_device(N_FRAMES, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(),
                              d_lab.data_ptr(), stream.cuda_stream)
    assert ctx.poll_status(wait=True) == rv.capi.OK
    torch.cuda.synchronize(dev)
    marg = d_marg.cpu().numpy()
    lab = d_lab.cpu().numpy()
    assert (lab != -99).all()

    # (a) the CPU oracle, frame by frame
    forest = oracle.Forest(blob)
    p = oracle.default_params(dcrf_iterations=5)

    def one(i):
        return oracle.segment_frame(p, forest, 0, rgb[i], depth[i], calib, label_mode=1, unknown=[8])

    with ThreadPoolExecutor(max(1, min(os.cpu_count() or 1, 16))) as ex:
        want = list(ex.map(one, range(N_FRAMES)))
    for i in [0, 9, 31, 63] + [k for k in range(N_FRAMES) if k not in (0, 9, 31, 63)]:
        _, wm, wl = want[i]
        assert np.array_equal(lab[i], wl), "labels of frame %d differ from the oracle" % i
        assert np.abs(marg[i] - wm).max() <= TOL, i
        assert np.array_equal(marg[i], wm), "marginals of frame %d are not bit-identical to the oracle" % i

    # (b) the same frames one at a time on the GPU (chunk of 1: other launch shapes, no XCD groups; the long lists of
    # such a launch are summed by scan blocks, kernels_crf.hip: splat_scan_item), and again with the serial adder only
    one_ctx = gpu_ctx_factory(max_batch=1, **kw)
    one_ctx.forest_load(blob)
    for i in range(N_FRAMES):
        out = one_ctx.segment_frames(rgb[i:i + 1], depth[i:i + 1], calib, want_posteriors=False)
        assert np.array_equal(out["labels"][0].ravel(), lab[i]), i
        assert np.array_equal(out["marginals"][0], marg[i]), i
    assert one_ctx.last_schedule()["splat"] == "list-major"
    serial_ctx = gpu_ctx_factory(max_batch=1, schedule=dict(serial_chains=1), **kw)
    serial_ctx.forest_load(blob)
    for i in (0, 9, 63):
        out = serial_ctx.segment_frames(rgb[i:i + 1], depth[i:i + 1], calib, want_posteriors=False)
        assert np.array_equal(out["marginals"][0], marg[i]), i
    # (c) chunks of 3 and 4 frames: scan blocks with several frames per launch (frame order of the items)
    for n, first in ((3, 5), (4, 40)):
        few_ctx = gpu_ctx_factory(max_batch=n, **kw)
        few_ctx.forest_load(blob)
        out = few_ctx.segment_frames(rgb[first:first + n], depth[first:first + n], calib, want_posteriors=False)
        assert np.array_equal(out["labels"].reshape(n, -1), lab[first:first + n]), n
        assert np.array_equal(out["marginals"].reshape(n, -1), marg[first:first + n]), n


def test_deep_scene_many_vertices_default_schedule(gpu_ctx_factory, oracle):
    """A scene with a 1-10 m depth range and textured colour: ~2 300 lattice vertices per frame instead of the ~350 of
    the flat scene every other test uses (the regime of a real photo).  8 frames of 640x480 through the host entry with
    default parameters: the default hash capacity (2^12 slots per frame) overflows once, is raised and the chunk
    repeated; the schedule is the library's own choice (no forcing) and is reported; every frame bit-exact against the
    oracle."""
    import rovinasemanticsegmentation_amd as rv
    n = 8
    blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 12, max_depth=20, single_classes=C, layer_classes=(8, 9))
    rgb, depth = synthetic.make_batch(n, W, H, holes=True, scene="deep")
    calib = synthetic.make_calib(W, H)
    kw = dict(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF, unknown_label=[8])
    ctx = gpu_ctx_factory(max_batch=n, **kw)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib, want_posteriors=False)
    info = ctx.last_schedule()
    assert info["vertices"] >= n * 1500 and info["capacity_log2"] >= 13, info      # many vertices; the capacity was raised
    assert info["splat"] == "list-major" and info["planner_fallback"] == 0 and info["csr_path"] == 1, info
    forest = oracle.Forest(blob)
    p = oracle.default_params(dcrf_iterations=5)

    def one(i):
        return oracle.segment_frame(p, forest, 0, rgb[i], depth[i], calib, label_mode=1, unknown=[8])

    with ThreadPoolExecutor(max(1, min(os.cpu_count() or 1, 16))) as ex:
        want = list(ex.map(one, range(n)))
    for i in range(n):
        _, wm, wl = want[i]
        assert np.array_equal(out["labels"][i].ravel(), wl), i
        assert np.abs(out["marginals"][i] - wm).max() <= TOL, i
        assert np.array_equal(out["marginals"][i], wm), i
    # the same frames under the resident band schedule (forced): its planner handles > 2 048 vertices per frame
    ctx2 = gpu_ctx_factory(max_batch=4, lattice_capacity_log2=13, schedule=dict(splat=2, resident_blocks=4), **kw)
    ctx2.forest_load(blob)
    out2 = ctx2.segment_frames(rgb[:4], depth[:4], calib, want_posteriors=False)
    info2 = ctx2.last_schedule()
    assert info2["splat"] == "resident" and info2["planner_fallback"] == 0, info2
    for i in range(4):
        assert np.array_equal(out2["marginals"][i], out["marginals"][i]), i
        assert np.array_equal(out2["labels"][i], out["labels"][i]), i
