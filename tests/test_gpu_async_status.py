"""The asynchronous surface of the frame path: hash-table overflow is never silent (host entry redoes
the chunk, the device entry reports it through rvseg_poll_status and raises the capacity), per-call
calibrations survive back-to-back calls without a synchronisation in between, and the model limits of
the device evaluator are enforced by the loader."""
import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic

pytestmark = pytest.mark.gpu

W, H = 160, 120


def _case(n, seed=41):
    blob = synthetic.make_forest_bytes(seed=seed, n_trees=4, leaves_per_tree=256, max_depth=12)
    rgb, depth = synthetic.make_batch(n, W, H, holes=True)
    return blob, rgb, depth, synthetic.make_calib(W, H)


def test_host_entry_recovers_from_hash_overflow(gpu_ctx_factory, oracle):
    """2^4 slots per frame cannot hold a frame's lattice (~150 vertices at this size): the host entry must
    notice the overflow after the chunk, raise the capacity and run the chunk again -- also for the LAST
    chunk of a call, which nobody would look at again."""
    blob, rgb, depth, calib = _case(3)
    forest = oracle.Forest(blob)
    kw = dict(width=W, height=H, use_dense_crf=1, dcrf_iterations=3, label_mode=1, max_batch=2, lattice_capacity_log2=4)
    ctx = gpu_ctx_factory(**kw)
    ctx.forest_load(blob)
    out = ctx.segment_frames(rgb, depth, calib)
    p = oracle.default_params(width=W, height=H, dcrf_iterations=3)
    for i in range(3):
        post, marg, lab = oracle.segment_frame(p, forest, 1, rgb[i], depth[i], calib, label_mode=1, unknown=[7, 8])
        assert np.array_equal(out["posteriors"][i], post), i
        assert np.array_equal(out["marginals"][i], marg), i
        assert np.array_equal(out["labels"][i].ravel(), lab), i
    # the context keeps the raised capacity: a second call needs no retry and still agrees
    out2 = ctx.segment_frames(rgb[:1], depth[:1], calib)
    assert np.array_equal(out2["marginals"][0], out["marginals"][0])


def test_device_entry_reports_overflow_through_poll_status(gpu_ctx_factory, oracle):
    torch = pytest.importorskip("torch")
    import rovinasemanticsegmentation_amd as rv
    dev = torch.device("cuda", 0)
    blob, rgb, depth, calib = _case(2)
    forest = oracle.Forest(blob)
    N = W * H
    kw = dict(width=W, height=H, use_dense_crf=1, dcrf_iterations=2, label_mode=1, max_batch=2, lattice_capacity_log2=4)
    ctx = gpu_ctx_factory(**kw)
    ctx.forest_load(blob)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.zeros((2, 17 * N), dtype=torch.float32, device=dev)
    d_lab = torch.zeros((2, 2, N), dtype=torch.int8, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream

    def call():
        ctx.segment_frames_device(2, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)

    call()
    seen = 0
    for _ in range(8):   # 2^4 -> 2^7 -> 2^10 ...: each report raises the capacity eightfold
        try:
            st = ctx.poll_status(wait=True)
            assert st == rv.capi.OK
            break
        except rv.capi.RvsegError as e:
            assert e.status == rv.capi.ERR_CAPACITY
            seen += 1
            call()
    assert seen >= 1, "2^4 slots per frame cannot have been enough"
    assert ctx.poll_status(wait=False) == rv.capi.OK      # nothing pending any more
    torch.cuda.synchronize(dev)
    p = oracle.default_params(width=W, height=H, dcrf_iterations=2)
    marg = d_marg.cpu().numpy()
    lab = d_lab.cpu().numpy()
    for i in range(2):
        _, wm, wl = oracle.segment_frame(p, forest, 1, rgb[i], depth[i], calib, label_mode=1, unknown=[7, 8])
        assert np.array_equal(marg[i], wm), i
        assert np.array_equal(lab[i].ravel(), wl), i


def test_pending_overflow_status_survives_a_cloud_crf_on_the_same_context(gpu_ctx_factory):
    """a frame build that overflowed and whose status nobody has polled yet must not be lost when
    a cloud CRF (a synchronous lattice build on the SAME context) runs next -- the two read their counters back into
    separate pinned slots.  Sequence: overflowed segment_frames_device -> crf_infer_device on a small cloud (clean)
    -> poll_status still reports ERR_CAPACITY for the frame call."""
    torch = pytest.importorskip("torch")
    import rovinasemanticsegmentation_amd as rv
    dev = torch.device("cuda", 0)
    blob, rgb, depth, calib = _case(2)
    N = W * H
    ctx = gpu_ctx_factory(width=W, height=H, use_dense_crf=1, dcrf_iterations=2, label_mode=1, max_batch=2, lattice_capacity_log2=4)
    ctx.forest_load(blob)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.zeros((2, 17 * N), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    ctx.segment_frames_device(2, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), 0, s)
    # a clean cloud CRF on the same context (its own lattice, its own overflow retry with the worst-case capacity)
    rng = np.random.default_rng(5)
    Np, Cc = 3000, 9
    U = torch.from_numpy((rng.random((Np, Cc)) * 3).astype(np.float32)).to(dev)
    F = torch.from_numpy((rng.random((Np, 6)) * 4).astype(np.float32)).to(dev)
    Q = torch.zeros((Np, Cc), dtype=torch.float32, device=dev)
    ctx.crf_infer_device(Np, Cc, 6, U.data_ptr(), True, F.data_ptr(), 3.0, 2, Q.data_ptr(), 0, stream=s)
    torch.cuda.synchronize(dev)
    assert abs(float(Q.sum(1).mean()) - 1.0) < 1e-4
    with pytest.raises(rv.capi.RvsegError) as e:
        ctx.poll_status(wait=True)
    assert e.value.status == rv.capi.ERR_CAPACITY
    assert ctx.poll_status(wait=True) == rv.capi.OK     # consumed; the capacity has been raised


def test_back_to_back_device_calls_keep_their_own_calibration(gpu_ctx_factory, oracle):
    """The device entry returns without synchronising; its pinned calibration staging must not be
    overwritten by the next call before the copy has run.  Two calls with different
    extrinsics, no synchronisation in between, each checked against the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    blob, rgb, depth, calib = _case(2, seed=43)
    forest = oracle.Forest(blob)
    N = W * H
    calib2 = calib.copy()
    c, s_ = np.cos(0.3), np.sin(0.3)
    R = calib[9:18].reshape(3, 3).astype(np.float64)
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
    calib2[9:18] = (Rz @ R).astype(np.float32).ravel()
    calib2[18:21] = np.array([0.4, -0.2, 1.3], np.float32)
    ctx = gpu_ctx_factory(width=W, height=H, use_dense_crf=1, dcrf_iterations=2, label_mode=1, max_batch=2)
    ctx.forest_load(blob)
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    outs = [(torch.zeros((2, 17 * N), dtype=torch.float32, device=dev), torch.zeros((2, 17 * N), dtype=torch.float32, device=dev))
            for _ in range(6)]
    s = torch.cuda.current_stream(dev).cuda_stream
    calibs = [calib, calib2, calib, calib2, calib2, calib]     # more calls than staging slots
    for (d_post, d_marg), cal in zip(outs, calibs):
        ctx.segment_frames_device(2, d_rgb.data_ptr(), d_depth.data_ptr(), cal, d_post.data_ptr(), d_marg.data_ptr(), 0, s)
    torch.cuda.synchronize(dev)
    p = oracle.default_params(width=W, height=H, dcrf_iterations=2)
    want = {}
    for k, cal in ((0, calib), (1, calib2)):
        want[k] = [oracle.segment_frame(p, forest, 1, rgb[i], depth[i], cal, label_mode=1, unknown=[7, 8]) for i in range(2)]
    assert not np.array_equal(want[0][0][1], want[1][0][1]), "the two calibrations must give different marginals"
    for (d_post, d_marg), cal in zip(outs, calibs):
        k = 0 if cal is calib else 1
        post, marg = d_post.cpu().numpy(), d_marg.cpu().numpy()
        for i in range(2):
            assert np.array_equal(post[i], want[k][i][0])
            assert np.array_equal(marg[i], want[k][i][1])


def test_loader_refuses_more_than_64_trees_and_writes_models_back(gpu_ctx_factory, golden_dir):
    import os
    import rovinasemanticsegmentation_amd as rv
    ctx = gpu_ctx_factory()
    big = synthetic.make_forest_bytes(seed=3, n_trees=65, leaves_per_tree=4, max_depth=4)
    with pytest.raises(rv.capi.RvsegError) as ei:
        ctx.forest_load(big)
    assert ei.value.status == rv.capi.ERR_CAPACITY
    # RandomForest::write: a reference-written file comes back byte for byte through a context
    blob = open(os.path.join(golden_dir, "forest_multi.dat"), "rb").read()
    ctx.forest_load(blob)
    assert rv.RandomForest(ctx).write() == blob
