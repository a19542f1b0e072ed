"""BASELINE.json configs[3] -- "local map of 256 key frames sharded across 8 x MI355X, per-frame RF+CRF then
RCCL label gather" -- as far as ONE GPU can exercise it: the per-rank shard of that configuration.

Rank 3 of 8 owns frames [96, 128) of the 256-frame map (`shard_frames`).  Its chunk of 32 synthetic 640x480 key
frames (holes on, bench forest) goes through rvseg_segment_frames_device with max_batch = 32 -- 9.8 M points -- under
BOTH splat schedules: the library's own choice for this shape (the resident bands, by the measured rule of
rvseg_crf.hip: resident_pays) and the list-major walk, which neither the 64-frame nor the single-frame tests run at
this size -- and then through the C-ABI RCCL gather (rvseg_comm_init / rvseg_gather_frames, world size 1 on this box:
the 8-GPU run is the driver's).

  (a) every frame's labels and marginals bit-exact against the CPU oracle (thread pool);
  (b) the gathered block equals the labels the chunk wrote.
The frame order of the 8-rank gather itself is covered on CPU by tests/test_distributed_gloo.py (world 8, 256 frames).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from rovinasemanticsegmentation_amd import synthetic
from rovinasemanticsegmentation_amd.distributed import shard_frames

pytestmark = pytest.mark.gpu

W, H, C = 640, 480, 9
MAP_FRAMES, WORLD, RANK = 256, 8, 3
TOL = 1e-4   # BASELINE.json north_star: CRF class marginals within 1e-4, argmax labels bit-exact


_ORACLE = {}


@pytest.mark.parametrize("splat,ran", [(0, "resident"), (1, "list-major")])
def test_config4_rank_shard_matches_oracle_and_gathers(gpu_ctx_factory, oracle, splat, ran):
    torch = pytest.importorskip("torch")
    import rovinasemanticsegmentation_amd as rv
    dev = torch.device("cuda", 0)
    N = W * H
    start, n = shard_frames(MAP_FRAMES, RANK, WORLD)
    assert (start, n) == (96, 32)
    blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30,
                                       single_classes=C, layer_classes=(8, 9))
    rgb, depth = synthetic.make_batch(n, W, H, holes=True, start=start)
    calib = synthetic.make_calib(W, H)
    ctx = gpu_ctx_factory(max_batch=n, multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF,
                          unknown_label=[8], schedule=dict(splat=splat))
    ctx.forest_load(blob)
    ctx.comm_init(0, 1, rv.Context.comm_unique_id())
    d_rgb = torch.from_numpy(rgb).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.zeros((n, C * N), dtype=torch.float32, device=dev)
    d_lab = torch.full((n, N), -99, dtype=torch.int8, device=dev)
    d_fused = torch.full((n, N), -98, dtype=torch.int8, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(),
                              stream.cuda_stream)
    ctx.gather_frames(d_lab.data_ptr(), d_lab.numel(), d_fused.data_ptr(), 0, stream.cuda_stream)
    assert ctx.poll_status(wait=True) == rv.capi.OK
    torch.cuda.synchronize(dev)
    sched = ctx.last_schedule()
    assert sched["splat"] == ran and sched["planner_fallback"] == 0 and sched["n_frames"] == n, sched
    marg = d_marg.cpu().numpy()
    lab = d_lab.cpu().numpy()
    assert (lab != -99).all()

    forest = oracle.Forest(blob)
    p = oracle.default_params(dcrf_iterations=5)

    def one(i):
        return oracle.segment_frame(p, forest, 0, rgb[i], depth[i], calib, label_mode=1, unknown=[8])

    if "want" not in _ORACLE:   # the two schedules are checked against the same oracle frames
        with ThreadPoolExecutor(max(1, min(os.cpu_count() or 1, 16))) as ex:
            _ORACLE["want"] = list(ex.map(one, range(n)))
    want = _ORACLE["want"]
    for i in range(n):
        _, wm, wl = want[i]
        assert np.array_equal(lab[i], wl), "labels of map frame %d differ from the oracle" % (start + i)
        assert np.abs(marg[i] - wm).max() <= TOL, i
        assert np.array_equal(marg[i], wm), "marginals of map frame %d are not bit-identical to the oracle" % (start + i)
    # (b) the fusion rank's receive buffer (world size 1: this rank's block) holds exactly these labels
    assert np.array_equal(d_fused.cpu().numpy(), lab)
