"""Driver for the two latency shapes of bench.py (configs[1]: one 640x480 frame; the local map's cloud CRF), meant to
run under `rocprofv3 --kernel-trace` (profiles/scripts/latency_trace.sh), which turns the trace into a per-kernel
timeline of the LAST repetition: start offset, duration, queue -- so gaps between launches and what the two streams
overlap are visible.  Usage: python3 profiles/scripts/latency_shapes.py {frame|cloud|deep|batch} [reps]
(deep / batch: 64 frames of the deep / the flat scene, for a per-kernel timeline of the whole step)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import rovinasemanticsegmentation_amd as rv
    from rovinasemanticsegmentation_amd import synthetic, bench_extras
    what = sys.argv[1] if len(sys.argv) > 1 else "frame"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    W, H, C = 640, 480, 9
    N = W * H
    blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=C,
                                       layer_classes=(8, 9))
    calib = synthetic.make_calib(W, H)
    if what in ("frame", "deep", "batch"):
        n = 1 if what == "frame" else 64
        rgb, depth = synthetic.make_batch(n, W, H, holes=True, scene="deep" if what == "deep" else "flat")
        ctx = rv.Context(width=W, height=H, multi_layer=0, use_dense_crf=1, dcrf_iterations=5,
                         label_mode=rv.capi.LABEL_CRF, unknown_label=[8], max_batch=n,
                         lattice_capacity_log2=13 if what == "deep" else 12)
        ctx.forest_load(blob)
        d_rgb = torch.from_numpy(rgb).to(dev)
        d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
        d_marg = torch.empty((n, C * N), dtype=torch.float32, device=dev)
        d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream

        def step():
            ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
            torch.cuda.synchronize(dev)
        for _ in range(3):
            step()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        print(what + ": %.3f ms per call" % ((time.perf_counter() - t0) / reps * 1e3))
        print(ctx.last_timing())
    else:
        rgb, depth = synthetic.make_batch(32, W, H, holes=True)

        def factory(**kw):
            p = dict(width=W, height=H)
            p.update(kw)
            return rv.Context(**p)
        out = bench_extras._local_map(factory, dev, blob, rgb, depth, calib)
        print(out)


if __name__ == "__main__":
    main()
