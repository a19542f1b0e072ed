"""Extra measurements that bench.py appends to its JSON line (rank 0, N = 1, outside the timed
region).  Each one is a bounded run of a caller of the hot path or of a configuration BASELINE.json
names next to the headline one; none of them is the reported `value`.

  host      rvseg_segment_frames, the host-buffer entry point that INTEGRATION.md puts into
            src/segmenter.cpp:351-434: numpy (pageable) inputs, outputs back in host memory, PCIe included
  localmap  the consumer of the per-frame output: local-map fusion + cloud CRF (src/segmenter.cpp:561-658)
            with device-resident buffers
  config5   BASELINE configs[4] on ONE GPU: 1280x960, dual-layer forest, 10 CRF iterations
"""
import time

import numpy as np

W, H = 640, 480


def _host_path(ctx_factory, blob, rgb, depth, calib):
    """64 frames per call through the host entry point; labels only, and labels + marginals."""
    import rovinasemanticsegmentation_amd as rv
    n = rgb.shape[0]
    ctx = ctx_factory(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF,
                      unknown_label=[8], max_batch=n, lattice_capacity_log2=12)
    out = {}
    try:
        ctx.forest_load(blob)
        for key, kw in (("labels_only", dict(want_posteriors=False, want_marginals=False)),
                        ("labels_and_marginals", dict(want_posteriors=False, want_marginals=True))):
            ctx.segment_frames(rgb, depth, calib, **kw)   # allocations, page-in
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.segment_frames(rgb, depth, calib, **kw)
            dt = (time.perf_counter() - t0) / reps
            out[key] = {"mpix_s": round(n * W * H / dt / 1e6, 1), "ms_per_call": round(dt * 1e3, 2), "frames_per_call": n}
        # the same call with page-locked caller buffers (rvseg_host_register): the copy engines write the caller's
        # memory directly, no staging copy on the host
        bufs = ctx.host_buffers(n, want_posteriors=False, want_marginals=True)
        try:
            bufs["rgb"][...] = rgb
            bufs["depth"][...] = depth
            ctx.segment_frames(bufs["rgb"], bufs["depth"], calib, out=bufs)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.segment_frames(bufs["rgb"], bufs["depth"], calib, out=bufs)
            dt = (time.perf_counter() - t0) / reps
            out["labels_and_marginals_pinned"] = {"mpix_s": round(n * W * H / dt / 1e6, 1), "ms_per_call": round(dt * 1e3, 2), "frames_per_call": n}
        finally:
            ctx.release_host_buffers(bufs)
    finally:
        ctx.close()
    out["note"] = ("rvseg_segment_frames; pageable numpy buffers: host copies into a pinned ring, H2D / D2H of neighbouring chunks under the "
                   "compute; `_pinned`: caller buffers registered with rvseg_host_register, DMA straight into them")
    return out


def _local_map(ctx_factory, dev, blob, rgb, depth, calib):
    """The consumer of the hot path's output, device resident: 32 key frames -> RF log-posteriors (dual layer, the
    node's configuration: no per-frame CRF) -> fusion into a ~1.2 M-point cloud through index images -> cloud
    DenseCRF (10 iterations, config.json:85) per layer -> labels.  Nothing crosses PCIe between the stages."""
    import torch
    import rovinasemanticsegmentation_amd as rv
    from . import synthetic
    n = min(32, rgb.shape[0])
    N = W * H
    # cloud: frames 0, 8, 16, 24 back-projected at full resolution, each from its own (drifting) camera position
    shifts = [np.array([0.01 * i, -0.02 * i, 0.005 * i]) for i in range(n)]
    xyz, col = [], []
    for i in range(0, n, 8):
        pts, valid = synthetic.back_project(depth[i], calib, W, H)
        xyz.append((pts[valid] + shifts[i]).astype(np.float32))
        col.append(rgb[i].reshape(-1, 3)[valid].astype(np.float32) / np.float32(255.0))
    xyz, col = np.concatenate(xyz), np.concatenate(col)
    idx = np.stack([synthetic.project_cloud(xyz, calib, W, H, shift=tuple(shifts[i])) for i in range(n)])
    P = xyz.shape[0]
    frames = ctx_factory(multi_layer=1, use_dense_crf=0, max_batch=n)
    cmap = ctx_factory(multi_layer=1, use_dense_crf=1, dcrf_iterations=10, unknown_label=[7, 8])
    try:
        frames.forest_load(blob)
        cmap.forest_load(blob)
        cc = frames.forest_info()["class_counts"]
        S = sum(cc)
        d_rgb = torch.from_numpy(rgb[:n]).to(dev)
        d_depth = torch.from_numpy(depth[:n].view(np.int16)).to(dev)
        d_post = torch.empty((n, S * N), dtype=torch.float32, device=dev)
        d_idx = torch.from_numpy(idx).to(dev)
        d_xyz = torch.from_numpy(xyz).to(dev)
        d_col = torch.from_numpy(col).to(dev)
        d_lab = torch.empty((len(cc), P), dtype=torch.int8, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream

        def run_frames():
            frames.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, d_post.data_ptr(), 0, 0, s)

        def run_map():
            cmap.process_map_device(n, d_idx.data_ptr(), d_post.data_ptr(), P, d_xyz.data_ptr(), d_col.data_ptr(), d_lab.data_ptr(), 0, s)

        out = {}
        for name, fn in (("frames_rf_ms", run_frames), ("process_map_ms", run_map)):
            fn()
            torch.cuda.synchronize(dev)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            out[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        cmap.poll_status(wait=True)
        st = cmap.last_timing()
        out["fusion_ms"] = round(st.get("fusion", 0.0), 3)
        out["cloud_crf_ms"] = round(sum(v for k, v in st.items() if k not in ("fusion", "cloud_features", "labels")), 3)
        out["stage_ms"] = {k: round(v, 3) for k, v in st.items()}
        seen = int((idx >= 0).sum())
        out.update({"frames": n, "cloud_points": int(P), "index_hits": seen, "layers": cc, "crf_iterations": 10,
                    "labelled_unknown_frac": [round(float((d_lab[l] == u).float().mean().item()), 4) for l, u in enumerate((7, 8))],
                    "note": "device resident: rvseg_segment_frames_device -> rvseg_process_map_device (src/segmenter.cpp:561-658)"})
        return out
    finally:
        frames.close()
        cmap.close()


def _config5(ctx_factory, dev, blob, rgb, depth, calib):
    """BASELINE configs[4] on one GPU: 1280x960 RGB-D, dual-layer (material + object) forest, 10 CRF iterations.
    16 frames per step (the pixel count of the headline's 64 x 640x480).  4 687 algorithmic B/px."""
    import torch
    from . import synthetic
    W5, H5, n, iters = 1280, 960, 16, 10
    N = W5 * H5
    rgb5, depth5 = synthetic.make_batch(n, W5, H5, holes=False)
    calib5 = synthetic.make_calib(W5, H5)
    ctx = ctx_factory(width=W5, height=H5, multi_layer=1, use_dense_crf=1, dcrf_iterations=iters, label_mode=1,
                      unknown_label=[7, 8], max_batch=n, lattice_capacity_log2=12)
    try:
        ctx.forest_load(blob)
        cc = ctx.forest_info()["class_counts"]
        S = sum(cc)
        d_rgb = torch.from_numpy(rgb5).to(dev)
        d_depth = torch.from_numpy(depth5.view(np.int16)).to(dev)
        d_marg = torch.empty((n, S * N), dtype=torch.float32, device=dev)
        d_lab = torch.empty((n, len(cc), N), dtype=torch.int8, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream

        def step():
            ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib5, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)

        step()
        ctx.poll_status(wait=True)
        torch.cuda.synchronize(dev)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        ctx.poll_status(wait=True)
        mpix = n * N / dt / 1e6
        bpp = 4687.0
        return {"mpix_s": round(mpix, 1), "ms_per_step": round(dt * 1e3, 3), "frames_per_step": n, "width": W5, "height": H5,
                "layers": cc, "crf_iterations": iters,
                "roofline": {"bound": "hbm", "algorithmic_bytes_per_px": bpp, "achieved": round(bpp * mpix * 1e6 / 1e9, 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(bpp * mpix * 1e6 / 8e12, 4), "scope": "whole path (all kernels of a step)"},
                "stage_ms_last_step": {k: round(v, 3) for k, v in ctx.last_timing().items()}}
    finally:
        ctx.close()


def _train(ctx_factory, dev, blob, rgb, depth, calib):
    """Forest training on the GPU straight from frames (rvseg_forest_train_frames): the reference's learner settings
    (4 trees, depth 30, min_split 50, bootstrap, sqrt(D) features per node; src/train.cpp:225-239) on 8 bench frames
    with label images; features are extracted and packed on the device (no P x D matrix, no 0.9 GB upload)."""
    ctx = ctx_factory()
    try:
        n = 8
        H, W = rgb.shape[1], rgb.shape[2]
        yy, xx = np.mgrid[0:H, 0:W]
        lab = np.empty((n, 2, H, W), np.int8)
        for i in range(n):
            # two label layers from image position + depth, like ground-truth label images
            lab[i, 0] = ((xx // 80) + 2 * (yy // 120)) % 8
            lab[i, 1] = ((xx // 64) + (depth[i] > 2500).astype(np.int64) * 3 + (yy // 160)) % 9
        t0 = time.perf_counter()
        model, n_ex = ctx.forest_train_frames(rgb[:n], depth[:n], calib, lab, [8, 9], augment=False, num_trees=4, max_depth=30,
                                              min_split_examples=50, seed=1)
        dt = time.perf_counter() - t0
        ctx.forest_load(model)
        info = ctx.forest_info()
        return {"seconds": round(dt, 3), "frames": n, "frames_per_s": round(n / dt, 1), "examples": int(n_ex), "features": ctx.feature_length,
                "layers": [8, 9], "trees": info["n_trees"], "nodes": info["n_nodes"], "max_depth": info["max_depth"], "model_bytes": len(model),
                "examples_per_s_per_tree": round(n_ex * 4 / dt),
                "note": "rvseg_forest_train_frames: extraction + exact split search (per-value histograms for the Lab bytes, a sort "
                        "per level for depth / height / normal) on the GPU; equals the oracle's depth-first learner byte for byte (tests)"}
    finally:
        ctx.close()


def _deep_scene(ctx_factory, dev, blob, rgb, depth, calib):
    """The headline shape (64 frames of 640x480, RF + 5-iteration CRF) on a scene with a 1-10 m depth range and textured
    colour: ~2 300 lattice vertices per frame instead of ~350.  Reports the rate, which schedule the library chose by
    itself and whether the splat planner had to fall back."""
    import torch
    from . import synthetic
    n, W, H = 64, rgb.shape[2], rgb.shape[1]
    N = W * H
    rgb_d, depth_d = synthetic.make_batch(n, W, H, holes=True, scene="deep")
    ctx = ctx_factory(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=1, unknown_label=[8], max_batch=n)
    try:
        ctx.forest_load(blob)
        d_rgb = torch.from_numpy(rgb_d).to(dev)
        d_depth = torch.from_numpy(depth_d.view(np.int16)).to(dev)
        d_marg = torch.empty((n, 9 * N), dtype=torch.float32, device=dev)
        d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        overflows = 0
        for _ in range(6):   # warm-up: the default capacity overflows once on this scene, the context raises it
            ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
            try:
                ctx.poll_status(True)
            except Exception:
                overflows += 1
        torch.cuda.synchronize(dev)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / reps * 1e3
        ctx.poll_status(True)
        info = ctx.last_schedule()
        return {"mpix_s": round(n * N / ms / 1e3, 1), "ms_per_step": round(ms, 3), "frames_per_step": n,
                "vertices_per_frame": info["vertices"] // n, "longest_list": info["longest_list"], "schedule": info["splat"],
                "planner_fallback": info["planner_fallback"], "capacity_log2": info["capacity_log2"], "overflow_retries": overflows,
                "stage_ms_last_step": {k: round(v, 3) for k, v in ctx.last_timing().items()}}
    finally:
        ctx.close()


def run(ctx, dev, blob, rgb_h, depth_h, calib, want=None):
    import rovinasemanticsegmentation_amd as rv

    def factory(**kw):
        return rv.Context(device=dev.index or 0, **kw)

    out = {}
    if want is None or "host" in want:
        try:
            r = _host_path(factory, blob, rgb_h, depth_h, calib)
            out["host_path"] = r
            out["host_path_mpix_s"] = {k: v["mpix_s"] for k, v in r.items() if isinstance(v, dict) and "mpix_s" in v}
        except Exception as e:  # an extra must never take the headline down with it
            out["host_path"] = {"error": repr(e)}
    keys = {"localmap": "local_map", "config5": "config5_1gpu", "train": "forest_train", "deep": "deep_scene"}
    for name, fn in (("localmap", globals().get("_local_map")), ("config5", globals().get("_config5")), ("train", globals().get("_train")),
                     ("deep", globals().get("_deep_scene"))):
        if fn is None or not (want is None or name in want):
            continue
        try:
            out[keys[name]] = fn(factory, dev, blob, rgb_h, depth_h, calib)
        except Exception as e:
            out[keys[name]] = {"error": repr(e)}
    return out
