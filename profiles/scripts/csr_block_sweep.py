import sys, time, json, numpy as np, torch
sys.path.insert(0, '.')
import rovinasemanticsegmentation_amd as rv
from rovinasemanticsegmentation_amd import synthetic
dev = torch.device('cuda', 0)
W, H, C = 640, 480, 9
N = W * H
blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=C, layer_classes=(8, 9))
calib = synthetic.make_calib(W, H)
for scene, n, cap in (("flat", 64, 12), ("deep", 64, 13), ("flat", 32, 12), ("flat", 16, 12), ("flat", 8, 12)):
    rgb, depth = synthetic.make_batch(n, W, H, holes=True, scene=scene)
    d_rgb = torch.from_numpy(rgb).to(dev); d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.empty((n, C * N), dtype=torch.float32, device=dev); d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
    ref = None
    for cs in (256, 512, 1024, 2048, 4096):
        ctx = rv.Context(width=W, height=H, multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF, unknown_label=[8], max_batch=n, lattice_capacity_log2=cap, schedule=dict(csr_block=cs))
        ctx.forest_load(blob)
        s = torch.cuda.current_stream(dev).cuda_stream
        def step():
            ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
        for _ in range(3): step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(6): step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 6 * 1e3
        lab = d_lab.cpu().numpy()
        if ref is None: ref = lab
        print(scene, n, "cs", cs, round(ms, 3), "ms", ctx.last_schedule()["splat"], "build", round(ctx.last_timing().get("lattice_build", 0), 2), "same", bool((lab == ref).all()), flush=True)
        ctx.close()
