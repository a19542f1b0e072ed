import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import rovinasemanticsegmentation_amd as rv
from rovinasemanticsegmentation_amd import synthetic
dev = torch.device('cuda', 0)
W, H, C = 640, 480, 9
N = W * H
blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=C, layer_classes=(8, 9))
calib = synthetic.make_calib(W, H)
for n in (1, 2, 4, 6, 8, 12, 16):
    rgb, depth = synthetic.make_batch(n, W, H, holes=True)
    d_rgb = torch.from_numpy(rgb).to(dev); d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    d_marg = torch.empty((n, C * N), dtype=torch.float32, device=dev); d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
    ctx = rv.Context(width=W, height=H, multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=rv.capi.LABEL_CRF, unknown_label=[8], max_batch=n)
    ctx.forest_load(blob)
    s = torch.cuda.current_stream(dev).cuda_stream
    def step():
        ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
    for _ in range(3): step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 8 * 1e3
    t = ctx.last_timing()
    print("frames", n, round(ms, 3), "ms", ctx.last_schedule()["splat"], "splat", round(t.get("splat", 0), 2), "build", round(t.get("lattice_build", 0), 2), "checksum", int(d_lab.to(torch.int64).sum().item()), flush=True)
    ctx.close()
