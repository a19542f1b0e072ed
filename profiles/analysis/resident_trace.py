"""Per-block timing of the resident band splat (rvseg_schedule.trace = 1): run on the GPU box.
Prints, for the last splat launch of a 64-frame chunk: span of the launch, and per block of one frame its tiles,
run time, time per tile and time spent waiting for the pace."""
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rovinasemanticsegmentation_amd as rv
from rovinasemanticsegmentation_amd import synthetic
W, H, N, n = 640, 480, 640 * 480, 64
dev = torch.device("cuda", 0)
blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=9, layer_classes=(8, 9))
rgb, depth = synthetic.make_batch(n); calib = synthetic.make_calib()
d_rgb = torch.from_numpy(rgb).to(dev); d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
d_marg = torch.empty((n, 9 * N), dtype=torch.float32, device=dev); d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
ctx = rv.Context(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=1, unknown_label=[8], max_batch=n, lattice_capacity_log2=12, schedule=dict(trace=1))
ctx.forest_load(blob)
s = torch.cuda.current_stream(dev).cuda_stream
for _ in range(2):
    ctx.segment_frames_device(n, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
torch.cuda.synchronize()
L = ctx.L
L.rvseg_debug_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
MAXB = 16
tr = np.zeros((n, MAXB, 8), np.uint64); t0s = np.zeros((n, MAXB + 1), np.uint32); meta = (C.c_int * 6)()
st = L.rvseg_debug_resident(ctx.h, tr.ctypes.data_as(C.c_void_p), tr.nbytes, t0s.ctypes.data_as(C.c_void_p), t0s.nbytes, meta)
B = meta[0]
print("status", st, "B", B, "band_wb", meta[1], "n_bands", meta[2], "frames", meta[3], "valid", meta[5])
tr = tr[:, :B]
start = tr[:, :, 0].astype(np.int64); end = tr[:, :, 1].astype(np.int64)
tiles = (tr[:, :, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64); nown = (tr[:, :, 2] >> np.uint64(32)).astype(np.int64)
spin = tr[:, :, 3].astype(np.int64)
t0 = start.min()
print("launch span %.1f us; block starts within %.1f us" % ((end.max() - t0) / 100.0, (start.max() - t0) / 100.0))
print("tiles per frame: mean %.0f; per block min %d max %d" % (tiles.sum(1).mean(), tiles.min(), tiles.max()))
for f in (0, 9):
    print("frame", f)
    for j in range(B):
        run = (end[f, j] - start[f, j]) / 100.0
        print("  block %2d groups %2d tiles %5d  run %7.1f us  %.3f us/tile  paced wait %6.1f us  ends at %.1f" %
              (j, nown[f, j], tiles[f, j], run, (run - spin[f, j] / 100.0) / max(1, tiles[f, j]), spin[f, j] / 100.0, (end[f, j] - t0) / 100.0))
fe = (end.max(1) - t0) / 100.0
print("frame end times us: min %.0f median %.0f max %.0f; slowest frames %s; by XCD (f %% 8) %s" % (fe.min(), np.median(fe), fe.max(), np.argsort(-fe)[:6].tolist(), [round(float(fe[x::8].mean())) for x in range(8)]))
run = (end - start) / 100.0
print("all blocks: us/tile (without waits) mean %.3f; heavy blocks (j=0) %.3f; others %.3f" %
      (((run - spin / 100.0).sum() / tiles.sum()), ((run - spin / 100.0)[:, 0].sum() / tiles[:, 0].sum()), ((run - spin / 100.0)[:, 1:].sum() / tiles[:, 1:].sum())))
clk = tr[:, :, 4].astype(np.float64)
print("shader clock: %.0f MHz (s_memtime ticks per elapsed us)" % (clk.sum() / np.maximum(run, 1e-9).sum()))
print("waiting for the pace: %.1f%% of block time" % (100.0 * spin.sum() / 100.0 / run.sum()))

