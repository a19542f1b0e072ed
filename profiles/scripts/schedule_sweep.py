"""Step time of the frame path under both splat schedules for several chunk shapes (run on the GPU box):
flat / deep synthetic scenes x frames per chunk x {list-major, resident}.  Prints one JSON line per case and writes
gpurun_out/schedule_sweep.json.  The rule in rvseg_crf.hip (resident_pays) is derived from this table."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rovinasemanticsegmentation_amd as rv
from rovinasemanticsegmentation_amd import synthetic

W, H = 640, 480
N = W * H
dev = torch.device("cuda", 0)
blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30, single_classes=9, layer_classes=(8, 9))
calib = synthetic.make_calib(W, H)
out = []
frames_list = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,16,32,64".split(","))]
for scene in ("flat", "deep"):
    rgb, depth = synthetic.make_batch(max(frames_list), W, H, holes=True, scene=scene)
    d_rgb_all = torch.from_numpy(rgb).to(dev)
    d_depth_all = torch.from_numpy(depth.view(np.int16)).to(dev)
    for n in frames_list:
        d_marg = torch.empty((n, 9 * N), dtype=torch.float32, device=dev)
        d_lab = torch.empty((n, N), dtype=torch.int8, device=dev)
        for splat in (1, 2):
            ctx = rv.Context(multi_layer=0, use_dense_crf=1, dcrf_iterations=5, label_mode=1, unknown_label=[8], max_batch=n,
                             lattice_capacity_log2=13 if scene == "deep" else 12, schedule=dict(splat=splat))
            ctx.forest_load(blob)
            s = torch.cuda.current_stream(dev).cuda_stream
            def step():
                ctx.segment_frames_device(n, d_rgb_all.data_ptr(), d_depth_all.data_ptr(), calib, 0, d_marg.data_ptr(), d_lab.data_ptr(), s)
            for _ in range(3):
                step()
            torch.cuda.synchronize(dev)
            ctx.poll_status(True)
            t0 = time.perf_counter()
            reps = 8
            for _ in range(reps):
                step()
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / reps * 1e3
            ctx.poll_status(True)
            info = ctx.last_schedule()
            st = ctx.last_timing()
            rec = {"scene": scene, "frames": n, "want": "list-major" if splat == 1 else "resident", "ran": info["splat"],
                   "planner_fallback": info["planner_fallback"], "ms_per_step": round(ms, 3), "mpix_s": round(n * N / ms / 1e3, 1),
                   "vertices_per_frame": info["vertices"] // n, "longest_list": info["longest_list"],
                   "splat_ms": round(st.get("splat", 0.0), 3), "mf_update_ms": round(st.get("mf_update", 0.0), 3),
                   "lattice_build_ms": round(st.get("lattice_build", 0.0), 3)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            ctx.close()
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "schedule_sweep.json"), "w"), indent=1)
