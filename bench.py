#!/usr/bin/env python3
"""bench.py -- Mpix/s of the RF + 5-iteration DenseCRF hot path at 640x480 RGB-D on MI355X.

A "step" is one pass of the hot path over one batch of 64 synthetic 640x480 key frames
(BASELINE.json configs[2], the HBM-roofline run; configs[1] -- a single frame -- is the same call
with n = 1 and is reported as `latency_ms_single_frame`).  Inputs are resident in HBM before the
timed region; every step writes CRF marginals and labels to HBM.  The frames have invalid-depth holes
(like every parity test) and the steps alternate between two different input sets.

`--config 3` times BASELINE.json configs[3] instead: every rank owns 32 key frames of the local map
(256 frames over 8 GPUs) and every step ends with the label gather to the fusion rank (RCCL; at N = 1
through the C ABI's own communicator of world size 1).

Multi-GPU: one process per GPU, frames shard across ranks with no data-path
collective except the final local-map label gather to rank 0 over RCCL, i.e. weak scaling.  Two ways
in: the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or plain `python bench.py --gpus N`, which starts the N
rank processes itself -- before anything in this process touches the GPU -- and fails loudly when the
node has fewer than N devices.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
FRAMES_PER_STEP = 64
CRF_ITERS = 5
C_CLASSES = 9
D_FEAT = 6
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MARGINAL_TOL = 1e-4    # BASELINE.json north_star

# Algorithmic HBM bytes per full-resolution pixel and LAUNCH GROUP (SURVEY.md 8d, C = 9, d = 6).
# `launches` = how many times the stage runs per step.
STAGE_BYTES_PER_PX = {
    "prep": 5.0,                      # 3 B rgb + 2 B depth in
    "window_map": 0.0, "normal_feature": 0.0,
    "rf_frames": 0.0,                 # reads cached forest / Lab; its output is the low-res image
    "upsample_pack": 4.0 * C_CLASSES,  # posteriors out
    "lattice_build": 8.0 * (D_FEAT + 1) + 8.0 * (D_FEAT + 1) + 4,  # coefficients once + normaliser pass
    "softmax": 4.0 * C_CLASSES * 2,   # read logits, write Q
    "splat": 4.0 * C_CLASSES + 8.0 * (D_FEAT + 1),
    "blur": 0.0,
    "slice": 8.0 * (D_FEAT + 1) + 4.0 * C_CLASSES + 4,
    "mf_update": 8.0 * (D_FEAT + 1) + 4.0 * C_CLASSES + 4 + 4.0 * C_CLASSES,  # slice reads + Q write
    "labels": 4.0 * C_CLASSES + 1,
}
PIPELINE_BYTES_PER_PX = 1350.0       # whole RF + 5-iteration CRF path, SURVEY.md 8d

# stage name -> kernel whose PMC counters (profiles/*_pmc_batch64.json, collected with separate
# rocprofv3 --pmc passes of this very command, profiles/scripts/pmc.sh) give the HBM traffic per launch
STAGE_KERNEL = {"splat": ("rvseg::splat_resident_kernel<9", "rvseg::splat_group_kernel<0, 9, true>"), "mf_update": "rvseg::mf_update_kernel<false, 9, 7>",
                "rf_frames": "rvseg::rf_frames_lazy_kernel", "normal_feature": "rvseg::normal_feature_tiled_kernel",
                "upsample_pack": "rvseg::upsample_pack_kernel<9>", "softmax": "rvseg::softmax_unary_kernel<9>",
                "prep": "rvseg::prep_kernel"}
# stages that are one kernel: the roofline object is reported for the largest of these.  (The lattice
# build is a group of small kernels on a side stream, overlapped with feature extraction + forest
# evaluation; its event time includes that sharing and is listed in stage_ms only.)
SINGLE_KERNEL_STAGES = tuple(STAGE_KERNEL)


def _latest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


class PmcNameError(RuntimeError):
    pass


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the committed PMC summary.  FETCH_SIZE and
    WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 128-byte fabric request as 64 bytes
    (MI355X_MICROARCH.md, HBM section); the factor applied to it is the one measured for the kernel's
    own access shape by profiles/scripts/fetch_calibration (known byte counts, same counter), stored in
    profiles/*_fetch_calibration.json: `stream16` for 16-B-per-lane streaming kernels, `gather36` for
    the splat's 36-byte row gathers + 8-byte CSR stream."""
    f = _latest("*_pmc_batch64.json")
    if not f or stage not in STAGE_KERNEL:
        return None
    with open(f) as fh:
        d = json.load(fh)
    # several candidates: the first that ran (the resident band schedule, else the list-major walk); template
    # arguments added later (block shape ...) follow the ones named here
    names = STAGE_KERNEL[stage] if isinstance(STAGE_KERNEL[stage], tuple) else (STAGE_KERNEL[stage],)
    r = None
    for name in names:
        want = name.rstrip(">")
        r = next((v for k, v in sorted(d.items())
                  if (k == name or k.startswith(want + ",")) and v.get("avg_us_profiled", 0) > 50.0), None)
        if r:
            break
    if not r or "FETCH_SIZE" not in r or "WRITE_SIZE" not in r:
        # a renamed kernel (new template arguments ...) must not turn `traffic` into a silent null
        raise PmcNameError("bench.py: no PMC record for stage %r (kernel %s) in %s -- re-run profiles/scripts/pmc.sh "
                           "and commit the summary, or fix STAGE_KERNEL" % (stage, " | ".join(names), os.path.basename(f)))
    factor = 2.0
    cal = _latest("*_fetch_calibration.json")
    if cal:
        with open(cal) as fh:
            c = json.load(fh)
        factor = float(c.get("factors", {}).get("gather36" if stage == "splat" else "stream16", factor))
    return (factor * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="key frames per step and GPU (default 64; 32 with --config 3)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3),
                    help="BASELINE.json configs[k]: 2 = batch of 64 frames per GPU (default, the headline), 3 = local map of "
                         "256 key frames sharded over 8 GPUs (32 per rank) + label gather to the fusion rank every step")
    ap.add_argument("--no-overlap", action="store_true", help="every kernel alone on the GPU (profiling: no side streams)")
    ap.add_argument("--no-holes", action="store_true", help="frames without invalid-depth holes (round 1/2 inputs)")
    ap.add_argument("--cpu-frames", type=int, default=2, help="frames timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency probe (profiling runs)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run oracle check of one batch frame")
    ap.add_argument("--extras", default="default",
                    help="comma list of extra measurements outside the timed region (N=1 only): host,localmap,config5,train,deep; "
                         "'default' = all that the build supports, 'none' = skip")
    ap.add_argument("--rehearsal", action="store_true",
                    help="CPU rehearsal of the multi-rank protocol (gloo, no GPU, no hot path): launcher, sharding, "
                         "gather, barriers, max-over-ranks timing")
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 32 if args.config == 3 else FRAMES_PER_STEP
    return args


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """Starts one rank process per GPU.  Nothing here initialises the GPU: the device count comes from
    torch.cuda.device_count() (does not create a context on this image) and the children are started
    with subprocess (never exec after GPU init)."""
    n = args.gpus
    if not args.rehearsal and os.environ.get("RVSEG_BENCH_SAME_DEVICE", "0") != "1":   # (one-GPU rehearsal of the rank code path)
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s) (torch.cuda.device_count()); "
                             "refusing to run ranks that would share a device\n" % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all ranks: when one exits with an error (or the whole run takes too long) the others would sit in
    # init_process_group / barrier for ever -- end them and report that rank's code
    deadline = time.time() + float(os.environ.get("RVSEG_BENCH_LAUNCH_TIMEOUT_S", "1500"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0
        if time.time() > deadline:
            sys.stderr.write("bench.py: ranks still running after the launch timeout; terminating them\n")
            rc = 124
            break
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.time() + 10
    for p in procs:
        try:
            p.wait(max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


# ------------------------------------------------------------------------------------------------
def cpu_baseline(n_frames, blob, rgb, depth, calib):
    """The CPU oracle (a port of the reference structure, single thread like the reference's
    inference path, src/segmenter.cpp:336-435) timed on this box's host cores."""
    from oracle import oracle as O
    forest = O.Forest(blob)
    p = O.default_params(dcrf_iterations=CRF_ITERS)
    O.segment_frame(p, forest, 0, rgb[0], depth[0], calib, label_mode=1, unknown=[8])  # warm caches / page in
    t0 = time.perf_counter()
    for i in range(n_frames):
        O.segment_frame(p, forest, 0, rgb[i % len(rgb)], depth[i % len(depth)], calib, label_mode=1, unknown=[8])
    dt = time.perf_counter() - t0
    out = {"value": n_frames * W * H / dt / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "%d synthetic 640x480 frames, RF (C=9) + 5-iter DenseCRF, oracle/rvseg_oracle.c, 1 thread"
                     % n_frames,
           "host_cpus": os.cpu_count()}
    # the same port with frames spread over the host cores (one frame per thread;
    # ctypes releases the GIL inside the C call).  Reported beside the 1-thread figure, not instead of it.
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(os.cpu_count() or 1, 32))

        def one(i):
            O.segment_frame(p, forest, 0, rgb[i % len(rgb)], depth[i % len(depth)], calib, label_mode=1, unknown=[8])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, range(threads)))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": threads * W * H / dt / 1e6, "unit": "Mpix/s", "cores": threads,
                            "sample": "%d frames, one per thread" % threads}
    except Exception as e:  # the single-thread figure above is the contract; this one is extra
        out["all_cores"] = {"error": str(e)}
    return out


def verify_frame(blob, rgb, depth, calib, marg, labels, frame):
    """Post-run check (outside the timed region) of one frame of the LAST timed step against the CPU
    oracle: labels bit-exact, marginals within the north star's 1e-4 (and whether they are bit-identical)."""
    import numpy as np
    from oracle import oracle as O
    forest = O.Forest(blob)
    p = O.default_params(dcrf_iterations=CRF_ITERS)
    _, wm, wl = O.segment_frame(p, forest, 0, rgb[frame], depth[frame], calib, label_mode=1, unknown=[8])
    diff = float(np.abs(marg - wm).max())
    lab_ok = bool(np.array_equal(labels.ravel(), wl))
    return {"ok": lab_ok and diff <= MARGINAL_TOL, "frame": int(frame), "labels_bit_exact": lab_ok,
            "marginals_max_abs_diff": diff, "marginals_bit_exact": bool(np.array_equal(marg, wm)), "tolerance": MARGINAL_TOL}


def rehearsal(args):
    """The multi-rank protocol on CPU (gloo): what runs here is the launcher, the rank environment, frame
    sharding, the preallocated gather and the barrier + max-over-ranks timing -- not the hot path."""
    import torch
    import torch.distributed as dist
    from rovinasemanticsegmentation_amd.distributed import FrameGatherer, shard_frames
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n = args.frames
    Hh, Ww = 6, 8
    start, count = shard_frames(n * world, rank, world)
    f = torch.arange(start, start + count).view(-1, 1)
    local = ((f * 7 + torch.arange(Hh * Ww).view(1, -1)) % 120).to(torch.int8)
    g = FrameGatherer(n * world, (Hh * Ww,), torch.int8, "cpu") if world > 1 else None
    ok = True
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        fused = g.gather(local) if g else local
        if rank == 0:
            fa = torch.arange(n * world).view(-1, 1)
            ok = ok and bool(torch.equal(fused, ((fa * 7 + torch.arange(Hh * Ww).view(1, -1)) % 120).to(torch.int8)))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"rehearsal": True, "metric": "Mpix/s RF+5-iter DenseCRF @640x480 RGB-D", "value": None,
                          "n_gpus": world, "world_size_seen": dist.get_world_size() if world > 1 else 1,
                          "backend": "gloo", "gather_ok": ok, "steps": args.steps, "warmup": args.warmup, "seconds": dt}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.rehearsal:
        sys.exit(rehearsal(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches for a one-GPU box: RVSEG_BENCH_BACKEND=gloo RVSEG_BENCH_SAME_DEVICE=1 runs the
    # multi-rank code path with all ranks on cuda:0 (never used by the driver)
    backend = os.environ.get("RVSEG_BENCH_BACKEND", "nccl")
    same_device = os.environ.get("RVSEG_BENCH_SAME_DEVICE", "0") == "1"
    device_count = torch.cuda.device_count()
    dev_index = 0 if (same_device or world == 1) else local_rank
    if dev_index >= device_count:
        sys.stderr.write("bench.py: rank %d needs cuda:%d but the node exposes %d GPU(s)\n" % (rank, dev_index, device_count))
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    n_gpus = world
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import rovinasemanticsegmentation_amd as rv
    from rovinasemanticsegmentation_amd import synthetic
    from rovinasemanticsegmentation_amd.distributed import FrameGatherer

    n = args.frames
    N = W * H
    blob = synthetic.make_forest_bytes(seed=7, n_trees=4, leaves_per_tree=1 << 14, max_depth=30,
                                       single_classes=C_CLASSES, layer_classes=(8, 9))
    # each rank owns different frames of the local map; the steps alternate between two input sets (frames
    # [rank * n, rank * n + n) of map 0 and of map 1), both resident in HBM before the timed region
    N_SETS = 2
    holes = not args.no_holes
    sets_h = [synthetic.make_batch(n, W, H, holes=holes, start=(k * world + rank) * n) for k in range(N_SETS)]
    rgb_h, depth_h = sets_h[0]
    calib = synthetic.make_calib(W, H)
    d_sets = [(torch.from_numpy(r).to(dev), torch.from_numpy(d.view(np.int16)).to(dev)) for r, d in sets_h]
    d_rgb, d_depth = d_sets[0]
    d_marg = torch.empty((n, C_CLASSES * N), dtype=torch.float32, device=dev)
    d_labels = torch.empty((n, N), dtype=torch.int8, device=dev)
    # the local-map label gather: receive buffers exist once, before the timed region
    gatherer = FrameGatherer(n * world, (N,), torch.int8, dev if backend == "nccl" else "cpu") if world > 1 else None
    d_fused = torch.empty((n, N), dtype=torch.int8, device=dev) if (world == 1 and args.config == 3) else None

    ctx = rv.Context(multi_layer=0, use_dense_crf=1, dcrf_iterations=CRF_ITERS, label_mode=rv.capi.LABEL_CRF,
                     unknown_label=[8], max_batch=n, device=dev.index or 0,
                     # the Segmenter kernel (xyz*0.5, rgb*4) yields ~300 lattice vertices per frame;
                     # 2^12 slots per frame keep the per-vertex launches small (overflow is detected)
                     lattice_capacity_log2=int(os.environ.get("RVSEG_BENCH_CAPACITY_LOG2", "12")),
                     schedule=dict(overlap_build=0, overlap_layers=0) if args.no_overlap else None)
    ctx.forest_load(blob)
    stream = torch.cuda.current_stream(dev)
    if d_fused is not None:   # configs[3] on one GPU: the C ABI's RCCL gather, world size 1
        ctx.comm_init(0, 1, rv.Context.comm_unique_id())
    steps_done = [0]

    def step():
        rgb_d, depth_d = d_sets[steps_done[0] % N_SETS]
        steps_done[0] += 1
        ctx.segment_frames_device(n, rgb_d.data_ptr(), depth_d.data_ptr(), calib, 0, d_marg.data_ptr(),
                                  d_labels.data_ptr(), stream.cuda_stream)
        if world > 1:  # local-map label fusion: one gather to the fusion rank over xGMI
            gatherer.gather(d_labels if backend == "nccl" else d_labels.cpu())
        elif d_fused is not None:
            ctx.gather_frames(d_labels.data_ptr(), d_labels.numel(), d_fused.data_ptr(), 0, stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    async_status = ctx.poll_status(wait=True)   # hash overflow of the last step would show here
    schedule = ctx.last_schedule()
    last_set = (steps_done[0] - 1) % N_SETS

    # per-stage durations of the LAST step, from HIP events recorded on the launch stream
    stages = ctx.last_timing()
    devices_seen = None
    if world > 1:
        devices_seen = [None] * world
        dist.all_gather_object(devices_seen, {"rank": rank, "device": dev_index,
                                              "name": torch.cuda.get_device_name(dev)})

    verified = None
    if rank == 0 and not args.no_verify:
        vf = 9 % n
        verified = verify_frame(blob, sets_h[last_set][0], sets_h[last_set][1], calib, d_marg[vf].cpu().numpy(),
                                d_labels[vf].cpu().numpy(), vf)
        verified["input_set"] = last_set
        if d_fused is not None:
            verified["gather_ok"] = bool(torch.equal(d_fused, d_labels))
            verified["ok"] = verified["ok"] and verified["gather_ok"]

    # single-frame latency (configs[1]) -- outside the timed region
    lat = None
    stages_single = {}
    if rank == 0 and not args.no_latency:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.segment_frames_device(1, d_rgb.data_ptr(), d_depth.data_ptr(), calib, 0, d_marg.data_ptr(),
                                      d_labels.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        lat = (time.perf_counter() - t1) / reps * 1e3
        stages_single = ctx.last_timing()
        ctx.poll_status(wait=True)

    extras = {}
    if rank == 0 and n_gpus == 1 and args.extras != "none":
        try:
            from rovinasemanticsegmentation_amd import bench_extras
            want = None if args.extras == "default" else set(args.extras.split(","))
            extras = bench_extras.run(ctx, dev, blob, rgb_h, depth_h, calib, want)
        except ImportError:
            extras = {}

    if rank == 0:
        px_per_step = n * N
        value = n_gpus * px_per_step * args.steps / dt / 1e6
        # dominant stage and its roofline position
        launches = {"softmax": 1, "splat": CRF_ITERS, "blur": CRF_ITERS, "slice": CRF_ITERS, "mf_update": CRF_ITERS}
        cand = {k: v for k, v in stages.items() if k in SINGLE_KERNEL_STAGES}
        dom = max(cand, key=cand.get) if cand else None
        roof = None
        if dom:
            k = launches.get(dom, 1)
            bytes_per_launch = STAGE_BYTES_PER_PX.get(dom, 0.0) * px_per_step
            avg_ms = stages[dom] / k
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    # PMC counters of this very command (profiles/scripts/pmc.sh); only the headline shape has them
                    "traffic": pmc_traffic(dom) if (n == FRAMES_PER_STEP and args.config == 2 and not args.no_overlap) else None,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": k,
                    "algorithmic_bytes_per_launch": bytes_per_launch}
        out = {
            "metric": "Mpix/s RF+5-iter DenseCRF @640x480 RGB-D", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3]: local map of %d key frames sharded over %d GPU(s), %d per rank, "
                                    % (n * world, world, n) if args.config == 3 else "BASELINE configs[2]: ") +
                                   "batch of %d synthetic 640x480 RGB-D key frames per GPU (10 %% invalid-depth holes: %s; "
                                   "%d input sets in rotation), 4-tree forest (2^14 leaves/tree, D=366, C=9), RF + 5-iter "
                                   "DenseCRF (d=6, Potts w=10), marginals + labels written to HBM"
                                   % (n, "yes" if holes else "no", N_SETS),
                       "baseline_config": args.config,
                       "frames_per_step_per_gpu": n,
                       "label_gather": ("%s gather to rank 0" % ("rccl" if backend == "nccl" else backend)) if world > 1
                                       else ("rccl gather, world size 1 (C ABI)" if d_fused is not None else "none")},
            "schedule": schedule,
            "roofline": roof,
            "verified": (verified["ok"] if verified else None), "verification": verified,
            "async_status": "ok" if async_status == rv.capi.OK else str(async_status),
            "world_size_seen": dist.get_world_size() if world > 1 else 1, "device_count": device_count,
            "devices_seen": devices_seen,
            "pipeline_hbm_frac": round(PIPELINE_BYTES_PER_PX * value * 1e6 / n_gpus / (HBM_PEAK_GBS * 1e9), 5),
            "stage_ms_last_step": {k: round(v, 3) for k, v in stages.items()},
            "latency_ms_single_frame": round(lat, 3) if lat else None,
            "stage_ms_single_frame": {k: round(v, 3) for k, v in stages_single.items()},
        }
        out.update(extras)
        if n_gpus == 1 and not args.no_cpu and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames, blob, rgb_h, depth_h, calib)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
