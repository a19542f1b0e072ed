"""Multi-process path on CPU: frame sharding + label gather with the gloo backend, world size 2
(and 3 for uneven shards).  The same functions run over RCCL in bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rovinasemanticsegmentation_amd.distributed import gather_frames, gather_labels, shard_frames


def test_shard_frames_partition():
    for n in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [shard_frames(n, r, world) for r in range(world)]
            assert sum(c for _, c in blocks) == n
            pos = 0
            for s, c in blocks:
                assert s == pos
                pos += c
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, count = shard_frames(n_frames, rank, world)
        L, H, W = 2, 6, 8
        # "labels" that encode (frame, layer, pixel) so that order and content are checkable
        f = torch.arange(start, start + count).view(-1, 1, 1, 1)
        l = torch.arange(L).view(1, -1, 1, 1)
        p = torch.arange(H * W).view(1, 1, H, W)
        local = ((f * 7 + l * 3 + p) % 120).to(torch.int8)
        fused = gather_labels(local, n_frames, dst=0)
        if rank == 0:
            fa = torch.arange(n_frames).view(-1, 1, 1, 1)
            want = ((fa * 7 + l * 3 + p) % 120).to(torch.int8)
            q.put(bool(torch.equal(fused, want)) and tuple(fused.shape) == (n_frames, L, H, W))
        else:
            assert fused is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (8, 256): BASELINE configs[3] -- the local map of 256 key frames over 8 ranks, 32 frames each
@pytest.mark.parametrize("world,n_frames", [(2, 8), (2, 5), (3, 7), (8, 256)])
def test_label_gather_gloo(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def _fusion_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        H, W, cc, P = 4, 6, [3, 2], 9
        rng = np.random.default_rng(11)
        post_all = (rng.standard_normal((n_frames, sum(cc) * H * W)) * 10.0 ** rng.integers(-2, 7, (n_frames, sum(cc) * H * W))).astype(np.float32)
        idx_all = rng.integers(-1, P, (n_frames, H, W)).astype(np.int32)
        start, count = shard_frames(n_frames, rank, world)
        fused_post = gather_frames(torch.from_numpy(post_all[start:start + count].copy()), n_frames, dst=0)
        if rank == 0:
            from oracle import oracle   # the checker: rank 0 fuses the gathered posteriors in frame order
            got = oracle.fuse_posteriors(idx_all, fused_post.numpy(), cc, P)
            want = oracle.fuse_posteriors(idx_all, post_all, cc, P)
            q.put(bool(np.array_equal(got, want)) and bool(np.array_equal(fused_post.numpy(), post_all)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_posterior_gather_keeps_the_fusion_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fusion_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def _gatherer_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rovinasemanticsegmentation_amd.distributed import FrameGatherer
        start, count = shard_frames(n_frames, rank, world)
        g = FrameGatherer(n_frames, (5,), torch.int32, "cpu")
        ok = True
        ptrs = set()
        for step in range(3):   # the same preallocated buffers serve every step
            local = (torch.arange(start, start + count).view(-1, 1) * 10 + torch.arange(5).view(1, -1) + 1000 * step).to(torch.int32)
            fused = g.gather(local)
            if rank == 0:
                want = (torch.arange(n_frames).view(-1, 1) * 10 + torch.arange(5).view(1, -1) + 1000 * step).to(torch.int32)
                ok = ok and bool(torch.equal(fused, want))
                ptrs.add(fused.data_ptr())
        if rank == 0:
            q.put(ok and len(ptrs) == 1)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 6), (3, 8), (8, 256)])
def test_frame_gatherer_reuses_its_buffers(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_bench_launcher_starts_its_own_ranks_in_rehearsal_mode():
    """`python bench.py --gpus 2` must itself start two rank processes (the driver's round-1 scaling run
    found a bench that ignored --gpus).  Rehearsal mode runs the rank protocol on CPU with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearsal", "--frames", "3",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["gather_ok"] is True and out["rehearsal"] is True


def test_bench_launcher_refuses_more_ranks_than_gpus():
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0
    assert "--gpus 2 requested" in r.stderr and "GPU(s)" in r.stderr
