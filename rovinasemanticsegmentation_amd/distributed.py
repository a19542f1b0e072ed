"""Key-frame sharding and the local-map label gather.

Frames are independent (src/segmenter.cpp:340-434 touches no cross-frame state), so a local map
of F key frames is cut into contiguous blocks, one per rank, with no data-path collective.  The
only exchange is the final label fusion: every rank sends its `uint8/int8[frames][L][H][W]` labels
to the fusion rank.  Backend-agnostic (`nccl` = RCCL over xGMI on the GPUs, `gloo` in CPU tests).

The reference's own fusion adds fp32 posteriors per cloud point in frame order
(src/segmenter.cpp:599-616).  For that variant gather the posteriors instead (`gather_frames`, 36x the
bytes) and run `Segmenter.processMap` / `rvseg_fuse_posteriors` on the fusion rank: the gathered
tensor is in global frame order, so the sums run in the same order as on one GPU.  An all-reduce
would be ring-bound and would change the fp32 order.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """Contiguous block of frames owned by `rank`: (start, count).  Blocks differ by at most one."""
    base, rem = divmod(n_frames, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_labels(local_labels, n_frames_total, dst=0, group=None):
    """Gathers per-rank label blocks `[count_r, ...]` to `dst` in frame order.

    Returns the fused `[n_frames_total, ...]` tensor on `dst`, None elsewhere.  Ranks may own
    different frame counts; blocks are padded to the largest count for the collective and
    trimmed afterwards (one direct peer-to-root transfer per rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_frames(n_frames_total, r, world)[1] for r in range(world)]
    cmax = max(counts)
    tail = tuple(local_labels.shape[1:])
    if local_labels.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d frames, expected %d" % (rank, local_labels.shape[0], counts[rank]))
    send = local_labels
    if counts[rank] != cmax:
        send = torch.zeros((cmax,) + tail, dtype=local_labels.dtype, device=local_labels.device)
        send[:counts[rank]] = local_labels
    send = send.contiguous()
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)


def gather_frames(local_block, n_frames_total, dst=0, group=None):
    """`gather_labels` for any per-frame tensor (e.g. fp32 posteriors `[count_r, S*H*W]`)."""
    return gather_labels(local_block, n_frames_total, dst=dst, group=group)
