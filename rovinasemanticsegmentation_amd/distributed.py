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


class FrameGatherer:
    """The local-map gather with every buffer allocated once: one direct peer-to-root transfer per
    rank and step, no per-step allocation and no concatenation on the root.

    The root owns ONE `[world * cmax, ...]` receive tensor; rank r's block lands in its slice
    `[r * cmax, r * cmax + count_r)`.  With equal shards (the bench: 64 frames per rank) the receive
    tensor IS the fused result; with uneven shards (counts differ by one) the last `world - rem`
    blocks are moved up by at most one frame each into a second preallocated tensor."""

    def __init__(self, n_frames_total, tail_shape, dtype, device, dst=0, group=None):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total = n_frames_total
        self.counts = [shard_frames(n_frames_total, r, self.world)[1] for r in range(self.world)]
        self.starts = [shard_frames(n_frames_total, r, self.world)[0] for r in range(self.world)]
        self.cmax = max(self.counts) if self.counts else 0
        tail = tuple(tail_shape)
        self.even = all(c == self.cmax for c in self.counts)
        self.send = None if self.counts[self.rank] == self.cmax else torch.zeros((self.cmax,) + tail, dtype=dtype, device=device)
        self.recv = self.views = self.out = None
        if self.rank == dst:
            self.recv = torch.empty((self.world * self.cmax,) + tail, dtype=dtype, device=device)
            self.views = [self.recv[r * self.cmax:(r + 1) * self.cmax] for r in range(self.world)]
            self.out = self.recv if self.even else torch.empty((n_frames_total,) + tail, dtype=dtype, device=device)

    def gather(self, local_block):
        if local_block.shape[0] != self.counts[self.rank]:
            raise ValueError("rank %d holds %d frames, expected %d" % (self.rank, local_block.shape[0], self.counts[self.rank]))
        send = local_block
        if self.send is not None:
            self.send[:self.counts[self.rank]].copy_(local_block)
            send = self.send
        dist.gather(send.contiguous(), self.views, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if not self.even:
            for r in range(self.world):
                self.out[self.starts[r]:self.starts[r] + self.counts[r]].copy_(self.views[r][:self.counts[r]])
        return self.out


def gather_labels(local_labels, n_frames_total, dst=0, group=None):
    """Gathers per-rank label blocks `[count_r, ...]` to `dst` in frame order (one-shot form of
    FrameGatherer; a caller with a steady frame count keeps a FrameGatherer instead).

    Returns the fused `[n_frames_total, ...]` tensor on `dst`, None elsewhere."""
    g = FrameGatherer(n_frames_total, local_labels.shape[1:], local_labels.dtype, local_labels.device, dst, group)
    return g.gather(local_labels)


def gather_frames(local_block, n_frames_total, dst=0, group=None):
    """`gather_labels` for any per-frame tensor (e.g. fp32 posteriors `[count_r, S*H*W]`)."""
    return gather_labels(local_block, n_frames_total, dst=dst, group=group)
