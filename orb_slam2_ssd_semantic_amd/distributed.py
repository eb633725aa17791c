"""Batched keyframe mode across the GPUs of one node.

Frames are independent, so a keyframe batch is sharded in contiguous blocks, one process per GPU, with NO
collective on the data path.  The single exchange step is one all-gather of the three fixed-size padded
result buffers (counts, keypoints, descriptors) so that every rank holds the whole batch's descriptors
(what KeyFrameDatabase / LoopClosing consume serially in the reference, src/LoopClosing.cc:312-342).
torch.distributed is plumbing: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests.
"""
import torch
import torch.distributed as dist


def shard_range(nframes, rank, world):
    """Contiguous block of frames owned by `rank`: [lo, hi). Remainder frames go to the lowest ranks."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(nframes, world):
    return -(-nframes // world)


def all_gather_keyframes(n, kps, desc, nframes_total=None, group=None):
    """All-gather per-rank results.

    n    : int32 [S]            keypoint counts of this rank's S frames (S = max_shard; unused slots 0)
    kps  : int32 [S, cap, 7]    orbfe_keypoint records, bit-cast to int32
    desc : uint8 [S, cap, 32]
    Returns (n_all [W*S], kps_all [W*S, cap, 7], desc_all [W*S, cap, 32]) in rank-major order; with
    `nframes_total` the padding slots of uneven shards are removed so the result is in frame order.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = (n, kps, desc)
    else:
        out = []
        for t in (n, kps, desc):
            g = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(g, t.contiguous(), group=group)
            out.append(g)
        out = tuple(out)
    if nframes_total is None:
        return out
    S = n.shape[0]
    idx = []
    for r in range(world):
        lo, hi = shard_range(nframes_total, r, world)
        idx.extend(range(r * S, r * S + (hi - lo)))
    idx = torch.as_tensor(idx, dtype=torch.long, device=n.device)
    return tuple(t.index_select(0, idx) for t in out)


class OverlappedKeyframeGather:
    """Double-buffered, asynchronous form of `all_gather_keyframes` for a steady stream of batches.

    The producer alternates between two output sets (n, kps, desc).  `acquire(k)` makes the current stream wait until
    the gather that last read set k has finished (stream-level on RCCL, host-level on gloo), `launch(k)` starts the
    all-gather of set k after the work already enqueued on the current stream; it then overlaps whatever is enqueued
    next.  `result(k)` waits for and returns the gathered (n, kps, desc) of set k in rank-major order.
    """

    def __init__(self, sets, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sets = sets
        self.pending = [[] for _ in sets]
        self.gathered = [tuple(torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                                           device=t.device) for t in s) for s in sets]

    def acquire(self, k):
        for w in self.pending[k]:
            w.wait()
        self.pending[k] = []

    def launch(self, k):
        if self.world == 1:
            for dst, src in zip(self.gathered[k], self.sets[k]):
                dst.copy_(src)
            return
        for dst, src in zip(self.gathered[k], self.sets[k]):
            self.pending[k].append(dist.all_gather_into_tensor(dst, src.contiguous(), group=self.group, async_op=True))

    def result(self, k):
        self.acquire(k)
        return self.gathered[k]
