"""world_size-2 gloo test of the batched-keyframe sharding + all-gather plumbing.  CPU only:
the per-rank producer here is the oracle (the GPU kernels cannot run on this box); what is under test is
shard_range / all_gather_keyframes, i.e. exactly what bench.py --gpus N executes around the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orb_slam2_ssd_semantic_amd.distributed import all_gather_keyframes, max_shard, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 1024, 1000):
        for w in (1, 2, 3, 4, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(n, w) or n == 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, cap, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle_ffi as O
        from orb_slam2_ssd_semantic_amd.synth import synth_frame
        e = O.OracleExtractor(300, 1.2, 8, 20, 7)
        lo, hi = shard_range(nframes, rank, world)
        S = max_shard(nframes, world)
        n = torch.zeros(S, dtype=torch.int32)
        kps = torch.zeros(S, cap, 7, dtype=torch.int32)
        desc = torch.zeros(S, cap, 32, dtype=torch.uint8)
        for i, f in enumerate(range(lo, hi)):
            k, d = e(synth_frame(500 + f, 240, 320))
            n[i] = len(k)
            kps[i, :len(k)] = torch.from_numpy(k.view(np.int32).reshape(-1, 7).copy())
            desc[i, :len(k)] = torch.from_numpy(d)
        n_all, k_all, d_all = all_gather_keyframes(n, kps, desc, nframes_total=nframes)
        assert n_all.shape[0] == nframes
        ret[rank] = (n_all.numpy().copy(), k_all.numpy().copy(), d_all.numpy().copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [4, 5])
def test_all_gather_world2(oracle, nframes):
    from orb_slam2_ssd_semantic_amd.synth import synth_frame
    world, cap = 2, 384
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, nframes, cap, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    e = oracle.OracleExtractor(300, 1.2, 8, 20, 7)
    for r in range(world):
        n_all, k_all, d_all = ret[r]
        for f in range(nframes):                   # every rank ends with the whole batch, in frame order
            k, d = e(synth_frame(500 + f, 240, 320))
            assert n_all[f] == len(k)
            assert np.array_equal(k_all[f, :len(k)].reshape(-1), k.view(np.int32).reshape(-1))
            assert np.array_equal(d_all[f, :len(k)], d)
            assert not d_all[f, len(k):].any()
    assert np.array_equal(ret[0][2], ret[1][2])


def _overlap_worker(rank, world, port, steps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orb_slam2_ssd_semantic_amd.distributed import OverlappedKeyframeGather
        S, cap = 3, 5
        sets = [(torch.zeros(S, dtype=torch.int32), torch.zeros(S, cap, 7, dtype=torch.int32),
                 torch.zeros(S, cap, 32, dtype=torch.uint8)) for _ in range(2)]
        g = OverlappedKeyframeGather(sets)
        seen = []
        for i in range(steps):  # exactly the loop of bench.py's step(): acquire set k, produce into it, launch its gather
            k = i & 1
            g.acquire(k)
            n, kps, desc = sets[k]
            n.fill_(100 * i + rank)
            kps.fill_(1000 * i + rank)
            desc.fill_((7 * i + rank) % 256)
            g.launch(k)
            if i >= 1:  # the gather of the previous step completes while this one is produced
                pn, pk, pd = g.result((i - 1) & 1)
                seen.append((pn.clone(), pk[:, 0, 0].clone(), pd[:, 0, 0].clone()))
        ret[rank] = [(a.numpy(), b.numpy(), c.numpy()) for a, b, c in seen]
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_world2():
    world, steps = 2, 6
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, port, steps, ret), nprocs=world, join=True)
    for rank in range(world):
        for j, (n, k, d) in enumerate(ret[rank]):   # j = index of the step whose gather this is
            for r in range(world):
                assert (n[r * 3:(r + 1) * 3] == 100 * j + r).all()
                assert (k[r * 3:(r + 1) * 3] == 1000 * j + r).all()
                assert (d[r * 3:(r + 1) * 3] == (7 * j + r) % 256).all()


def _prefix_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from orb_slam2_ssd_semantic_amd.distributed import OverlappedKeyframeGather
        S, cap, gcap = 3, 8, 5
        sets = [(torch.zeros(S, dtype=torch.int32), torch.zeros(S, cap, 7, dtype=torch.int32),
                 torch.zeros(S, cap, 32, dtype=torch.uint8)) for _ in range(2)]
        g = OverlappedKeyframeGather(sets, gather_cap=gcap)
        full = OverlappedKeyframeGather(sets)
        out = {"bytes": (g.bytes_per_rank, full.bytes_per_rank), "trunc": []}
        for i in range(4):
            k = i & 1
            g.acquire(k)
            n, kps, desc = sets[k]
            cnt = [3, 5, 4] if i < 3 else [3, 7, 4]        # the last step has a frame with more keypoints than slots travel
            n.copy_(torch.tensor(cnt, dtype=torch.int32))
            for f in range(S):
                kps[f].zero_(); desc[f].zero_()
                kps[f, :cnt[f]] = 1000 * i + 10 * rank + f
                desc[f, :cnt[f]] = (16 * i + 4 * rank + f) % 256
            g.launch(k)
            gn, gk, gd = g.result(k)
            assert gk.shape == (world * S, gcap, 7) and gd.shape == (world * S, gcap, 32) and gn.shape == (world * S,)
            for r in range(world):
                for f in range(S):
                    c = min(cnt[f], gcap)
                    assert int(gn[r * S + f]) == cnt[f]
                    assert (gk[r * S + f, :c] == 1000 * i + 10 * r + f).all() and not gk[r * S + f, c:].any()
                    assert (gd[r * S + f, :c] == (16 * i + 4 * r + f) % 256).all()
            out["trunc"].append(g.truncated(k))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_of_the_valid_prefix_world2():
    """gather_cap: only the first gather_cap slots of every frame travel (bench.py: 1024 of 1088 at 1000 features); the counts travel
    whole, so a frame that had more keypoints than slots travelled is reported by truncated()."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_prefix_worker, args=(world, port, ret), nprocs=world, join=True)
    for rank in range(world):
        small, full = ret[rank]["bytes"]
        assert small == 3 * 4 + 3 * 5 * 28 + 3 * 5 * 32 and full == 3 * 4 + 3 * 8 * 28 + 3 * 8 * 32
        assert ret[rank]["trunc"] == [0, 0, 0, world]
