"""The C-ABI's multi-device layer (include/orbfe.h orbfe_group_*): batched keyframe mode for C / C++ hosts -- contiguous
shards, one in-place ncclAllGather (RCCL) of the padded blocks, and the consumer of the gather.  CPU: the sharding
arithmetic equals distributed.shard_range.  GPU (one device on the box): a world-1 group of both kinds, so that RCCL is
loaded, a communicator is created and ncclAllGather has executed on the blocks; extraction and the consumer's matches are
compared with the oracle."""
import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd.distributed import KeyframeGroup, shard_range
from orb_slam2_ssd_semantic_amd.synth import synth_frame


def test_shard_range_of_the_c_abi_equals_the_python_one():
    for n in (0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 1024, 1025):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = KeyframeGroup.shard_range_c(n, r, world)
                assert (lo, hi) == shard_range(n, r, world)
                covered += list(range(lo, hi))
            assert covered == list(range(n))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["local", "rank"])
def test_world1_group_extract_allgather_match(oracle, kind):
    nf, w, h, n = 1000, 640, 480, 6
    if kind == "local":
        g = KeyframeGroup(nf, 1.2, 8, 20, 7, w, h, 8, devices=(0,))
    else:
        g = KeyframeGroup(nf, 1.2, 8, 20, 7, w, h, 8, rank_of_world=(0, 1, KeyframeGroup.unique_id()), device=0)
    assert g.world == 1 and g.frames_padded == 8 and g.cap >= nf
    frames = np.stack([synth_frame(700 + i, h, w, sparse=(i == 3)) for i in range(n)])
    oe = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    ref = [oe(f) for f in frames]
    for rep in range(2):   # the second round reuses the blocks behind the previous gather
        g.extract_batch(frames)
        g.allgather()
        g.synchronize()
        for i in range(n):
            assert g.block_index(i) == i
            gk, gd = g.get_frame(i)
            ok, od = ref[i]
            assert len(gk) == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(gd, od), (kind, rep, i)
        q = np.array([0, 1, 2, 5, 3], np.int32)
        t = np.array([1, 0, 5, 2, 3], np.int32)
        m, nm = g.match(q, t, 0.9, 100, True)
        for p in range(len(q)):
            (qk, qd), (tk, td) = ref[q[p]], ref[t[p]]
            om, _, _, on = oracle.match_bf(qd, td, qk["angle"], tk["angle"], 0.9, 100, True)
            assert nm[p] == on and np.array_equal(m[p, :len(qk)], om), (kind, p)
            assert (m[p, len(qk):] == -1).all()
    g.close()


def test_owner_rank_and_block_index_over_uneven_shards():
    """Pure index arithmetic of the block layout (no device): every frame's owner is the rank whose shard_range holds it, its
    slot is rank * shard + offset inside the shard, slots are distinct, and the cut is base + remainder (10 over 4 = 3, 3, 2, 2)."""
    assert [KeyframeGroup.shard_range_c(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    for world in (1, 2, 3, 4, 5, 8):
        for n in (1, 2, 3, 5, 7, 8, 9, 10, 63, 64, 65, 1000, 1024, 1025):
            for shard in {-(-n // world), -(-n // world) + 3}:
                seen = set()
                for f in range(n):
                    r = KeyframeGroup.owner_rank_c(n, world, f)
                    lo, hi = shard_range(n, r, world)
                    assert lo <= f < hi, (n, world, f, r)
                    b = KeyframeGroup.block_index_c(n, world, shard, f)
                    assert b == r * shard + (f - lo) and b not in seen
                    seen.add(b)
            assert KeyframeGroup.block_index_c(n, world, -(-n // world) - 1, 0) == -1 or world * (-(-n // world) - 1) >= n
            assert KeyframeGroup.owner_rank_c(n, world, n) == -1 and KeyframeGroup.owner_rank_c(n, world, -1) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("members", [2, 3])
def test_multi_member_group_on_one_device_copy_transport(oracle, members):
    """2- and 3-member local groups on device 0 through the copy transport (hipMemcpyAsync between the members' blocks in
    place of ncclAllGather, same streams and events): uneven shards (5, 7, 64 frames in slices sized for 64), block_index,
    zero tails and untouched slots, every member holding every frame after the gather, a second round that reuses the
    blocks, and the consumer with pairs that cross shard boundaries -- all against the oracle."""
    nf, w, h, nmax = 500, 320, 240, 64
    g = KeyframeGroup(nf, 1.2, 8, 20, 7, w, h, nmax, devices=(0,) * members, transport=KeyframeGroup.COPY)
    assert g.world == members and g.members == members and g.transport == KeyframeGroup.COPY
    shard = -(-nmax // members)
    assert g.frames_padded == members * shard
    frames = np.stack([synth_frame(7100 + i, h, w, sparse=(i % 5 == 3)) for i in range(nmax)])
    oe = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    ref = [oe(f) for f in frames]
    rng = np.random.default_rng(members)
    for n in (64, 5, 7, 64):                      # shrinking then growing: stale slots of the larger round must read as empty
        sub = frames[:n] if n != 7 else frames[20:27]
        refs = ref[:n] if n != 7 else ref[20:27]
        g.extract_batch(sub)
        g.allgather()
        g.synchronize()
        expect_n = np.zeros(g.frames_padded, np.int32)
        for f in range(n):
            b = g.block_index(f)
            assert b == KeyframeGroup.block_index_c(n, members, shard, f)
            expect_n[b] = len(refs[f][0])
        for m in range(members):
            assert np.array_equal(g.counts(m), expect_n), (n, m)      # zero tails, zero unused slices, right slots
            for f in range(n):
                gk, gd = g.get_frame(f, member=m)
                ok, od = refs[f]
                assert len(gk) == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(gd, od), (n, m, f)
        # the consumer: query frames of every shard against frames of OTHER shards (and one of its own)
        q = rng.integers(0, n, 12).astype(np.int32)
        t = rng.integers(0, n, 12).astype(np.int32)
        q[:members], t[:members] = [shard_range(n, r, members)[0] for r in range(members)], [shard_range(n, (r + 1) % members, members)[1] - 1 for r in range(members)]
        assert len({KeyframeGroup.owner_rank_c(n, members, int(a)) for a in q}) == min(members, n)
        mm, nm = g.match(q, t, 0.9, 100, True)
        for p in range(len(q)):
            (qk, qd), (tk, td) = refs[q[p]], refs[t[p]]
            om, _, _, on = oracle.match_bf(qd, td, qk["angle"], tk["angle"], 0.9, 100, True)
            assert nm[p] == on and np.array_equal(mm[p, :len(qk)], om), (n, p)
            assert (mm[p, len(qk):] == -1).all()
    g.close()


@pytest.mark.gpu
def test_multi_member_group_device_shards_and_device_consumer(oracle):
    """The device-resident form with three members on device 0: each member is handed ITS shard (cut by
    orbfe_group_shard_range) as a device pointer, the gather runs, and orbfe_group_match_device matches block indices on
    each member's own stream -- what bench.py's config 4 does with one member per rank."""
    import torch
    members, nf, w, h, n = 3, 500, 320, 240, 10
    g = KeyframeGroup(nf, 1.2, 8, 20, 7, w, h, 12, devices=(0,) * members, transport=KeyframeGroup.COPY)
    frames = np.stack([synth_frame(7300 + i, h, w) for i in range(n)])
    oe = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    ref = [oe(f) for f in frames]
    dg = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    for r in range(members):
        lo, hi = KeyframeGroup.shard_range_c(n, r, members)
        assert (hi - lo) == (4, 3, 3)[r]
        g.extract_shard_device(r, dg[lo:].data_ptr() if hi > lo else None, n, w, h, w, w * h)
    g.allgather()
    pairs = [(0, 9), (4, 0), (9, 4), (3, 7), (7, 7)]
    for r in range(members):
        lo, hi = KeyframeGroup.shard_range_c(n, r, members)
        mine = [(q, t) for q, t in pairs if lo <= q < hi]
        qb = torch.tensor([g.block_index(q) for q, _ in mine], dtype=torch.int32, device="cuda")
        tb = torch.tensor([g.block_index(t) for _, t in mine], dtype=torch.int32, device="cuda")
        dm = torch.full((len(mine), g.cap), -7, dtype=torch.int32, device="cuda")
        dn = torch.zeros(len(mine), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g.match_device(r, qb.data_ptr(), tb.data_ptr(), len(mine), dm.data_ptr(), dn.data_ptr())
        g.synchronize()
        mm, nm = dm.cpu().numpy(), dn.cpu().numpy()
        for i, (q, t) in enumerate(mine):
            (qk, qd), (tk, td) = ref[q], ref[t]
            om, _, _, on = oracle.match_bf(qd, td, qk["angle"], tk["angle"], 0.9, 100, True)
            assert nm[i] == on and np.array_equal(mm[i, :len(qk)], om), (r, q, t)
    g.close()


@pytest.mark.gpu
def test_rank_groups_refuse_the_copy_transport_and_duplicate_devices_need_it():
    """RCCL refuses two ranks on one device, which is why the copy transport exists; asking for it is explicit."""
    from orb_slam2_ssd_semantic_amd._ffi import OrbfeError
    with pytest.raises(OrbfeError):
        KeyframeGroup(500, 1.2, 8, 20, 7, 320, 240, 8, devices=(0, 0), transport=KeyframeGroup.RCCL)
    with pytest.raises(OrbfeError):
        KeyframeGroup(500, 1.2, 8, 20, 7, 320, 240, 8, devices=(0, 0), transport=7)
