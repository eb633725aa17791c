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
