"""MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:284-345), batched.

CPU part: the C oracle against a definition-level numpy twin.  GPU part: the HIP path (through the C-ABI) against the
oracle, index- and value-exact."""
import numpy as np
import pytest


def make_case(seed, npoints, max_obs, npool=400):
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 256, (npool, 32), dtype=np.uint8)
    # map points are clusters: noisy copies of one descriptor (so medians differ meaningfully), plus a few duplicates
    base = rng.integers(0, 256, (max(npoints, 1), 32), dtype=np.uint8)
    counts = rng.integers(0, max_obs + 1, npoints)
    if npoints > 3:
        counts[0], counts[1], counts[2] = 0, 1, 2
    off = np.zeros(npoints + 1, np.uint32)
    off[1:] = np.cumsum(counts)
    idx = rng.integers(0, npool, int(off[-1])).astype(np.uint32)
    for p in range(npoints):
        for k in range(off[p], off[p + 1]):
            if rng.random() < 0.7:
                flips = rng.integers(0, 60)
                d = base[p].copy()
                bits = rng.integers(0, 256, flips)
                for b in bits:
                    d[b >> 3] ^= 1 << (b & 7)
                pool[idx[k]] = d
    return pool, off, idx


def twin(pool, off, idx):
    bits = np.unpackbits(pool, axis=1).astype(np.int32)
    best, med = [], []
    for p in range(len(off) - 1):
        ob = idx[off[p]:off[p + 1]]
        n = len(ob)
        if n == 0:
            best.append(-1)
            med.append(-1)
            continue
        b = bits[ob]
        d = (b[:, None, :] != b[None, :, :]).sum(-1)
        m = np.sort(d, axis=1)[:, int(0.5 * (n - 1))]
        best.append(int(np.argmin(m)))  # first minimum
        med.append(int(m.min()))
    return np.array(best, np.int32), np.array(med, np.int32)


@pytest.mark.parametrize("seed,npoints,max_obs", [(0, 60, 12), (1, 20, 70), (2, 0, 5), (3, 5, 1)])
def test_oracle_distinctive_vs_twin(oracle, seed, npoints, max_obs):
    pool, off, idx = make_case(seed, npoints, max_obs)
    b, m = oracle.distinctive(pool, off, idx)
    tb, tm = twin(pool, off, idx)
    assert np.array_equal(b, tb) and np.array_equal(m, tm)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,npoints,max_obs", [(0, 300, 12), (1, 40, 70), (2, 0, 5), (3, 5, 1), (4, 3, 200),
                                                  (5, 2000, 8)])
def test_gpu_distinctive_parity(oracle, seed, npoints, max_obs):
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    mt = ORBmatcher(0.9, True)
    pool, off, idx = make_case(seed, npoints, max_obs)
    b, m = mt.ComputeDistinctiveDescriptors(pool, off, idx)
    rb, rm = oracle.distinctive(pool, off, idx)
    assert np.array_equal(b, rb) and np.array_equal(m, rm)


def test_distinctive_bad_args_cpu():
    from orb_slam2_ssd_semantic_amd import _ffi
    assert _ffi.lib().orbfe_distinctive_descriptors(None, None, 0, None, None, 0, None, None) == _ffi.ORBFE_ERR_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("seed,npoints,max_obs,lds_obs", [(0, 300, 12, 12), (1, 40, 70, 128), (4, 3, 200, 1024), (6, 50, 30, 16)])
def test_gpu_distinctive_device_buffers(oracle, seed, npoints, max_obs, lds_obs):
    """The device entry point on torch tensors: same answers as the oracle; points with more observations than the LDS was
    sized for (last case) are marked -2 instead of being computed."""
    import torch
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    pool, off, idx = make_case(seed, npoints, max_obs)
    rb, rm = oracle.distinctive(pool, off, idx)
    dp = torch.from_numpy(pool).cuda()
    do = torch.from_numpy(off.astype(np.int32)).cuda()
    di = torch.from_numpy(np.ascontiguousarray(idx).astype(np.int32)).cuda() if len(idx) else torch.zeros(1, dtype=torch.int32, device="cuda")
    db = torch.full((npoints,), 99, dtype=torch.int32, device="cuda")
    dm = torch.full((npoints,), 99, dtype=torch.int32, device="cuda")
    ORBmatcher(0.9, True).ComputeDistinctiveDescriptors_device(dp.data_ptr(), do.data_ptr(), di.data_ptr(), npoints, lds_obs, db.data_ptr(),
                                                               dm.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    b, m = db.cpu().numpy(), dm.cpu().numpy()
    nobs = np.diff(off.astype(np.int64))
    big = nobs > lds_obs
    assert np.array_equal(b[~big], rb[~big]) and np.array_equal(m[~big], rm[~big])
    assert np.all(b[big] == -2) and np.all(m[big] == -2)
    assert big.any() == (lds_obs < max_obs and nobs.max() > lds_obs)
