"""Parity of the HIP matcher kernels (through the C-ABI) against the CPU oracle.  Integer work: bit-exact."""
import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd.synth import synth_frame
from test_matcher_oracle import make_desc, random_fv, to_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mt():
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    return ORBmatcher(0.9, True)


@pytest.mark.parametrize("seed,nq,nt,flips,ratio,th", [(0, 1000, 1004, 40, 0.9, 100), (1, 37, 2100, 60, 0.6, 50),
                                                       (2, 1, 5, 10, 0.9, 100), (3, 300, 1, 10, 0.9, 100),
                                                       (4, 64, 64, 0, 0.9, 100), (5, 0, 10, 0, 0.9, 100),
                                                       (6, 10, 0, 0, 0.9, 100), (7, 65, 17, 3, 0.75, 100),
                                                       (8, 4000, 4030, 80, 0.9, 100)])
def test_match_bf_parity(oracle, seed, nq, nt, flips, ratio, th):
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    rng = np.random.default_rng(seed)
    t = make_desc(rng, nt)
    q = make_desc(rng, nq, base=t[rng.integers(0, nt, nq)], flips=flips) if nt and nq else make_desc(rng, nq)
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    for ori in (True, False):
        m = ORBmatcher(ratio, ori)
        got = m.MatchBruteForce(q, t, qa, ta, th)
        ref = oracle.match_bf(q, t, qa, ta, ratio, th, ori)
        for g, r in zip(got[:3], ref[:3]):
            assert np.array_equal(g, r)
        assert got[3] == ref[3]


def test_match_bf_ties(mt, oracle):
    rng = np.random.default_rng(0)
    t = make_desc(rng, 40)
    t[33] = t[2]
    t[17] = t[2]                      # three identical rows spread over different wave slices
    q = t[[2, 5]].copy()
    m, b, s, n = mt.MatchBruteForce(q, t, None, None)
    mo, bo, so, no = oracle.match_bf(q, t, None, None, 0.9, 100, False)
    assert np.array_equal(m, mo) and np.array_equal(b, bo) and np.array_equal(s, so) and n == no
    assert b[0] == 0 and s[0] == 0 and m[0] == -1 and m[1] == 5


def test_match_consecutive_frames_config3(oracle):
    """BASELINE config 3: extract + BF match to the previous frame with the frame-to-frame parameters."""
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher
    e = ORBextractor()
    m = ORBmatcher(0.9, True)
    prev = None
    for i in range(4):
        k, d = e(synth_frame(200 + i))
        if prev is not None:
            got = m.MatchBruteForce(d, prev[1], k["angle"], prev[0]["angle"], ORBmatcher.TH_HIGH)
            ref = oracle.match_bf(d, prev[1], k["angle"], prev[0]["angle"], 0.9, 100, True)
            assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3]
        prev = (k, d)
    # a frame against itself: every descriptor finds itself at distance 0
    got = ORBmatcher(0.9, False).MatchBruteForce(d, d, None, None)
    assert (got[1] == 0).all()
    dup = got[2] == 0                                   # exact duplicate descriptors fail the ratio test
    assert np.array_equal(got[0][~dup], np.arange(len(d))[~dup])


@pytest.mark.parametrize("seed,strict,with_valid_f,nK,nF,nn", [(0, False, False, 150, 160, 20), (1, True, True, 150, 160, 20),
                                                               (2, False, False, 1000, 1010, 110), (3, True, True, 1200, 900, 90),
                                                               (4, False, True, 5, 7, 3)])
def test_search_by_bow_parity(oracle, seed, strict, with_valid_f, nK, nF, nn):
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    rng = np.random.default_rng(seed)
    dF = make_desc(rng, nF)
    dK = make_desc(rng, nK, base=dF[rng.integers(0, nF, nK)], flips=40)
    vK = (rng.uniform(size=nK) < 0.8).astype(np.uint8)
    vF = (rng.uniform(size=nF) < 0.8).astype(np.uint8) if with_valid_f else None
    aK = rng.uniform(0, 360, nK).astype(np.float32)
    aF = np.mod(aK[rng.integers(0, nK, nF)] + rng.normal(0, 20, nF), 360).astype(np.float32)
    ids = np.sort(rng.choice(100000, nn + 5, replace=False))
    fvK = random_fv(rng, nK, nn, ids[:nn])
    fvF = random_fv(rng, nF, nn, ids[5:])
    for ori in (True, False):
        m = ORBmatcher(0.7, ori)
        got = m.SearchByBoW(dK, vK, aK, fvK, dF, vF, aF, fvF, strict_lt=strict)
        ref = oracle.search_by_bow(dK, vK, aK, to_csr(fvK), dF, vF, aF, to_csr(fvF), 0.7, 50, strict, ori)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]


def test_search_by_bow_reference_golden():
    """HIP SearchByBoW x2 against vectors produced by the reference's own compiled ORBmatcher
    (tests/golden/make_golden.py, oracle/_ref): M1 :217-363 and M2 :665-812."""
    import os
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_golden.npz"))
    k0, d0, k1, d1 = g["A_dense_s0/kps"], g["A_dense_s0/desc"], g["bow/k1"], g["bow/d1"]
    fv0 = tuple(g[f"bow/fv0_{k}"] for k in ("node", "off", "idx"))
    fv1 = tuple(g[f"bow/fv1_{k}"] for k in ("node", "off", "idx"))
    v0, v1 = g["bow/valid0"], g["bow/valid1"]
    m, n = ORBmatcher(0.7, True).SearchByBoW(d0, v0, k0["angle"], fv0, d1, None, k1["angle"], fv1)
    assert n == int(g["bow_kf_f/n"]) and np.array_equal(m, g["bow_kf_f/match"])
    m21, n = ORBmatcher(0.75, True).SearchByBoW(d0, v0, k0["angle"], fv0, d1, v1, k1["angle"], fv1)
    m12 = np.full(len(d0), -1, np.int32)
    m12[m21[m21 >= 0]] = np.flatnonzero(m21 >= 0)
    assert n == int(g["bow_kf_kf/n"]) and np.array_equal(m12, g["bow_kf_kf/match12"])


def test_search_by_bow_rejects_bad_csr(mt):
    from orb_slam2_ssd_semantic_amd import OrbfeError, _ffi
    d = np.zeros((4, 32), np.uint8)
    a = np.zeros(4, np.float32)
    good = {1: [0, 1], 2: [2, 3]}
    dup = (np.array([1, 2], np.uint32), np.array([0, 2, 4], np.uint32), np.array([0, 1, 1, 3], np.uint32))
    with pytest.raises(OrbfeError) as ei:
        mt.SearchByBoW(d, None, a, good, d, None, a, dup)
    assert ei.value.status == _ffi.ORBFE_ERR_ARG


def test_hamming_csr_parity(mt, oracle):
    rng = np.random.default_rng(4)
    q, t = make_desc(rng, 700), make_desc(rng, 1500)
    lens = rng.integers(0, 60, 700)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cand = rng.integers(0, 1500, int(off[-1])).astype(np.uint32)
    got = mt.HammingCSR(q, t, off, cand)
    ref = oracle.hamming_csr(q, t, off, cand)
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    # with the runner-up's owner (bestLevel2 bookkeeping of SearchByProjection), duplicate rows force distance ties
    t[rng.integers(0, 1500, 400)] = t[rng.integers(0, 1500, 400)]
    got = mt.HammingCSR2(q, t, off, cand)
    ref = oracle.hamming_csr2(q, t, off, cand)
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    # device-resident form
    import torch
    from orb_slam2_ssd_semantic_amd import _ffi
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    do, dc = torch.from_numpy(off.view(np.int32)).cuda(), torch.from_numpy(cand.view(np.int32)).cuda()
    outs = [torch.full((700,), -9, dtype=torch.int32, device="cuda") for _ in range(4)]
    rc = _ffi.lib().orbfe_hamming_csr_device(mt.handle, dq.data_ptr(), 700, dt.data_ptr(), do.data_ptr(), dc.data_ptr(),
                                             outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        assert np.array_equal(o.cpu().numpy(), r)


def test_batched_frame_pairs_device(oracle):
    """orbfe_match_bf_frames_device (bench.py's extract+match step) on device-resident extractor output."""
    import ctypes as C
    import torch
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher, KP_DTYPE, _ffi
    B, cap = 5, 1088
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    frames = np.stack([synth_frame(300 + i) for i in range(B)])
    dg = torch.from_numpy(frames).cuda()
    dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
    dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    dn = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e.extract_batch_device(dg.data_ptr(), B, 640, 480, 640, 640 * 480, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), st)
    qf = torch.arange(1, B, dtype=torch.int32, device="cuda")
    tf = torch.arange(0, B - 1, dtype=torch.int32, device="cuda")
    dm = torch.full((B - 1, cap), -7, dtype=torch.int32, device="cuda")
    dnm = torch.zeros(B - 1, dtype=torch.int32, device="cuda")
    m = ORBmatcher(0.9, True)
    rc = _ffi.lib().orbfe_match_bf_frames_device(m.handle, dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), cap, qf.data_ptr(),
                                                 tf.data_ptr(), B - 1, 0.9, 100, 1, dm.data_ptr(), dnm.data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    n = dn.cpu().numpy()
    kps = dk.cpu().numpy()
    desc = dd.cpu().numpy()
    for p in range(B - 1):
        kq = kps[p + 1, :n[p + 1]].copy().view(KP_DTYPE).reshape(-1)
        kt = kps[p, :n[p]].copy().view(KP_DTYPE).reshape(-1)
        ref = oracle.match_bf(desc[p + 1, :n[p + 1]], desc[p, :n[p]], kq["angle"], kt["angle"], 0.9, 100, True)
        got = dm[p].cpu().numpy()
        assert np.array_equal(got[:n[p + 1]], ref[0]) and (got[n[p + 1]:] == -1).all()
        assert int(dnm[p]) == ref[3]


@pytest.mark.parametrize("seed,nq,nt", [(0, 1000, 1004), (1, 37, 2100), (2, 1, 5), (3, 300, 1), (4, 257, 129), (5, 0, 10),
                                        (6, 10, 0), (7, 4000, 4030)])
def test_popcount_all_pairs_kernel_equals_mfma_kernel_and_oracle(oracle, seed, nq, nt):
    """The xor / popcount all-pairs kernel the north star names (kept for the A/B against the matrix-core kernel) returns
    the same matches, best and second distances, on host buffers and in the batched device form."""
    import torch
    from orb_slam2_ssd_semantic_amd import ORBmatcher, _ffi
    rng = np.random.default_rng(900 + seed)
    t = make_desc(rng, nt)
    q = make_desc(rng, nq, base=t[rng.integers(0, nt, nq)], flips=40) if nt and nq else make_desc(rng, nq)
    if nt > 8:
        t[rng.integers(0, nt, nt // 5)] = t[rng.integers(0, nt, nt // 5)]   # duplicate rows: lowest index wins, second == best
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    ref = oracle.match_bf(q, t, qa, ta, 0.9, 100, True)
    outs = []
    for kern in (0, 1):
        m = ORBmatcher(0.9, True)
        m.set_bf_kernel(kern)
        got = m.MatchBruteForce(q, t, qa, ta, 100)
        assert all(np.array_equal(g, r) for g, r in zip(got[:3], ref[:3])) and got[3] == ref[3], kern
        outs.append(got)
    if nq and nt:   # batched device form: two "frames" (q, t) padded to cap
        cap = max(nq, nt) + 5
        desc = np.zeros((2, cap, 32), np.uint8)
        desc[0, :nq], desc[1, :nt] = q, t
        kps = np.zeros((2, cap, 7), np.float32)
        kps[0, :nq, 3], kps[1, :nt, 3] = qa, ta
        dd, dk = torch.from_numpy(desc).cuda(), torch.from_numpy(kps).cuda()
        dn = torch.tensor([nq, nt], dtype=torch.int32, device="cuda")
        qf, tf = torch.tensor([0], dtype=torch.int32, device="cuda"), torch.tensor([1], dtype=torch.int32, device="cuda")
        for kern in (0, 1):
            m = ORBmatcher(0.9, True)
            m.set_bf_kernel(kern)
            dm = torch.full((1, cap), -7, dtype=torch.int32, device="cuda")
            dnm = torch.zeros(1, dtype=torch.int32, device="cuda")
            rc = _ffi.lib().orbfe_match_bf_frames_device(m.handle, dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), cap, qf.data_ptr(),
                                                         tf.data_ptr(), 1, 0.9, 100, 1, dm.data_ptr(), dnm.data_ptr(), None)
            assert rc == 0
            torch.cuda.synchronize()
            got = dm[0].cpu().numpy()
            assert np.array_equal(got[:nq], ref[0]) and (got[nq:] == -1).all() and int(dnm[0]) == ref[3], kern
