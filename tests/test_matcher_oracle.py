"""Matcher oracle vs plain-Python twins.  CPU only."""
import numpy as np
import pytest

import twins


def make_desc(rng, n, base=None, flips=0):
    if base is None:
        return rng.integers(0, 256, (n, 32), dtype=np.uint8)
    out = base.copy()
    for i in range(len(out)):
        bits = rng.choice(256, size=int(rng.integers(0, flips + 1)), replace=False)
        for b in bits:
            out[i, b // 8] ^= np.uint8(1 << (b % 8))
    return out


@pytest.mark.parametrize("seed,nq,nt,flips,ratio,th", [(0, 60, 70, 30, 0.9, 100), (1, 40, 40, 60, 0.6, 50),
                                                       (2, 1, 5, 10, 0.9, 100), (3, 30, 1, 10, 0.9, 100),
                                                       (4, 50, 50, 0, 0.9, 100), (5, 0, 10, 0, 0.9, 100),
                                                       (6, 10, 0, 0, 0.9, 100)])
def test_match_bf(oracle, seed, nq, nt, flips, ratio, th):
    rng = np.random.default_rng(seed)
    t = make_desc(rng, nt)
    if nt and nq:
        q = make_desc(rng, nq, base=t[rng.integers(0, nt, nq)], flips=flips)
    else:
        q = make_desc(rng, nq)
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = (rng.uniform(0, 360, nt)).astype(np.float32)
    for ori in (True, False):
        got = oracle.match_bf(q, t, qa, ta, ratio, th, ori)
        ref = twins.match_bf(q, t, qa, ta, ratio, th, ori)
        for g, r in zip(got[:3], ref[:3]):
            assert np.array_equal(g, r)
        assert got[3] == ref[3]


def test_match_bf_tie_lowest_index_and_duplicate_second(oracle):
    rng = np.random.default_rng(0)
    t = make_desc(rng, 6)
    t[4] = t[1]                      # exact duplicate rows: best == second, lowest index wins, ratio test fails
    q = t[[1]].copy()
    m, b, s, n = oracle.match_bf(q, t, None, None, 0.9, 100, False)
    assert b[0] == 0 and s[0] == 0 and m[0] == -1 and n == 0   # 0 < 0.9*0 is false
    t[4, 0] ^= 1                     # now second = 1 -> accepted, index 1
    m, b, s, n = oracle.match_bf(q, t, None, None, 0.9, 100, False)
    assert (m[0], b[0], s[0], n) == (1, 0, 1, 1)


def random_fv(rng, nfeat, nnodes, node_ids):
    assign = rng.integers(0, nnodes, nfeat)
    fv = {}
    for i in rng.permutation(nfeat):
        fv.setdefault(int(node_ids[assign[i]]), []).append(int(i))
    return fv


def to_csr(fv):
    from orb_slam2_ssd_semantic_amd.matcher import feature_vector_to_csr
    return feature_vector_to_csr(fv)


@pytest.mark.parametrize("seed,strict,with_valid_f", [(0, False, False), (1, True, True), (2, False, False),
                                                      (3, True, True), (4, False, True)])
def test_search_by_bow(oracle, seed, strict, with_valid_f):
    rng = np.random.default_rng(seed)
    nK, nF = 150, 160
    dF = make_desc(rng, nF)
    dK = make_desc(rng, nK, base=dF[rng.integers(0, nF, nK)], flips=40)
    vK = (rng.uniform(size=nK) < 0.8).astype(np.uint8)
    vF = (rng.uniform(size=nF) < 0.8).astype(np.uint8) if with_valid_f else None
    aK = rng.uniform(0, 360, nK).astype(np.float32)
    aF = np.mod(aK[rng.integers(0, nK, nF)] + rng.normal(0, 20, nF), 360).astype(np.float32)
    ids = np.sort(rng.choice(1000, 25, replace=False))
    fvK = random_fv(rng, nK, 20, ids[:20])          # partially overlapping node sets
    fvF = random_fv(rng, nF, 20, ids[5:])
    for ori in (True, False):
        got = oracle.search_by_bow(dK, vK, aK, to_csr(fvK), dF, vF, aF, to_csr(fvF), 0.7, 50, strict, ori)
        ref = twins.search_by_bow(dK, vK, aK, fvK, dF, vF, aF, fvF, 0.7, 50, strict, ori)
        assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]
    assert got[1] > 0


def test_search_by_bow_greedy_exclusion(oracle):
    # two KF features in one node both closest to the same F feature: the first claims it (:273-274),
    # the second must settle for its next-best candidate among the remaining F features
    f = np.zeros((3, 32), np.uint8)
    f[1, 0] = 0b00001111
    f[2, :4] = 0xFF
    k = np.zeros((2, 32), np.uint8)
    k[1, 0] = 0b00000001  # dist to f0 = 1, f1 = 3, f2 = 31
    fvK, fvF = {7: [0, 1]}, {7: [0, 1, 2]}
    m, n = oracle.search_by_bow(k, None, np.zeros(2, np.float32), to_csr(fvK), f, None, np.zeros(3, np.float32),
                                to_csr(fvF), 0.9, 50, False, False)
    assert m.tolist() == [0, 1, -1] and n == 2
    mt, nt = twins.search_by_bow(k, None, np.zeros(2), fvK, f, None, np.zeros(3), fvF, 0.9, 50, False, False)
    assert mt.tolist() == m.tolist() and nt == n


def test_hamming_csr(oracle):
    rng = np.random.default_rng(4)
    q, t = make_desc(rng, 40), make_desc(rng, 90)
    lens = rng.integers(0, 12, 40)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cand = rng.integers(0, 90, int(off[-1])).astype(np.uint32)
    bi, b, s = oracle.hamming_csr(q, t, off, cand)
    D = twins.hamming_matrix(q, t)
    for i in range(40):
        b1, b2, ix = 256, 256, -1
        for c in cand[off[i]:off[i + 1]]:
            d = int(D[i, c])
            if d < b1:
                b2, b1, ix = b1, d, int(c)
            elif d < b2:
                b2 = d
        assert (bi[i], b[i], s[i]) == (ix, b1, b2)


def test_hamming_csr2_second_owner(oracle):
    """second_idx = the candidate that last set bestDist2 in the sequential idiom of src/ORBmatcher.cc:128-140, checked
    against a literal Python replay (ties, single candidates, empty lists)."""
    rng = np.random.default_rng(9)
    t = make_desc(rng, 300)
    t[rng.integers(0, 300, 120)] = t[rng.integers(0, 300, 120)]
    q = make_desc(rng, 200, base=t[rng.integers(0, 300, 200)], flips=12)
    lens = rng.integers(0, 25, 200)
    lens[:5] = [0, 1, 2, 1, 0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cand = rng.integers(0, 300, int(off[-1])).astype(np.uint32)
    bi, b, s, si = oracle.hamming_csr2(q, t, off, cand)
    for i in range(200):
        b1, b2, i1, i2 = 256, 256, -1, -1
        for c in cand[off[i]:off[i + 1]]:
            d = int(np.unpackbits(q[i] ^ t[c]).sum())
            if d < b1:
                b2, i2, b1, i1 = b1, i1, d, int(c)
            elif d < b2:
                b2, i2 = d, int(c)
        assert (bi[i], b[i], s[i], si[i]) == (i1, b1, b2, i2), i
    assert np.array_equal(oracle.hamming_csr(q, t, off, cand)[2], s)
