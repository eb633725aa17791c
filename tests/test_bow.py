"""DBoW2 TemplatedVocabulary::transform as called by Frame::ComputeBoW (reference src/Frame.cc:553).

DBoW2 is not vendored by the reference, so the oracle restates the published algorithm; it is checked here against
an independently written twin on random vocabulary trees.  GPU: the HIP path through the C-ABI against the oracle --
word / node ids exact, BowVector doubles bit-exact (same summation order), FeatureVector CSR exact."""
import numpy as np
import pytest


def make_voc(seed, k, L, zero_frac=0.1):
    """random k-ary tree of depth L in DBoW2's creation order (breadth first, children after parents)"""
    rng = np.random.default_rng(seed)
    child_off, child_idx, level = [0], [], [0]
    nodes = 1
    i = 0
    while i < nodes:
        if level[i] < L:
            kk = k if rng.random() < 0.8 else max(2, k - 2)   # ragged fan-out
            child_idx.extend(range(nodes, nodes + kk))
            level.extend([level[i] + 1] * kk)
            nodes += kk
        child_off.append(len(child_idx))
        i += 1
    child_off = np.asarray(child_off, np.uint32)
    child_idx = np.asarray(child_idx, np.uint32)
    node_desc = rng.integers(0, 256, (nodes, 32), dtype=np.uint8)
    leaf = child_off[1:] == child_off[:-1]
    word_id = np.zeros(nodes, np.uint32)
    word_id[leaf] = np.arange(leaf.sum(), dtype=np.uint32)
    weight = rng.uniform(0.1, 9.0, nodes)
    weight[rng.random(nodes) < zero_frac] = 0.0
    # make some sibling descriptors identical so that "first child wins" matters
    for p in rng.integers(0, nodes, 10):
        c0, c1 = child_off[p], child_off[p + 1]
        if c1 - c0 >= 2:
            node_desc[child_idx[c0 + 1]] = node_desc[child_idx[c0]]
    return dict(child_off=child_off, child_idx=child_idx, node_desc=node_desc, word_id=word_id, weight=weight, L=L)


def twin(voc, desc, levelsup):
    bits = np.unpackbits(voc["node_desc"], axis=1).astype(np.int16)
    fb = np.unpackbits(np.ascontiguousarray(desc, np.uint8).reshape(-1, 32), axis=1).astype(np.int16)
    co, ci = voc["child_off"], voc["child_idx"]
    nid_level = voc["L"] - levelsup
    bow, fv = {}, {}
    for i in range(len(fb)):
        fin, nid, lev = 0, 0, 0
        while True:
            lev += 1
            ch = ci[co[fin]:co[fin + 1]]
            d = np.abs(bits[ch] - fb[i]).sum(1)
            fin = int(ch[int(np.argmin(d))])   # first minimum
            if lev == nid_level:
                nid = fin
            if co[fin + 1] == co[fin]:
                break
        w = float(voc["weight"][fin])
        if w > 0:
            wid = int(voc["word_id"][fin])
            bow[wid] = bow.get(wid, 0.0) + w
            fv.setdefault(nid, []).append(i)
    ids = sorted(bow)
    norm = 0.0
    for k in ids:
        norm += abs(bow[k])
    vals = [bow[k] / norm if norm > 0 else bow[k] for k in ids]
    return ids, vals, fv


@pytest.mark.parametrize("seed,k,L,n,levelsup", [(0, 4, 4, 300, 2), (1, 10, 3, 500, 1), (2, 3, 6, 200, 4), (3, 5, 2, 50, 4),
                                                  (4, 6, 3, 0, 1)])
def test_oracle_bow_vs_twin(oracle, seed, k, L, n, levelsup):
    voc = make_voc(seed, k, L)
    desc = np.random.default_rng(100 + seed).integers(0, 256, (n, 32), dtype=np.uint8)
    r = oracle.bow_transform(voc, desc, levelsup)
    ids, vals, fv = twin(voc, desc, levelsup)
    assert r["bow_id"].tolist() == ids
    assert r["bow_val"].tolist() == vals                     # doubles, same summation order
    assert r["fv_node"].tolist() == sorted(fv)
    for j, nd in enumerate(r["fv_node"]):
        assert r["fv_idx"][r["fv_off"][j]:r["fv_off"][j + 1]].tolist() == fv[int(nd)]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,k,L,n,levelsup", [(0, 4, 4, 300, 2), (1, 10, 3, 1004, 1), (2, 3, 6, 200, 4), (3, 5, 2, 50, 4),
                                                  (4, 6, 3, 0, 1), (5, 10, 4, 4100, 2), (6, 8, 3, 1, 1)])
def test_gpu_bow_parity(oracle, seed, k, L, n, levelsup):
    from orb_slam2_ssd_semantic_amd import ORBmatcher, ORBVocabulary
    voc = make_voc(seed, k, L)
    desc = np.random.default_rng(100 + seed).integers(0, 256, (n, 32), dtype=np.uint8)
    r = oracle.bow_transform(voc, desc, levelsup)
    mt = ORBmatcher(0.9, True)
    v = ORBVocabulary(mt, **voc)
    (bid, bval), (fvn, fvo, fvi), (fw, fn, fwt) = v.transform(desc, levelsup, per_feature=True)
    assert np.array_equal(fw, r["word"]) and np.array_equal(fn, r["node"]) and np.array_equal(fwt, r["weight"])
    assert np.array_equal(bid, r["bow_id"])
    assert np.array_equal(bval.view(np.uint64), r["bow_val"].view(np.uint64))
    assert np.array_equal(fvn, r["fv_node"]) and np.array_equal(fvo, r["fv_off"]) and np.array_equal(fvi, r["fv_idx"])


@pytest.mark.gpu
def test_gpu_bow_feeds_search_by_bow(oracle):
    """extract -> BoW transform -> SearchByBoW, all through the C-ABI, against the same chain of oracle calls"""
    from orb_slam2_ssd_semantic_amd import ORBmatcher, ORBVocabulary
    rng = np.random.default_rng(9)
    voc = make_voc(9, 10, 3, zero_frac=0.02)
    dk = rng.integers(0, 256, (600, 32), dtype=np.uint8)
    df = dk[rng.permutation(600)[:500]].copy()
    flips = rng.integers(0, 256, (500, 12))
    for i in range(500):
        for b in flips[i]:
            df[i, b >> 3] ^= 1 << (b & 7)
    mt = ORBmatcher(0.75, True)
    v = ORBVocabulary(mt, **voc)
    _, fvk = v.transform(dk, 1)
    _, fvf = v.transform(df, 1)
    rk, rf = oracle.bow_transform(voc, dk, 1), oracle.bow_transform(voc, df, 1)
    ak = rng.uniform(0, 360, 600).astype(np.float32)
    af = rng.uniform(0, 360, 500).astype(np.float32)
    valid = (rng.random(600) < 0.9).astype(np.uint8)
    got = mt.SearchByBoW(dk, valid, ak, fvk, df, None, af, fvf)
    ref = oracle.search_by_bow(dk, valid, ak, (rk["fv_node"], rk["fv_off"], rk["fv_idx"]), df, None, af,
                               (rf["fv_node"], rf["fv_off"], rf["fv_idx"]), 0.75, 50, False, True)
    assert np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[1] > 50


def test_bow_bad_args_cpu():
    from orb_slam2_ssd_semantic_amd import _ffi
    L = _ffi.lib()
    assert L.orbfe_vocabulary_create(0, 0, None, None, None, None, None, 1, None) == _ffi.ORBFE_ERR_ARG
    assert L.orbfe_bow_transform(None, None, None, 0, 4, *([None] * 10)) == _ffi.ORBFE_ERR_ARG
