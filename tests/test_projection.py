"""The projection-gated searches of the per-frame tracker:
  ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)   src/ORBmatcher.cc:1578-1724
  ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)       src/ORBmatcher.cc:63-157
  (+ the perfect/ overload that also returns the 2-D point pairs, perfect/src/ORBmatcher.cc:1727-1911)
GPU: orbfe_search_by_projection (HIP) == the oracle's sequential core on the same queries, bit for bit, including the
"slot taken by an earlier query" dependency (relaxation on the device, a plain loop in the oracle); and the product's shim
members, called through the reference's own class on mock Frames, == the reference's compiled bodies.  The oracle itself is
pinned to those bodies on the CPU (tests/test_ref_pin.py)."""
import numpy as np
import pytest

import proj_cases as PC
from oracle import oracle_ffi as O
from oracle import ref_ffi as R


def _core_both(mat, cur, q, qdesc, th, nnratio, rule):
    ci = PC.core_inputs(cur)
    om, ob, os_ = O.search_by_projection(queries=q, qdesc=qdesc, th=th, nnratio=nnratio, ratio_rule=rule, **ci)
    gm, gb, gs = mat.SearchByProjectionCore(queries=q, qdesc=qdesc, th=th, nnratio=nnratio, ratio_rule=rule, **ci)
    return (om, ob, os_), (gm, gb, gs)


@pytest.fixture(scope="module", params=[0, 1], ids=["one-launch", "four-kernel"])
def mat(request):
    """both forms of the device core: k_proj_fused (default; falls back by itself when a query has > 512 candidates -- the tight
    cluster of test_core_long_dependency_chains does that) and count -> scan -> fill -> resolve"""
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    m = ORBmatcher(0.9, True)
    m.set_projection_kernel(request.param)
    return m


@pytest.mark.gpu
def test_core_equals_oracle_on_tracker_shaped_cases(mat):
    nq_tot = nm_tot = 0
    for seed in range(120):
        rng = np.random.default_rng(11_000 + seed)
        if seed % 2 == 0:
            nC, nL = int(rng.choice([1, 30, 300, 1000, 2000])), int(rng.choice([1, 40, 400, 1000, 2500]))
            mono = seed % 8 == 6
            cur, last = PC.last_frame_case(rng, nC, nL, ["small", "forward", "backward"][(seed // 2) % 3], stereo=not mono)
            q, valid = O.proj_queries_last_frame(cur["Tcw"], last["Tcw"], cur["K"], cur["bounds"], cur["scale_factors"], last["has_mp"],
                                                 last["outlier"], last["world_pos"], last["octave"], last["obs_gt0"],
                                                 float(rng.choice([7, 15, 30])), mono)
            sel = valid.astype(bool)
            q, qd, th, nn, rule = q[sel], last["mpdesc"][sel], 100, 0.0, 0
        else:
            nF, nmp = int(rng.choice([1, 30, 300, 1000, 2000])), int(rng.choice([1, 40, 400, 1500, 3000]))
            cur, mps = PC.local_map_case(rng, nF, nmp)
            q, valid = O.proj_queries_local_map(cur["scale_factors"], mps["in_view"], mps["bad"], mps["scale_level"], mps["view_cos"],
                                                mps["proj_xyr"], mps["obs_gt0"], float(rng.choice([1, 3, 5])))
            sel = valid.astype(bool)
            q, qd, th, nn, rule = q[sel], mps["mpdesc"][sel], int(rng.choice([100, 50])), float(rng.choice([0.8, 0.7])), 1
        (om, ob, os_), (gm, gb, gs) = _core_both(mat, cur, q, qd, th, nn, rule)
        assert np.array_equal(gm, om), (seed, np.nonzero(gm != om)[0][:5])
        assert np.array_equal(gb, ob) and np.array_equal(gs, os_), seed
        nq_tot += len(q)
        nm_tot += int((om >= 0).sum())
    assert nq_tot > 30_000 and nm_tot > 8_000


@pytest.mark.gpu
def test_core_reprojection_error_gate_equals_oracle(mat):
    """ORBFE_PROJ_CHI2_GATE (Fuse's candidate gate, src/ORBmatcher.cc:1112-1139) in the device core against the oracle's loop:
    queries that land near features with sub-pixel to several-pixel offsets, stereo and monocular keypoints, all octaves,
    mixed with queries without the flag.  The reference-side anchor of the flag is the Fuse shim test below (the shim's Fuse
    runs through this path and equals the reference's compiled body)."""
    tot = 0
    for seed in range(40):
        rng = np.random.default_rng(33_000 + seed)
        nF, nq = int(rng.choice([1, 30, 300, 1000, 2000])), int(rng.choice([1, 40, 400, 2000]))
        cur = PC.current_frame(rng, nF)
        sf = cur["scale_factors"]
        is2 = (1.0 / (sf * sf)).astype(np.float32)
        tgt = rng.integers(0, nF, nq)
        q = np.zeros(nq, O.PROJ_QUERY_DTYPE)
        off = rng.normal(0, 1, (nq, 2)) * rng.choice([0.3, 1.5, 2.5, 6.0], (nq, 1)) * sf[cur["octave"][tgt]][:, None]
        q["u"], q["v"] = (cur["xy"][tgt, 0] + off[:, 0]).astype(np.float32), (cur["xy"][tgt, 1] + off[:, 1]).astype(np.float32)
        lvl = cur["octave"][tgt] + rng.choice([0, 0, 1], nq)
        q["r"] = (rng.choice([3.0, 5.0], nq) * sf[np.clip(lvl, 0, 7)]).astype(np.float32)
        q["min_level"], q["max_level"] = lvl - 1, lvl
        q["ur"] = (np.where(cur["uRight"][tgt] >= 0, cur["uRight"][tgt], q["u"] - 20) + rng.normal(0, 1.5, nq)).astype(np.float32)
        q["flags"] = np.where(rng.random(nq) < 0.85, 4, 0)
        qd = PC.noisy_copy(rng, cur["desc"][tgt], 60)
        ci = PC.core_inputs(cur)
        ci["blocked"] = None
        om = O.search_by_projection(queries=q, qdesc=qd, th=50, nnratio=0.0, ratio_rule=0, inv_level_sigma2=is2, **ci)
        gm = mat.SearchByProjectionCore(queries=q, qdesc=qd, th=50, nnratio=0.0, ratio_rule=0, inv_level_sigma2=is2, **ci)
        nogate = O.search_by_projection(queries=q, qdesc=qd, th=50, nnratio=0.0, ratio_rule=0, **ci)
        assert all(np.array_equal(a, b) for a, b in zip(om, gm)), seed
        tot += int((om[0] != nogate[0]).sum())
    assert tot > 200   # the gate changed that many decisions


@pytest.mark.gpu
def test_window_distances_and_csr_all_equal_oracle(mat):
    """orbfe_window_distances: per query the candidates of Frame::GetFeaturesInArea in the reference's order (oracle
    orc_features_in_area, pinned to the sliced reference body) with their Hamming distances; orbfe_hamming_csr_all: every
    distance of caller-built lists.  Windows of SearchForInitialization's size (100 px), tracker-sized ones, empty ones,
    out-of-image centres, level filters on and off."""
    ncand = 0
    for seed in range(25):
        rng = np.random.default_rng(35_000 + seed)
        nF, nq = int(rng.choice([1, 30, 300, 2000])), int(rng.choice([1, 40, 400]))
        cur = PC.current_frame(rng, nF)
        goff, gidx = PC.frame_grid(cur)
        minx, _, miny, _ = cur["bounds"]
        q = np.zeros(nq, O.PROJ_QUERY_DTYPE)
        q["u"] = rng.uniform(-40, 680, nq).astype(np.float32)
        q["v"] = rng.uniform(-40, 520, nq).astype(np.float32)
        q["r"] = rng.choice([3.0, 15.0, 100.0], nq).astype(np.float32)
        lv = rng.integers(0, 8, nq)
        mode = rng.integers(0, 3, nq)   # 0: no level filter, 1: one level, 2: a window of levels
        q["min_level"] = np.where(mode == 0, -1, np.where(mode == 1, lv, lv - 1))
        q["max_level"] = np.where(mode == 0, -1, lv)
        qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        off, cand, dist = mat.WindowDistances(cur["desc"], cur["xy"], cur["octave"], (goff, gidx), (minx, miny, cur["gw_inv"], cur["gh_inv"]),
                                              q, qd, cap=64)   # cap too small on purpose for the larger cases: the retry path
        ref_c, ref_off = [], [0]
        for i in range(nq):
            c = O.features_in_area(cur["xy"], cur["octave"], goff, gidx, minx, miny, cur["gw_inv"], cur["gh_inv"], float(q["u"][i]),
                                   float(q["v"][i]), float(q["r"][i]), int(q["min_level"][i]), int(q["max_level"][i]))
            ref_c.append(c)
            ref_off.append(ref_off[-1] + len(c))
        ref_c = np.concatenate(ref_c) if ref_c else np.zeros(0, np.uint32)
        assert np.array_equal(off, np.array(ref_off, np.uint32)) and np.array_equal(cand, ref_c), seed
        qi = np.repeat(np.arange(nq), np.diff(ref_off))
        ref_d = np.unpackbits(qd[qi] ^ cur["desc"][ref_c], axis=1).sum(1) if len(ref_c) else np.zeros(0, np.int64)
        assert np.array_equal(dist.astype(np.int64), ref_d), seed
        d2 = mat.HammingCSRAll(qd, cur["desc"], np.array(ref_off, np.uint32), ref_c)
        assert np.array_equal(d2.astype(np.int64), ref_d), seed
        ncand += len(ref_c)
    assert ncand > 30_000


@pytest.mark.gpu
def test_core_long_dependency_chains(mat):
    """Every query wants the same few slots: query i can only settle after all earlier claiming queries have -- the device's
    relaxation needs as many rounds as the chain is long and must still land on the sequential result."""
    for seed, (nF, nq) in enumerate([(64, 64), (300, 300), (40, 500), (1000, 1200)]):
        rng = np.random.default_rng(500 + seed)
        cur = PC.current_frame(rng, nF, stereo=False, dense_states=False)
        cur["xy"][:] = (np.array([320.0, 240.0]) + rng.normal(0, 3.0, (nF, 2))).astype(np.float32)   # one tight cluster
        cur["octave"][:] = 2
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        bits = np.tile(np.unpackbits(base), (nF, 1))
        for i in range(nF):   # slot i is at distance ~i / 4 from the common query descriptor: a strict preference order
            bits[i, rng.choice(256, min(i // 4, 200), replace=False)] ^= 1
        cur["desc"] = np.packbits(bits, axis=1)
        q = np.zeros(nq, O.PROJ_QUERY_DTYPE)
        q["u"], q["v"], q["r"] = 320.0, 240.0, 40.0
        q["min_level"], q["max_level"] = 1, 3
        q["flags"] = np.where(rng.random(nq) < 0.9, 1, 0) | 2
        qd = np.tile(base, (nq, 1))
        for th, nn, rule in ((100, 0.0, 0), (255, 0.9, 1)):
            (om, ob, os_), (gm, gb, gs) = _core_both(mat, cur, q, qd, th, nn, rule)
            assert np.array_equal(gm, om) and np.array_equal(gb, ob) and np.array_equal(gs, os_), (seed, th)
            assert len(set(om[(om >= 0) & ((q["flags"] & 1) == 1)].tolist())) == int(((om >= 0) & ((q["flags"] & 1) == 1)).sum())


@pytest.mark.gpu
def test_core_edge_cases(mat):
    rng = np.random.default_rng(3)
    cur = PC.current_frame(rng, 50)
    ci = PC.core_inputs(cur)
    q0 = np.zeros(0, O.PROJ_QUERY_DTYPE)
    m, b, s = mat.SearchByProjectionCore(queries=q0, qdesc=np.zeros((0, 32), np.uint8), th=100, nnratio=0.8, ratio_rule=1, **ci)
    assert len(m) == 0
    # queries far outside the grid, zero radius, every slot blocked
    q = np.zeros(5, O.PROJ_QUERY_DTYPE)
    q["u"], q["v"], q["r"] = [-500, 5000, 320, 320, 320], [-500, 5000, 240, 240, 240], [10, 10, 0, 1000, 1000]
    q["min_level"], q["max_level"], q["flags"] = -1, -1, 3
    qd = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    for blocked in (ci["blocked"], np.ones(50, np.uint8), None):
        ci2 = dict(ci, blocked=blocked)
        om, ob, os_ = O.search_by_projection(queries=q, qdesc=qd, th=255, nnratio=0.8, ratio_rule=0, **ci2)
        gm, gb, gs = mat.SearchByProjectionCore(queries=q, qdesc=qd, th=255, nnratio=0.8, ratio_rule=0, **ci2)
        assert np.array_equal(gm, om) and np.array_equal(gb, ob) and np.array_equal(gs, os_)
    with pytest.raises(Exception):   # th >= 256 would accept "no candidate"
        mat.SearchByProjectionCore(queries=q, qdesc=qd, th=256, nnratio=0.8, ratio_rule=0, **ci)
    # an empty frame
    cur0 = PC.current_frame(rng, 0)
    (om, _, _), (gm, _, _) = _core_both(mat, cur0, q, qd, 100, 0.8, 1)
    assert np.array_equal(gm, om) and (gm == -1).all()


needs_shim = pytest.mark.skipif(not R.shim_available(), reason="oracle/_ref/libshim_ref.so not built and /root/reference absent")


@needs_shim
def test_shims_export_the_projection_members():
    for L in (R.shim_lib(), R.shim_perfect_lib()):
        assert L.shim_search_by_projection_last_frame and L.shim_search_by_projection_local_map


@needs_shim
@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_shim_search_by_projection_equals_reference_bodies(block):
    """reference class -> shim member (host gating on cv::Mat, one orbfe_search_by_projection call, replay) == the reference's
    own compiled body on the same mock Frames: CurrentFrame.mvpMapPoints slot by slot + the return value; 240 cases over both
    overloads, mono / stereo, forward / backward / small motion, rotation check on / off."""
    total = 0
    for it in range(30):
        seed = block * 30 + it
        rng = np.random.default_rng(13_000 + seed)
        nC, nL = int(rng.choice([0, 1, 30, 300, 1000])), int(rng.choice([0, 1, 40, 400, 1000]))
        mono = seed % 4 == 3
        cur, last = PC.last_frame_case(rng, nC, nL, ["small", "forward", "backward"][seed % 3], stereo=not mono)
        th, ori = float(rng.choice([7, 15, 15, 30])), bool(seed % 5)
        ra, rn = R.search_by_projection_last_frame(cur, last, th, mono, check_ori=ori)
        sa, sn = R.search_by_projection_last_frame(cur, last, th, mono, check_ori=ori, shim=True)
        assert sn == rn and np.array_equal(sa, ra), ("last frame", seed, nC, nL, mono, th)
        cur2, mps = PC.local_map_case(rng, nC, nL + 200)
        th2, nn = float(rng.choice([1, 3, 5])), float(rng.choice([0.8, 0.7, 0.9]))
        ra2, rn2 = R.search_by_projection_local_map(cur2, mps, th2, nn)
        sa2, sn2 = R.search_by_projection_local_map(cur2, mps, th2, nn, shim=True)
        assert sn2 == rn2 and np.array_equal(sa2, ra2), ("local map", seed, nC, nL, th2, nn)
        total += rn + rn2
    assert total > 1000


@needs_shim
@pytest.mark.gpu
def test_shim_perfect_overload_with_point_pairs_equals_reference_body():
    """M9 through perfect/'s class: the shim built with -DORBFE_SHIM_PERFECT against perfect/include/ORBmatcher.h"""
    for seed in range(40):
        rng = np.random.default_rng(15_000 + seed)
        nC, nL = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1000]))
        mono = seed % 4 == 3
        cur, last = PC.last_frame_case(rng, nC, nL, ["small", "forward", "backward"][seed % 3], stereo=not mono)
        th = float(rng.choice([7, 15, 30]))
        ra, rn, rpl, rpc = R.search_by_projection_last_frame(cur, last, th, mono, perfect=True, points=True)
        sa, sn, spl, spc = R.search_by_projection_last_frame(cur, last, th, mono, perfect=True, points=True, shim=True)
        assert sn == rn and np.array_equal(sa, ra) and np.array_equal(spl, rpl) and np.array_equal(spc, rpc), seed
        sa2, sn2 = R.search_by_projection_last_frame(cur, last, th, mono, perfect=True, shim=True)
        assert sn2 == rn and np.array_equal(sa2, ra), seed


@needs_shim
@pytest.mark.gpu
def test_shim_fuse_equals_reference_body():
    """ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:1031-1182, LocalMapping::SearchInNeighbors)
    through the reference's class: shim (all gates on the host, ONE orbfe_hamming_csr call, decisions replayed in order) ==
    the reference's compiled body -- the keyframe's MapPoint per feature, which points were replaced by which, the count."""
    fused = 0
    for seed in range(80):
        rng = np.random.default_rng(17_000 + seed)
        nKF, nmp = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1500]))
        kf, mps = PC.fuse_case(rng, nKF, nmp)
        th = float(rng.choice([3.0, 3.0, 5.0]))
        r = R.fuse(kf, mps, th)
        s = R.fuse(kf, mps, th, shim=True)
        assert s[3] == r[3] and all(np.array_equal(a, b) for a, b in zip(s[:3], r[:3])), (seed, nKF, nmp, r[3], s[3])
        fused += r[3]
    assert fused > 2000


@needs_shim
@pytest.mark.gpu
def test_shim_keyframe_side_search_by_projection_equals_reference_bodies():
    """The two remaining SearchByProjection overloads through the reference's class: (Frame&, KeyFrame*, sAlreadyFound, th,
    ORBdist) (src/ORBmatcher.cc:1757-1867, Tracking::Relocalization: any MapPoint takes a slot, rotation histogram on the
    keyframe's angles) and (KeyFrame*, Scw, vpPoints, vpMatched, th) (:378-470, LoopClosing: a Sim3-projected search on a
    KeyFrame's grid) -- both on the same device core as the per-frame forms."""
    n1 = n2 = 0
    for seed in range(60):
        rng = np.random.default_rng(19_000 + seed)
        nC, nK = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1000]))
        cur, kfp = PC.frame_kf_case(rng, nC, nK)
        th, od, ori = float(rng.choice([10, 3, 15])), int(rng.choice([100, 64])), bool(seed % 4)
        ra, rn = R.search_by_projection_frame_kf(cur, kfp, th, od, ori)
        sa, sn = R.search_by_projection_frame_kf(cur, kfp, th, od, ori, shim=True)
        assert sn == rn and np.array_equal(sa, ra), ("frame/kf", seed, nC, nK, rn, sn)
        n1 += rn
        kf, Scw, pts, mi = PC.kf_sim3_case(rng, nC, nK + 100)
        th2 = int(rng.choice([10, 4]))
        rm, rn2 = R.search_by_projection_kf_sim3(kf, Scw, pts, mi, th2)
        sm, sn2 = R.search_by_projection_kf_sim3(kf, Scw, pts, mi, th2, shim=True)
        assert sn2 == rn2 and np.array_equal(sm, rm), ("kf/sim3", seed, nC, nK, rn2, sn2)
        n2 += rn2
    assert n1 > 1000 and n2 > 1000


@pytest.mark.gpu
def test_triangulation_core_equals_oracle(mat):
    """orbfe_search_for_triangulation (HIP: one thread per keyframe-1 feature, Hamming + epipolar gate in the reference's float
    operation order) == the oracle's loop, match12 element by element"""
    tot = 0
    for seed in range(80):
        rng = np.random.default_rng(23_000 + seed)
        n1, n2 = int(rng.choice([1, 30, 300, 1000, 2000])), int(rng.choice([1, 40, 400, 1000, 2000]))
        k1, k2, F = PC.triangulation_case(rng, n1, n2, int(rng.choice([1, 10, 100, 400])))
        a, b, ex, ey = PC.tri_core_inputs(k1, k2, seed % 3 == 0)
        om = O.search_for_triangulation(a, b, F, ex, ey, 50)
        gm = mat.SearchForTriangulationCore(a, b, F, ex, ey, 50)
        assert np.array_equal(gm, om), (seed, n1, n2, np.nonzero(gm != om)[0][:5])
        tot += int((om >= 0).sum())
    assert tot > 3000


@needs_shim
@pytest.mark.gpu
def test_shim_search_for_triangulation_equals_reference_body():
    """ORBmatcher::SearchForTriangulation through the reference's class: shim (epipole and flags on the host, ONE device call,
    rotation histogram replayed) == the reference's compiled body: vMatchedPairs and the return value"""
    tot = 0
    for seed in range(60):
        rng = np.random.default_rng(25_000 + seed)
        n1, n2 = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1000]))
        k1, k2, F = PC.triangulation_case(rng, n1, n2, int(rng.choice([1, 10, 100])))
        only, ori = seed % 3 == 0, bool(seed % 4)
        rp, rn = R.search_for_triangulation(k1, k2, F, only, ori)
        sp, sn = R.search_for_triangulation(k1, k2, F, only, ori, shim=True)
        assert sn == rn and np.array_equal(sp, rp), (seed, n1, n2, rn, sn)
        tot += rn
    assert tot > 800


@needs_shim
@pytest.mark.gpu
def test_shim_fuse_sim3_equals_reference_body():
    """ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:1198-1299, LoopClosing::SearchAndFuse)
    through the reference's class: the keyframe's MapPoint per feature, vpReplacePoint entry by entry, the count"""
    fused = 0
    for seed in range(60):
        rng = np.random.default_rng(27_000 + seed)
        nKF, nmp = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1500]))
        kf, Scw, pts, _ = PC.kf_sim3_case(rng, nKF, nmp)
        mps = dict(pts, null=(rng.random(nmp) < 0.03).astype(np.uint8))
        th = float(rng.choice([4.0, 10.0]))
        r = R.fuse_sim3(kf, Scw, mps, th)
        s = R.fuse_sim3(kf, Scw, mps, th, shim=True)
        assert s[2] == r[2] and np.array_equal(s[0], r[0]) and np.array_equal(s[1], r[1]), (seed, nKF, nmp, r[2], s[2])
        fused += r[2]
    assert fused > 1500


@needs_shim
@pytest.mark.gpu
def test_shim_search_by_sim3_equals_reference_body():
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1334-1548, LoopClosing::ComputeSim3) through the reference's class: both
    directions in one device call, then the mutual-agreement rule; vpMatches12 entry by entry and the count"""
    found = 0
    for seed in range(60):
        rng = np.random.default_rng(28_000 + seed)
        n1, n2 = int(rng.choice([1, 30, 300, 1000])), int(rng.choice([1, 40, 400, 1200]))
        k1, k2, s12, R12, t12, m_in = PC.sim3_pair_case(rng, n1, n2)
        th = float(rng.choice([7.5, 10.0]))
        r = R.search_by_sim3(k1, k2, s12, R12, t12, th, m_in)
        s = R.search_by_sim3(k1, k2, s12, R12, t12, th, m_in, shim=True)
        assert s[1] == r[1] and np.array_equal(s[0], r[0]), (seed, n1, n2, r[1], s[1])
        found += r[1]
    assert found > 800


@needs_shim
@pytest.mark.gpu
def test_shim_search_for_initialization_equals_reference_body():
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:523-651, Tracking::MonocularInitialization) through the
    reference's class: vnMatches12, the updated vbPrevMatched and the count; a second call continues from the first one's
    vbPrevMatched as the tracker does"""
    total = 0
    for seed in range(60):
        rng = np.random.default_rng(29_000 + seed)
        n1, n2 = int(rng.choice([1, 50, 500, 2000])), int(rng.choice([1, 60, 600, 2000]))
        f1, f2, prev = PC.initialization_case(rng, n1, n2)
        win = int(rng.choice([30, 100]))
        ori = bool(seed % 5)
        r = R.search_for_initialization(f1, f2, prev, win, 0.9, ori)
        s = R.search_for_initialization(f1, f2, prev, win, 0.9, ori, shim=True)
        assert s[2] == r[2] and np.array_equal(s[0], r[0]) and np.array_equal(s[1], r[1]), (seed, n1, n2, r[2], s[2])
        r2 = R.search_for_initialization(f1, f2, r[1], win, 0.9, ori)
        s2 = R.search_for_initialization(f1, f2, s[1], win, 0.9, ori, shim=True)
        assert s2[2] == r2[2] and np.array_equal(s2[0], r2[0]) and np.array_equal(s2[1], r2[1]), (seed, "second call")
        total += r[2]
    assert total > 1000


@needs_shim
def test_standalone_shim_is_the_whole_translation_unit():
    """libshim_full.so links with --no-undefined WITHOUT the reference's src/ORBmatcher.cc: constants, constructor, helpers and
    every public member come from shim/ORBmatcher_orbfe.cc (-DORBFE_SHIM_STANDALONE).  Host-side members here, no GPU."""
    L = R.shim_full_lib()
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    for i in range(0, 64, 2):
        assert R.descriptor_distance(d[i], d[i + 1], shim="full") == R.descriptor_distance(d[i], d[i + 1])
    import ctypes as C
    for it in range(300):   # ComputeThreeMaxima, restated in the shim, against the reference's (histograms with ties / empty bins)
        counts = rng.integers(0, [2, 5, 40, 400][it % 4], 30).astype(np.int32)
        if it % 7 == 0:
            counts[rng.integers(0, 30, 25)] = 0
        out = []
        for lib, pre in ((L, "shim_"), (R.lib(), "ref_")):
            i = [C.c_int(-7) for _ in range(3)]
            getattr(lib, pre + "three_maxima")(counts.ctypes.data, 30, C.byref(i[0]), C.byref(i[1]), C.byref(i[2]))
            out.append([v.value for v in i])
        assert out[0] == out[1], (counts, out)
    th = [np.zeros(1, np.int32) for _ in range(3)]
    tr = [np.zeros(1, np.int32) for _ in range(3)]
    L.shim_matcher_constants(*[a.ctypes.data for a in th])
    R.lib().ref_matcher_constants(*[a.ctypes.data for a in tr])
    assert [int(a[0]) for a in th] == [int(a[0]) for a in tr]


@needs_shim
@pytest.mark.gpu
def test_standalone_shim_every_member_equals_reference_bodies():
    """every public member of ORBmatcher through libshim_full.so (the shim alone, no reference ORBmatcher.cc behind it) against
    the reference's compiled bodies"""
    from test_ref_pin import _bow_case
    for seed in range(12):
        rng = np.random.default_rng(31_000 + seed)
        n1, n2 = int(rng.choice([30, 300, 1000])), int(rng.choice([40, 400, 1000]))
        mono = seed % 4 == 3
        cur, last = PC.last_frame_case(rng, n1, n2, ["small", "forward", "backward"][seed % 3], stereo=not mono)
        a, b = R.search_by_projection_last_frame(cur, last, 15.0, mono), R.search_by_projection_last_frame(cur, last, 15.0, mono, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("last frame", seed)
        cur2, mps = PC.local_map_case(rng, n1, n2 + 200)
        a, b = R.search_by_projection_local_map(cur2, mps, 3.0, 0.8), R.search_by_projection_local_map(cur2, mps, 3.0, 0.8, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("local map", seed)   # RadiusByViewingCos is the shim's own here
        cur3, kfp = PC.frame_kf_case(rng, n1, n2)
        a, b = R.search_by_projection_frame_kf(cur3, kfp, 10.0, 100, True), R.search_by_projection_frame_kf(cur3, kfp, 10.0, 100, True, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("frame/kf", seed)
        kf, Scw, pts, mi = PC.kf_sim3_case(rng, n1, n2 + 100)
        a, b = R.search_by_projection_kf_sim3(kf, Scw, pts, mi, 10), R.search_by_projection_kf_sim3(kf, Scw, pts, mi, 10, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("kf/sim3", seed)
        m2 = dict(pts, null=np.zeros(len(pts["bad"]), np.uint8))
        a, b = R.fuse_sim3(kf, Scw, m2, 4.0), R.fuse_sim3(kf, Scw, m2, 4.0, shim="full")
        assert b[2] == a[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), ("fuse sim3", seed)
        kf2, mps2 = PC.fuse_case(rng, n1, n2)
        a, b = R.fuse(kf2, mps2, 3.0), R.fuse(kf2, mps2, 3.0, shim="full")
        assert b[3] == a[3] and all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])), ("fuse", seed)
        k1, k2, F12 = PC.triangulation_case(rng, n1, n2, 60)
        a, b = R.search_for_triangulation(k1, k2, F12, False), R.search_for_triangulation(k1, k2, F12, False, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("triangulation", seed)
        s1, s2, s12, R12, t12, m_in = PC.sim3_pair_case(rng, n1, n2)
        a, b = R.search_by_sim3(s1, s2, s12, R12, t12, 7.5, m_in), R.search_by_sim3(s1, s2, s12, R12, t12, 7.5, m_in, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("sim3", seed)
        f1, f2, prev = PC.initialization_case(rng, n1, n2)
        a, b = R.search_for_initialization(f1, f2, prev, 100), R.search_for_initialization(f1, f2, prev, 100, shim="full")
        assert b[2] == a[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), ("initialization", seed)
        (d1, v1, a1, fv1), (d2, v2, a2, fv2) = _bow_case(rng, n1, n2, 30, 0.8, seed % 2)
        a, b = R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, 0.7, True), R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, 0.7, True, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("bow kf/f", seed)
        a = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, 0.75, True)
        b = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, 0.75, True, shim="full")
        assert b[1] == a[1] and np.array_equal(a[0], b[0]), ("bow kf/kf", seed)
