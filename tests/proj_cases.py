"""Seeded cases for the projection-gated searches (ORBmatcher::SearchByProjection(Frame&, const Frame&,
th, bMono) src/ORBmatcher.cc:1578-1724, (Frame&, const vector<MapPoint*>&, th) :63-157) and the host-side replay that turns
the core's per-query matches into what the reference leaves in CurrentFrame.mvpMapPoints.  Test support only."""
import numpy as np

from oracle import oracle_ffi as O

GRID_COLS, GRID_ROWS = 64, 48
SCALE = np.float32(1.2) ** np.arange(8, dtype=np.float32)


def scale_factors(nlevels=8, sf=1.2):
    s = np.ones(nlevels, np.float32)
    for i in range(1, nlevels):
        s[i] = np.float32(s[i - 1] * np.float32(sf))   # mvScaleFactor[i] = mvScaleFactor[i-1] * scaleFactor (float)
    return s


def _rot(rng, deg):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.deg2rad(rng.normal(0, deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def _pose(rng, rot_deg, trans):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = _rot(rng, rot_deg).astype(np.float32)
    T[:3, 3] = np.asarray(trans, np.float32)
    return T


def current_frame(rng, n, w=640, h=480, stereo=True, dense_states=True):
    """a mock current frame: keypoints over the image (some clustered so that search windows hold many candidates), random
    descriptors, octaves, angles, mvuRight, and a MapPoint state per feature (0 none, 1 Observations() == 0, 2 > 0)"""
    sf = scale_factors()
    xy = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1).astype(np.float32)
    if n > 20:   # clusters
        k = n // 4
        c = rng.integers(0, n, 6)
        xy[:k] = (xy[c[rng.integers(0, 6, k)]] + rng.normal(0, 6, (k, 2))).astype(np.float32)
        xy[:, 0] = np.clip(xy[:, 0], 0, w - 1e-3)
        xy[:, 1] = np.clip(xy[:, 1], 0, h - 1e-3)
    octave = rng.integers(0, 8, n).astype(np.int32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    angle = rng.uniform(0, 360, n).astype(np.float32)
    uR = np.full(n, -1, np.float32)
    if stereo:
        m = rng.random(n) < 0.7
        uR[m] = (xy[m, 0] - rng.uniform(2, 60, m.sum())).astype(np.float32)
    state = rng.choice([0, 0, 0, 1, 2], n).astype(np.uint8) if dense_states else np.zeros(n, np.uint8)
    fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6   # TUM3.yaml
    bf = 40.0
    return dict(desc=desc, xy=xy, octave=octave, angle=angle, uRight=uR, state=state, Tcw=np.eye(4, dtype=np.float32),
                K=(fx, fy, cx, cy, bf, bf / fx), bounds=(0.0, float(w), 0.0, float(h)),
                gw_inv=np.float32(GRID_COLS) / np.float32(w), gh_inv=np.float32(GRID_ROWS) / np.float32(h), scale_factors=sf)


def noisy_copy(rng, desc, max_flips):
    d = desc.copy()
    bits = np.unpackbits(d, axis=1)
    for i in range(len(d)):
        k = int(rng.integers(0, max_flips + 1))
        if k:
            bits[i, rng.choice(256, k, replace=False)] ^= 1
    return np.packbits(bits, axis=1)


def last_frame_case(rng, nC, nL, motion="small", stereo=True):
    """(cur, last): last-frame MapPoints are 3-D points that project near features of the current frame under the poses;
    several last features may target the same current feature (the greedy "slot taken" rule is exercised)"""
    cur = current_frame(rng, nC, stereo=stereo)
    fx, fy, cx, cy, mbf, mb = cur["K"]
    tz = {"small": 0.0, "forward": 0.4, "backward": -0.4}[motion]
    TcwL = _pose(rng, 2.0, rng.normal(0, 0.05, 3))
    TcwC = TcwL.copy()
    TcwC[:3, :3] = (_rot(rng, 1.0) @ TcwL[:3, :3].astype(np.float64)).astype(np.float32)
    TcwC[:3, 3] = TcwL[:3, 3] + np.array([rng.normal(0, 0.02), rng.normal(0, 0.02), -tz], np.float32)
    cur["Tcw"] = TcwC
    tgt = rng.integers(0, max(nC, 1), nL) if nC else np.zeros(nL, np.int64)
    if nL > 8 and nC:
        tgt[: nL // 5] = tgt[rng.integers(0, nL, nL // 5)]   # shared targets
    z = rng.uniform(0.4, 9.0, nL)
    if nL > 10:
        z[rng.integers(0, nL, 3)] *= -1   # behind the camera: invzc < 0
    px = (cur["xy"][tgt] if nC else np.zeros((nL, 2))) + rng.normal(0, 2.5, (nL, 2))
    if nL > 10:
        px[rng.integers(0, nL, 3)] += 900   # outside the image bounds
    Xc = np.stack([(px[:, 0] - cx) / fx * z, (px[:, 1] - cy) / fy * z, z], 1)
    Rcw, tcw = TcwC[:3, :3].astype(np.float64), TcwC[:3, 3].astype(np.float64)
    world = ((Xc - tcw) @ Rcw).astype(np.float32)   # Rcw^T (Xc - tcw)
    octL = (np.clip(cur["octave"][tgt] + rng.integers(-1, 2, nL), 0, 7) if nC else np.zeros(nL)).astype(np.int32)
    mpdesc = noisy_copy(rng, cur["desc"][tgt] if nC else np.zeros((nL, 32), np.uint8), 70)
    last = dict(Tcw=TcwL, has_mp=(rng.random(nL) < 0.85).astype(np.uint8), outlier=(rng.random(nL) < 0.08).astype(np.uint8),
                world_pos=world, mpdesc=mpdesc, obs_gt0=(rng.random(nL) < 0.8).astype(np.uint8), octave=octL,
                angle=((cur["angle"][tgt] if nC else np.zeros(nL)) + rng.choice([0, 0, 0, 35, 120], nL) + rng.normal(0, 4, nL)).astype(np.float32) % np.float32(360),
                xy=px.astype(np.float32))
    return cur, last


def local_map_case(rng, nF, nmp):
    cur = current_frame(rng, nF)
    tgt = rng.integers(0, max(nF, 1), nmp) if nF else np.zeros(nmp, np.int64)
    if nmp > 8 and nF:
        tgt[: nmp // 5] = tgt[rng.integers(0, nmp, nmp // 5)]
    pxy = (cur["xy"][tgt] if nF else np.zeros((nmp, 2))) + rng.normal(0, 2.0, (nmp, 2))
    uR = cur["uRight"][tgt] if nF else np.zeros(nmp)
    pxr = np.where(uR > 0, uR + rng.normal(0, 3.0, nmp), pxy[:, 0] - 10)
    mps = dict(in_view=(rng.random(nmp) < 0.9).astype(np.uint8), bad=(rng.random(nmp) < 0.05).astype(np.uint8),
               scale_level=(np.clip(cur["octave"][tgt] + rng.integers(0, 2, nmp), 0, 7) if nF else np.zeros(nmp)).astype(np.int32),
               view_cos=rng.choice([0.9, 0.99, 0.9985, 1.0], nmp).astype(np.float32),
               proj_xyr=np.concatenate([pxy, pxr[:, None]], 1).astype(np.float32),
               mpdesc=noisy_copy(rng, cur["desc"][tgt] if nF else np.zeros((nmp, 32), np.uint8), 80),
               obs_gt0=(rng.random(nmp) < 0.85).astype(np.uint8))
    return cur, mps


def frame_grid(cur):
    """(cell_off, cell_idx) of the frame as Frame::AssignFeaturesToGrid builds it (oracle; pinned to the sliced reference body)"""
    minx, _, miny, _ = cur["bounds"]
    return O.assign_grid(cur["xy"], minx, miny, cur["gw_inv"], cur["gh_inv"])


def core_inputs(cur):
    minx, _, miny, _ = cur["bounds"]
    blocked = (np.asarray(cur["state"]) == 2).astype(np.uint8)   # a MapPoint with Observations() > 0 sits in the slot
    return dict(descF=cur["desc"], xyF=cur["xy"], octF=cur["octave"], grid=frame_grid(cur),
                bounds=(minx, miny, cur["gw_inv"], cur["gh_inv"]), uRight=cur["uRight"], blocked=blocked)


def replay_last_frame(cur, last, valid, match_valid, check_ori=True):
    """what the reference's loop leaves behind, from the core's per-query matches (queries = the valid last features in
    order): assigned[nC] (-1 NULL, -2 pre-existing MapPoint, i >= 0 the MapPoint of last feature i), the return value, and
    the (last, current) feature pairs in match order (the perfect/ overload's point lists)"""
    nC = len(cur["desc"])
    assigned = np.where(np.asarray(cur["state"]) > 0, -2, -1).astype(np.int32)
    hist = [[] for _ in range(30)]
    pairs = []
    nm = 0
    vi = np.nonzero(valid)[0]
    for k, i in enumerate(vi):
        f = int(match_valid[k])
        if f < 0:
            continue
        assigned[f] = i
        nm += 1
        pairs.append((int(i), f))
        if check_ori:
            hist[O.rot_bin(float(last["angle"][i]), float(cur["angle"][f]))].append(f)
    if check_ori:
        keep = O.three_maxima([len(b) for b in hist])
        for b in range(30):
            if b not in keep:
                for f in hist[b]:
                    assigned[f] = -1
                    nm -= 1
    assert nC == len(assigned)
    return assigned, nm, pairs


def replay_local_map(cur, valid, match_valid):
    assigned = np.where(np.asarray(cur["state"]) > 0, -2, -1).astype(np.int32)
    nm = 0
    vi = np.nonzero(valid)[0]
    for k, i in enumerate(vi):
        f = int(match_valid[k])
        if f >= 0:
            assigned[f] = i
            nm += 1
    return assigned, nm


def oracle_last_frame(cur, last, th, mono, check_ori=True, th_high=100):
    """oracle end to end: host gating (orc_proj_queries_last_frame) -> core (orc_search_by_projection) -> replay"""
    q, valid = O.proj_queries_last_frame(cur["Tcw"], last["Tcw"], cur["K"], cur["bounds"], cur["scale_factors"], last["has_mp"],
                                         last["outlier"], last["world_pos"], last["octave"], last["obs_gt0"], th, mono)
    sel = valid.astype(bool)
    m, b, s = O.search_by_projection(queries=q[sel], qdesc=np.asarray(last["mpdesc"]).reshape(-1, 32)[sel], th=th_high, nnratio=0.0,
                                     ratio_rule=0, **core_inputs(cur))
    return replay_last_frame(cur, last, valid, m, check_ori) + (q, valid, m)


def oracle_local_map(cur, mps, th, nnratio=0.8, th_high=100):
    q, valid = O.proj_queries_local_map(cur["scale_factors"], mps["in_view"], mps["bad"], mps["scale_level"], mps["view_cos"],
                                        mps["proj_xyr"], mps["obs_gt0"], th)
    sel = valid.astype(bool)
    m, b, s = O.search_by_projection(queries=q[sel], qdesc=np.asarray(mps["mpdesc"]).reshape(-1, 32)[sel], th=th_high, nnratio=nnratio,
                                     ratio_rule=1, **core_inputs(cur))
    return replay_local_map(cur, valid, m) + (q, valid, m)


def fuse_case(rng, nKF, nmp):
    """(kf, mps) for ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th): map points that project near keypoints of the
    keyframe at a distance consistent with the keypoint's octave, some null / bad / already observed by the keyframe, some
    behind the camera, out of the image, out of their distance range or seen from the wrong side"""
    cur = current_frame(rng, nKF)
    fx, fy, cx, cy, mbf, mb = cur["K"]
    T = _pose(rng, 5.0, rng.normal(0, 0.2, 3))
    Rcw, tcw = T[:3, :3], T[:3, 3]
    Ow = (-(Rcw.astype(np.float64).T @ tcw.astype(np.float64))).astype(np.float32)
    sf = cur["scale_factors"]
    kf = dict(desc=cur["desc"], xy=cur["xy"], octave=cur["octave"], uRight=cur["uRight"], state=rng.choice([0, 0, 1, 1, 2], nKF).astype(np.uint8),
              obs=rng.integers(0, 6, nKF).astype(np.int32), Rcw=Rcw, tcw=tcw, Ow=Ow, K=(fx, fy, cx, cy, mbf), bounds=cur["bounds"],
              gw_inv=cur["gw_inv"], gh_inv=cur["gh_inv"], scale_factors=sf, inv_sigma2=(1.0 / (sf * sf)).astype(np.float32),
              log_scale=np.float32(np.log(np.float32(1.2))))
    tgt = rng.integers(0, max(nKF, 1), nmp) if nKF else np.zeros(nmp, np.int64)
    z = rng.uniform(0.5, 8.0, nmp)
    if nmp > 12:
        z[rng.integers(0, nmp, 3)] *= -1
    px = (cur["xy"][tgt] if nKF else np.zeros((nmp, 2))) + rng.normal(0, 1.5, (nmp, 2))
    if nmp > 12:
        px[rng.integers(0, nmp, 3)] += 900
    Xc = np.stack([(px[:, 0] - cx) / fx * z, (px[:, 1] - cy) / fy * z, z], 1)
    world = ((Xc - tcw.astype(np.float64)) @ Rcw.astype(np.float64)).astype(np.float32)
    PO = world.astype(np.float64) - Ow.astype(np.float64)
    d3 = np.linalg.norm(PO, axis=1)
    lvl = (cur["octave"][tgt] if nKF else np.zeros(nmp)).astype(np.float64) + rng.choice([0, 0, 0, 1], nmp)
    maxd = (d3 * 1.2 ** lvl * rng.uniform(0.93, 0.999, nmp)).astype(np.float32)   # PredictScale -> about lvl
    mind = (maxd / 1.2 ** 7).astype(np.float32)
    if nmp > 12:
        maxd[rng.integers(0, nmp, 3)] *= 0.3   # out of the distance range
    normal = PO / np.maximum(d3[:, None], 1e-9) + rng.normal(0, 0.3, (nmp, 3))
    normal /= np.maximum(np.linalg.norm(normal, axis=1, keepdims=True), 1e-9)
    if nmp > 12:
        normal[rng.integers(0, nmp, 4)] *= -1
    mps = dict(null=(rng.random(nmp) < 0.03).astype(np.uint8), bad=(rng.random(nmp) < 0.05).astype(np.uint8),
               in_kf=(rng.random(nmp) < 0.05).astype(np.uint8), world_pos=world, normal=normal.astype(np.float32), max_dist=maxd,
               min_dist=mind, mpdesc=noisy_copy(rng, cur["desc"][tgt] if nKF else np.zeros((nmp, 32), np.uint8), 60),
               obs=rng.integers(0, 6, nmp).astype(np.int32))
    return kf, mps


def frame_kf_case(rng, nC, nK):
    """(cur, kfp) for SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (Tracking::Relocalization): the
    keyframe's MapPoints project near features of the current frame at a distance consistent with their octave"""
    cur = current_frame(rng, nC, stereo=False)
    fx, fy, cx, cy, mbf, mb = cur["K"]
    T = _pose(rng, 4.0, rng.normal(0, 0.15, 3))
    cur["Tcw"] = T
    Rcw, tcw = T[:3, :3].astype(np.float64), T[:3, 3].astype(np.float64)
    Ow = -(Rcw.T @ tcw)
    tgt = rng.integers(0, max(nC, 1), nK) if nC else np.zeros(nK, np.int64)
    if nK > 8 and nC:
        tgt[: nK // 5] = tgt[rng.integers(0, nK, nK // 5)]
    z = rng.uniform(0.5, 8.0, nK)
    px = (cur["xy"][tgt] if nC else np.zeros((nK, 2))) + rng.normal(0, 2.0, (nK, 2))
    if nK > 12:
        px[rng.integers(0, nK, 3)] += 900
    Xc = np.stack([(px[:, 0] - cx) / fx * z, (px[:, 1] - cy) / fy * z, z], 1)
    world = ((Xc - tcw) @ Rcw).astype(np.float32)
    d3 = np.linalg.norm(world.astype(np.float64) - Ow, axis=1)
    lvl = (cur["octave"][tgt] if nC else np.zeros(nK)).astype(np.float64) + rng.choice([-1, 0, 0, 1], nK)
    maxd = (d3 * 1.2 ** np.clip(lvl, 0, 7) * rng.uniform(0.93, 0.999, nK)).astype(np.float32)
    if nK > 12:
        maxd[rng.integers(0, nK, 3)] *= 0.3
    kfp = dict(angle=((cur["angle"][tgt] if nC else np.zeros(nK)) + rng.choice([0, 0, 0, 35, 120], nK) + rng.normal(0, 4, nK)).astype(np.float32) % np.float32(360),
               has=(rng.random(nK) < 0.85).astype(np.uint8), bad=(rng.random(nK) < 0.05).astype(np.uint8),
               found=(rng.random(nK) < 0.1).astype(np.uint8), world_pos=world, max_dist=maxd, min_dist=(maxd / 1.2 ** 7).astype(np.float32),
               mpdesc=noisy_copy(rng, cur["desc"][tgt] if nC else np.zeros((nK, 32), np.uint8), 70))
    return cur, kfp


def kf_sim3_case(rng, nKF, npts):
    """(kf, Scw, pts, matched_in) for SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (LoopClosing)"""
    kf, mps = fuse_case(rng, nKF, npts)
    s = float(rng.uniform(0.9, 1.1))
    Scw = np.eye(4, dtype=np.float32)
    Scw[:3, :3] = (s * kf["Rcw"].astype(np.float64)).astype(np.float32)
    Scw[:3, 3] = (s * kf["tcw"].astype(np.float64)).astype(np.float32)
    pts = dict(bad=mps["bad"], world_pos=mps["world_pos"], normal=mps["normal"], max_dist=mps["max_dist"], min_dist=mps["min_dist"],
               mpdesc=mps["mpdesc"])
    matched = np.full(nKF, -1, np.int32)
    if nKF > 4:
        k = max(1, nKF // 10)
        matched[rng.choice(nKF, k, replace=False)] = -2
        if npts > 4:
            sl = rng.choice(nKF, max(1, nKF // 20), replace=False)
            matched[sl] = rng.integers(0, npts, len(sl))
    return kf, Scw, pts, matched


def initialization_case(rng, n1, n2):
    """(f1, f2, prev_matched) for ORBmatcher::SearchForInitialization: two monocular frames a small motion apart; most of f2's
    level-0 features are noisy copies of f1 features (several f1 features may aim at one f2 feature, so matches get taken
    over :590-598), angles mostly consistent"""
    f1 = current_frame(rng, n1, stereo=False, dense_states=False)
    f1["octave"] = rng.choice([0, 0, 0, 1, 2, 5], n1).astype(np.int32)
    f2 = current_frame(rng, n2, stereo=False, dense_states=False)
    f2["octave"] = rng.choice([0, 0, 0, 1, 3], n2).astype(np.int32)
    if n1 and n2:
        src = rng.integers(0, n1, n2)
        m = rng.random(n2) < 0.75
        shift = rng.normal(0, 12, 2)
        f2["xy"][m] = np.clip(f1["xy"][src[m]] + shift + rng.normal(0, 3, (int(m.sum()), 2)), [0, 0], [639.9, 479.9]).astype(np.float32)
        f2["desc"][m] = noisy_copy(rng, f1["desc"][src[m]], 45)
        f2["angle"][m] = ((f1["angle"][src[m]] + rng.choice([0, 0, 0, 0, 40, 150], int(m.sum())) + rng.normal(0, 3, int(m.sum()))) % 360).astype(np.float32)
        dup = rng.random(n2) < 0.1   # a second f2 feature with (almost) the same descriptor nearby: the ratio test has work
        f2["desc"][dup] = noisy_copy(rng, f2["desc"][rng.integers(0, n2, int(dup.sum()))], 10)
    prev = f1["xy"].copy()   # Tracking::MonocularInitialization starts from the features' own positions (:622-624)
    return f1, f2, prev


def sim3_pair_case(rng, n1, n2):
    """(k1, k2, s12, R12, t12, matches_in) for ORBmatcher::SearchBySim3 (LoopClosing::ComputeSim3): two keyframes with poses of
    their own and a candidate similarity between their camera frames; a share of the features are mutual pairs (the point of
    k1's feature lands on k2's feature and the other way round), the rest aim at random features or nowhere"""
    c1, c2 = current_frame(rng, n1), current_frame(rng, n2)
    fx, fy, cx, cy, mbf, mb = c1["K"]
    sf = c1["scale_factors"]
    T1, T2 = _pose(rng, 5.0, rng.normal(0, 0.2, 3)).astype(np.float64), _pose(rng, 5.0, rng.normal(0, 0.2, 3)).astype(np.float64)
    s12 = float(rng.uniform(0.85, 1.15))
    R12 = _rot(rng, 3.0).astype(np.float64)
    t12 = rng.normal(0, 0.05, 3)
    sR21 = (1.0 / s12) * R12.T
    t21 = -sR21 @ t12

    def side(cs, n_s, T_s, co, n_o, to_other_R, to_other_t, back_R, back_t, tgt):
        """points of the features of `cs` that land on features tgt of `co` after (to_other_R, to_other_t)"""
        z = rng.uniform(0.5, 8.0, n_s)
        px = (co["xy"][tgt] if n_o else np.zeros((n_s, 2))) + rng.normal(0, 1.5, (n_s, 2))
        if n_s > 12:
            z[rng.integers(0, n_s, 2)] *= -1
            px[rng.integers(0, n_s, 2)] += 900
        Xo = np.stack([(px[:, 0] - cx) / fx * z, (px[:, 1] - cy) / fy * z, z], 1)       # in the other camera
        Xs = (Xo - to_other_t) @ np.linalg.inv(to_other_R).T                             # in the own camera
        world = ((Xs - T_s[:3, 3]) @ T_s[:3, :3]).astype(np.float32)
        d3 = np.linalg.norm(Xo, axis=1)
        lvl = (co["octave"][tgt] if n_o else np.zeros(n_s)).astype(np.float64) + rng.choice([0, 0, 0, 1], n_s)
        maxd = (d3 * 1.2 ** lvl * rng.uniform(0.93, 0.999, n_s)).astype(np.float32)
        if n_s > 12:
            maxd[rng.integers(0, n_s, 2)] *= 0.3
        k = dict(desc=cs["desc"], xy=cs["xy"], octave=cs["octave"], uRight=cs["uRight"], K=(fx, fy, cx, cy, mbf), bounds=cs["bounds"],
                 gw_inv=cs["gw_inv"], gh_inv=cs["gh_inv"], scale_factors=sf, inv_sigma2=(1.0 / (sf * sf)).astype(np.float32),
                 log_scale=np.float32(np.log(np.float32(1.2))), state=rng.choice([0, 1, 1, 1, 1, 2], n_s).astype(np.uint8),
                 world_pos=world, max_dist=maxd, min_dist=(maxd / 1.2 ** 7).astype(np.float32),
                 mpdesc=noisy_copy(rng, co["desc"][tgt] if n_o else np.zeros((n_s, 32), np.uint8), 70),
                 Rcw=T_s[:3, :3].astype(np.float32), tcw=T_s[:3, 3].astype(np.float32))
        return k

    tgt12 = rng.integers(0, max(n2, 1), n1)
    tgt21 = rng.integers(0, max(n1, 1), n2)
    if n1 and n2:   # mutual pairs: a random partial matching
        m = min(n1, n2) * 3 // 5
        a, b = rng.permutation(n1)[:m], rng.permutation(n2)[:m]
        tgt12[a] = b
        tgt21[b] = a
    k1 = side(c1, n1, T1, c2, n2, sR21, t21, None, None, tgt12)
    k2 = side(c2, n2, T2, c1, n1, s12 * R12, t12, None, None, tgt21)
    matches = np.full(n1, -1, np.int32)
    if n1 > 6 and n2 > 2:
        sl = rng.choice(n1, n1 // 8, replace=False)
        matches[sl] = np.where(rng.random(len(sl)) < 0.5, -2, rng.integers(0, n2, len(sl)))
        good2 = k2["state"] > 0
        matches[(matches >= 0) & ~good2[np.clip(matches, 0, n2 - 1)]] = -1   # a preset match is a point k2 really has
    return k1, k2, s12, R12.astype(np.float32), t12.astype(np.float32), matches


def _fv_csr(nodes_of_feature, present):
    """FeatureVector of a keyframe as CSR: ascending node ids, features of a node in ascending index (DBoW2's insertion order)"""
    ids = sorted(set(int(v) for v, p in zip(nodes_of_feature, present) if p))
    node, off, idx = [], [0], []
    for nid in ids:
        node.append(nid)
        idx += [i for i in range(len(nodes_of_feature)) if present[i] and nodes_of_feature[i] == nid]
        off.append(len(idx))
    return np.array(node, np.uint32), np.array(off, np.uint32), np.array(idx, np.uint32)


def triangulation_case(rng, n1, n2, nnodes):
    """(k1, k2, F12) for ORBmatcher::SearchForTriangulation: two keyframes of one scene with a known relative pose (F12 as
    LocalMapping::ComputeF12 builds it), shared points with noisy descriptors in a common vocabulary node, keypoint noise
    small or large against the epipolar gate, distractors, stereo / monocular keypoints, features that already have MapPoints"""
    w, h = 640, 480
    fx, fy, cx, cy, mbf = 535.4, 539.2, 320.1, 247.6, 40.0
    sf = scale_factors()
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    T1 = _pose(rng, 3.0, rng.normal(0, 0.1, 3)).astype(np.float64)
    T2 = T1.copy()
    T2[:3, :3] = _rot(rng, 4.0) @ T1[:3, :3]
    T2[:3, 3] = T1[:3, 3] + np.array([rng.choice([-1, 1]) * rng.uniform(0.1, 0.4), rng.normal(0, 0.05), rng.normal(0, 0.05)])
    R1, t1, R2, t2 = T1[:3, :3], T1[:3, 3], T2[:3, :3], T2[:3, 3]
    R12 = R1 @ R2.T
    t12 = -R12 @ t2 + t1
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(np.float32)
    npts = max(n1, n2)
    # world points in front of camera 1
    z = rng.uniform(1.0, 8.0, npts)
    p1 = np.stack([rng.uniform(20, w - 20, npts), rng.uniform(20, h - 20, npts)], 1)
    Xc1 = np.stack([(p1[:, 0] - cx) / fx * z, (p1[:, 1] - cy) / fy * z, z], 1)
    Xw = (Xc1 - t1) @ R1
    Xc2 = Xw @ R2.T + t2
    p2 = np.stack([fx * Xc2[:, 0] / Xc2[:, 2] + cx, fy * Xc2[:, 1] / Xc2[:, 2] + cy], 1)
    base = rng.integers(0, 256, (npts, 32), dtype=np.uint8)
    pnode = rng.integers(0, max(nnodes, 1), npts)

    def side(n, pts, other):
        sel = rng.permutation(npts)[:n]   # which scene points this keyframe sees
        noise = np.where(rng.random(n)[:, None] < 0.8, rng.normal(0, 0.6, (n, 2)), rng.normal(0, 6.0, (n, 2)))
        xy = (pts[sel] + noise).astype(np.float32)
        octave = rng.integers(0, 8, n).astype(np.int32)
        nodes = np.where(rng.random(n) < 0.9, pnode[sel], rng.integers(0, max(nnodes, 1), n))
        present = rng.random(n) < 0.95
        uR = np.where(rng.random(n) < 0.5, xy[:, 0] - rng.uniform(2, 40, n), -1).astype(np.float32)
        return dict(desc=noisy_copy(rng, base[sel], 45), xy=xy, octave=octave, angle=(rng.choice([10.0, 10.0, 10.0, 200.0], n) + rng.normal(0, 5, n)).astype(np.float32) % np.float32(360),
                    uRight=uR, has_mp=(rng.random(n) < 0.3).astype(np.uint8), fv=_fv_csr(nodes, present), K=(fx, fy, cx, cy, mbf),
                    bounds=(0.0, float(w), 0.0, float(h)), gw_inv=np.float32(GRID_COLS) / np.float32(w), gh_inv=np.float32(GRID_ROWS) / np.float32(h),
                    scale_factors=sf, inv_sigma2=(1.0 / (sf * sf)).astype(np.float32), level_sigma2=(sf * sf).astype(np.float32),
                    log_scale=np.float32(np.log(np.float32(1.2))))
    k1, k2 = side(n1, p1, None), side(n2, p2, None)
    k1["Ow"] = (-(R1.T @ t1)).astype(np.float32)
    k2["Rcw"], k2["tcw"] = R2.astype(np.float32), t2.astype(np.float32)
    return k1, k2, F12


def tri_core_inputs(k1, k2, only_stereo):
    """the arrays of orbfe_search_for_triangulation / orc_search_for_triangulation from two keyframe dicts, as the shim builds them"""
    def prep(k):
        st = (np.asarray(k["uRight"]) >= 0).astype(np.uint8)
        el = ((np.asarray(k["has_mp"]) == 0) & ((st == 1) | (not only_stereo))).astype(np.uint8)
        return dict(desc=k["desc"], xy=k["xy"], elig=el, stereo=st, fv=k["fv"], octave=k["octave"], scale_factors=k["scale_factors"],
                    level_sigma2=k["level_sigma2"])
    # the epipole of keyframe 1 in keyframe 2 (:833-843), float operation order of the reference on the stub's cv::Mat
    f32 = np.float32
    R, t, C = np.asarray(k2["Rcw"], f32).reshape(3, 3), np.asarray(k2["tcw"], f32), np.asarray(k1["Ow"], f32)
    C2 = np.zeros(3, f32)
    for y in range(3):
        s = f32(0)
        for k in range(3):
            s = f32(s + f32(R[y, k] * C[k]))
        C2[y] = f32(s + t[y])
    fx, fy, cx, cy = [f32(v) for v in k2["K"][:4]]
    invz = f32(f32(1.0) / C2[2])
    ex = f32(f32(f32(fx * C2[0]) * invz) + cx)
    ey = f32(f32(f32(fy * C2[1]) * invz) + cy)
    return prep(k1), prep(k2), ex, ey


def replay_triangulation(k1, k2, m12, check_ori=True):
    """rotation histogram (:919-985) and vMatchedPairs (:987-997) from the core's match12"""
    m = np.array(m12, np.int32).copy()
    nm = int((m >= 0).sum())
    if check_ori:
        hist = [[] for _ in range(30)]
        for i in np.nonzero(m >= 0)[0]:
            hist[O.rot_bin(float(k1["angle"][i]), float(k2["angle"][m[i]]))].append(i)
        keep = O.three_maxima([len(b) for b in hist])
        for b in range(30):
            if b not in keep:
                for i in hist[b]:
                    m[i] = -1
                    nm -= 1
    pairs = np.array([(i, m[i]) for i in range(len(m)) if m[i] >= 0], np.int32).reshape(-1, 2)
    return pairs, nm
