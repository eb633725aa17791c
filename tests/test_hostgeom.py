"""The host-side geometry of the projection-gated matchers behind the C-ABI (csrc/orbfe_hostgeom.hip): orbfe_project_points,
orbfe_proj_queries_local_map, orbfe_rotation_consistency, orbfe_initialization_resolve.  CPU only (no device is touched):
against the oracle's restatements -- which tests/test_ref_pin.py pins to the reference's compiled bodies -- and against
independent numpy / pure-Python twins.  The members of shim/ORBmatcher_orbfe.cc built on these are compared with the reference's
compiled bodies end to end in the gpu-marked tests (tests/test_projection.py, tests/test_shim_ref.py)."""
import ctypes as C

import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd import _ffi
from orb_slam2_ssd_semantic_amd._ffi import ptr

F32 = np.float32
PJ_NEG_DEPTH, PJ_NEG_INVZ, PJ_CHAINED, PJ_CLOSED = 1, 2, 4, 8
Q_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"), ("ur", "<f4"),
                    ("flags", "<i4"), ("pad", "<i4")])


def project(R, t, pos, K, bounds, flags, R2=None, t2=None, Ow=None, normal=None, dmin=None, dmax=None):
    L = _ffi.lib()
    n = len(pos)
    a = lambda x: None if x is None else np.ascontiguousarray(x, F32)
    R, t, R2, t2, Ow, pos, normal, dmin, dmax = (a(x) for x in (R, t, R2, t2, Ow, pos, normal, dmin, dmax))
    u, v, iz, d, ur = (np.zeros(max(n, 1), F32) for _ in range(5))
    ok = np.zeros(max(n, 1), np.uint8)
    p = lambda x: None if x is None else ptr(x)
    rc = L.orbfe_project_points(p(R), p(t), p(R2), p(t2), p(Ow), *[float(k) for k in K], *[float(b) for b in bounds], flags, n, p(pos),
                                p(normal), p(dmin), p(dmax), ptr(u), ptr(v), ptr(iz), ptr(d), ptr(ur), ptr(ok))
    assert rc == 0
    return u[:n], v[:n], iz[:n], d[:n], ur[:n], ok[:n].astype(bool)


def rand_pose(rng, scale=1.0):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return (q * scale).astype(F32), rng.normal(0, 0.3, 3).astype(F32)


def twin_project(R, t, pos, K, bounds, flags, R2=None, t2=None, Ow=None, normal=None, dmin=None, dmax=None):
    """float32 arithmetic in the documented order, written independently (numpy scalars, no vectorisation tricks)"""
    fx, fy, cx, cy, bf = (F32(k) for k in K)
    minx, maxx, miny, maxy = (F32(b) for b in bounds)
    out = []
    for i, xw in enumerate(np.asarray(pos, F32)):
        def aff(M, tt, x):
            return np.array([F32(F32(F32(M[k, 0] * x[0]) + F32(M[k, 1] * x[1])) + F32(M[k, 2] * x[2])) + tt[k] for k in range(3)], F32)
        pc = aff(R, t, xw)
        if R2 is not None:
            pc = aff(R2, t2, pc)
        if flags & PJ_NEG_DEPTH and pc[2] < 0:
            out.append(None)
            continue
        with np.errstate(all="ignore"):
            iz = F32(1.0) / pc[2]
        if flags & PJ_NEG_INVZ and iz < 0:
            out.append(None)
            continue
        with np.errstate(all="ignore"):   # a point at the camera centre: inf * 0
            if flags & PJ_CHAINED:
                u, v = F32(F32(fx * pc[0]) * iz) + cx, F32(F32(fy * pc[1]) * iz) + cy
            else:
                u, v = F32(fx * F32(pc[0] * iz)) + cx, F32(fy * F32(pc[1] * iz)) + cy
        inside = (not (u < minx or u > maxx or v < miny or v > maxy)) if flags & PJ_CLOSED else (minx <= u < maxx and miny <= v < maxy)
        if not inside:
            out.append(None)
            continue
        d3 = F32(0)
        if dmin is not None:
            po = (xw - Ow).astype(F32) if Ow is not None else pc
            s = float(po[0]) * float(po[0]) + float(po[1]) * float(po[1]) + float(po[2]) * float(po[2])
            d3 = F32(np.sqrt(s))
            if d3 < dmin[i] or d3 > dmax[i]:
                out.append(None)
                continue
            if normal is not None and Ow is not None:
                nn = normal[i]
                if float(po[0]) * float(nn[0]) + float(po[1]) * float(nn[1]) + float(po[2]) * float(nn[2]) < 0.5 * float(d3):
                    out.append(None)
                    continue
        out.append((u, v, iz, d3, F32(u - F32(bf * iz))))
    return out


@pytest.mark.parametrize("seed,flags,second,centre,with_normal,with_range", [
    (0, PJ_NEG_INVZ | PJ_CHAINED | PJ_CLOSED, False, False, False, False),   # last frame (:1620-1642)
    (1, PJ_CHAINED | PJ_CLOSED, False, True, False, True),                  # relocalisation (:1778-1803)
    (2, PJ_NEG_DEPTH, False, True, True, True),                             # Sim3 projection / Fuse (:401-433, :1060-1101)
    (3, PJ_NEG_DEPTH, True, False, False, True),                            # SearchBySim3 (:1389-1424)
    (4, 0, False, False, False, False)])
def test_project_points_equals_the_twin(seed, flags, second, centre, with_normal, with_range):
    rng = np.random.default_rng(seed)
    n = 4000
    R, t = rand_pose(rng)
    R2, t2 = rand_pose(rng, 1.13) if second else (None, None)
    pos = rng.normal(0, 3, (n, 3)).astype(F32)
    pos[:, 2] += 2
    pos[7] = (-R.T @ t)                     # a point at the camera centre: z_c ~ 0
    Ow = rng.normal(0, 0.5, 3).astype(F32) if centre else None
    normal = rng.normal(0, 1, (n, 3)).astype(F32) if with_normal else None
    dmin = rng.uniform(0.2, 2, n).astype(F32) if with_range else None
    dmax = (dmin + rng.uniform(1.0, 9, n).astype(F32)) if with_range else None
    K, bounds = (517.3, 516.5, 318.6, 255.3, 40.0), (0.0, 640.0, 0.0, 480.0)
    u, v, iz, d, ur, ok = project(R, t, pos, K, bounds, flags, R2, t2, Ow, normal, dmin, dmax)
    ref = twin_project(R, t, pos, K, bounds, flags, R2, t2, Ow, normal, dmin, dmax)
    assert 60 < ok.sum() < n
    for i in range(n):
        assert ok[i] == (ref[i] is not None), i
        if ok[i]:
            got = np.array([u[i], v[i], iz[i], d[i], ur[i]], F32)
            assert np.array_equal(got.view(np.uint32), np.array(ref[i], F32).view(np.uint32)), (i, got, ref[i])


def test_last_frame_projection_equals_the_oracle_queries(oracle):
    """orbfe_project_points with the last-frame flags + the shim's level-window rule == orc_proj_queries_last_frame (pinned to the
    compiled body of SearchByProjection(Frame&, const Frame&, ...) by tests/test_ref_pin.py)"""
    rng = np.random.default_rng(5)
    n = 3000
    Tc, Tl = np.eye(4, dtype=F32), np.eye(4, dtype=F32)
    Tc[:3, :3], Tc[:3, 3] = rand_pose(rng)
    Tl[:3, :3], Tl[:3, 3] = rand_pose(rng)
    pos = (rng.normal(0, 2, (n, 3)) + (-Tc[:3, :3].T @ Tc[:3, 3]) + Tc[2, :3] * 3).astype(F32)
    K, bounds = (517.3, 516.5, 318.6, 255.3, 40.0, 0.08), (0.0, 640.0, 0.0, 480.0)
    sf = (F32(1.2) ** np.arange(8)).astype(F32)
    has, outl, obs = rng.random(n) < 0.8, rng.random(n) < 0.1, rng.random(n) < 0.6
    octv = rng.integers(0, 8, n)
    q, valid = oracle.proj_queries_last_frame(Tc, Tl, K, bounds, sf, has, outl, pos, octv, obs, 7.0, False)
    u, v, iz, d, ur, ok = project(Tc[:3, :3], Tc[:3, 3], pos, K[:5], bounds, PJ_NEG_INVZ | PJ_CHAINED | PJ_CLOSED)
    ok &= has & ~outl
    assert np.array_equal(ok, valid.astype(bool)) and 100 < ok.sum() < n
    for f, a in (("u", u), ("v", v), ("ur", ur)):
        assert np.array_equal(q[f][ok].view(np.uint32), a[ok].view(np.uint32)), f
    assert np.array_equal(q["r"][ok].view(np.uint32), (F32(7.0) * sf[octv[ok]]).view(np.uint32))


def test_local_map_queries_equal_the_oracle(oracle):
    L = _ffi.lib()
    rng = np.random.default_rng(6)
    for th in (1.0, 3.0, 5.0):
        n = 2500
        sf = (F32(1.2) ** np.arange(8)).astype(F32)
        in_view, bad, obs = (rng.random(n) < p for p in (0.7, 0.1, 0.5))
        level = rng.integers(0, 8, n).astype(np.int32)
        vc = rng.uniform(0.99, 1.0, n).astype(F32)
        vc[:5] = F32(0.998)
        uvr = rng.uniform(0, 640, (n, 3)).astype(F32)
        q, valid = oracle.proj_queries_local_map(sf, in_view, bad, level, vc, uvr, obs, th)
        out, src, nq = np.zeros(n, Q_DTYPE), np.zeros(n, np.int32), C.c_int32()
        iv, bd, ob = (np.ascontiguousarray(x, np.uint8) for x in (in_view, bad, obs))
        assert L.orbfe_proj_queries_local_map(ptr(sf), n, ptr(iv), ptr(bd), ptr(level), ptr(vc), ptr(uvr), ptr(ob), th, ptr(out), ptr(src),
                                              C.byref(nq)) == 0
        want = np.nonzero(valid)[0]
        assert nq.value == len(want) and np.array_equal(src[:nq.value], want)
        for f in ("u", "v", "r", "min_level", "max_level", "ur"):
            assert np.array_equal(out[f][:nq.value].view(np.uint32), q[f][want].view(np.uint32)), f
        assert np.array_equal(out["flags"][:nq.value], np.where(obs[want], 3, 2))     # RIGHT_GATE | CLAIMS iff the point has observations


def test_rotation_consistency_equals_bins_plus_three_maxima(oracle):
    L = _ffi.lib()
    rng = np.random.default_rng(7)
    for case in range(300):
        n = int(rng.integers(0, 400))
        a = rng.uniform(0, 360, n).astype(F32)
        b = rng.uniform(0, 360, n).astype(F32) if case % 3 else (a + rng.normal(case % 7 * 20, 8, n)).astype(F32) % F32(360)
        if case % 5 == 0 and n:
            b[: n // 2] = a[: n // 2]                                  # a dominant bin 0: the 0.1x rules fire
        drop = np.ones(max(n, 1), np.uint8)
        assert L.orbfe_rotation_consistency(ptr(a), ptr(b), n, 30, ptr(drop)) == 0
        bins = np.array([oracle.rot_bin(float(x), float(y)) for x, y in zip(a, b)], np.int64)
        counts = np.bincount(bins, minlength=30)[:30]
        keep = set(int(i) for i in oracle.three_maxima(counts) if i >= 0)
        assert np.array_equal(drop[:n].astype(bool), np.array([bb not in keep for bb in bins], bool)), case


def test_rotation_consistency_rejects_bins_outside_the_histogram():
    """histo_len < 19 with ordinary angles, NaN and angles outside [0, 360) index past the histogram in the
    reference (it asserts, src/ORBmatcher.cc:314); the C-ABI returns ORBFE_ERR_ARG and writes nothing."""
    L = _ffi.lib()
    a = np.array([350.0, 10.0, 20.0], F32)
    b = np.array([0.0, 10.0, 20.0], F32)
    drop = np.full(3, 7, np.uint8)
    assert L.orbfe_rotation_consistency(ptr(a), ptr(b), 3, 5, ptr(drop)) == _ffi.ORBFE_ERR_ARG   # round(350 / 5) = 70 >= 5
    assert np.all(drop == 7)
    assert L.orbfe_rotation_consistency(ptr(a), ptr(b), 3, 19, ptr(drop)) == 0                   # round(350 / 19) = 18 < 19
    for bad in (np.nan, np.inf, -np.inf, 1.0e6, -1.0e6):
        a2 = a.copy()
        a2[1] = bad
        drop[:] = 7
        assert L.orbfe_rotation_consistency(ptr(a2), ptr(b), 3, 30, ptr(drop)) == _ffi.ORBFE_ERR_ARG, bad
        assert np.all(drop == 7)
    # histo_len == bin (360 / 1 at histo_len 360 -> bin 1; d = 360 - eps at histo_len 19 -> bin 19 -> wraps to 0)
    a3 = np.array([359.99], F32)
    b3 = np.array([0.0], F32)
    assert L.orbfe_rotation_consistency(ptr(a3), ptr(b3), 1, 19, ptr(drop)) == 0 and drop[0] == 0


def test_initialization_resolve_equals_the_sequential_rule():
    """pure-Python transcription of the rule's MEANING (not of the reference's statements): queries in order; a candidate held
    by an earlier query at a distance <= the own one is invisible; accept best <= th and best < second * nnratio; take over."""
    L = _ffi.lib()
    rng = np.random.default_rng(8)
    for case in range(200):
        nq, n2 = int(rng.integers(0, 120)), int(rng.integers(1, 60))
        lists = [[(int(j), int(rng.integers(0, 120))) for j in rng.choice(n2, int(rng.integers(0, min(n2, 12))), replace=False)] for _ in range(nq)]
        off = np.zeros(nq + 1, np.uint32)
        ent = []
        for k, lst in enumerate(lists):
            ent += [j | d << 16 for j, d in lst]
            off[k + 1] = len(ent)
        ent = np.array(ent or [0], np.uint32)
        acc, hold = np.full(max(nq, 1), -9, np.int32), np.full(n2, -9, np.int32)
        ratio = float(rng.choice([0.6, 0.9]))
        assert L.orbfe_initialization_resolve(ptr(off), ptr(ent), nq, n2, 50, ratio, ptr(acc), ptr(hold)) == 0
        held, holder, want = {}, {}, []
        for k, lst in enumerate(lists):
            vis = [(d, pos, j) for pos, (j, d) in enumerate(lst) if not (j in held and held[j] <= d)]
            vis.sort(key=lambda e: (e[0], e[1]))
            if vis and vis[0][0] <= 50 and F32(vis[0][0]) < F32(F32(vis[1][0] if len(vis) > 1 else 2147483647) * F32(ratio)):
                j = vis[0][2]
                held[j], holder[j] = vis[0][0], k
                want.append(j)
            else:
                want.append(-1)
        assert acc[:nq].tolist() == want, case
        assert hold.tolist() == [holder.get(j, -1) for j in range(n2)], case
