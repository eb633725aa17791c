"""Frame grid index (AssignFeaturesToGrid / GetFeaturesInArea, reference src/Frame.cc:319-334, 465-531).

CPU part: the C oracle against a definition-level Python twin.  GPU part: the HIP path (through the C-ABI) against the
oracle, index-exact including candidate order."""
import math

import numpy as np
import pytest

F = np.float32


def grid_case(seed, n, w=640.0, h=480.0, spill=20.0):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(-spill, w + spill, n), rng.uniform(-spill, h + spill, n)], 1).astype(F)
    if n > 8:  # exact half-cell positions: round() ties away from zero
        xy[:4, 0] = F(5.0) * np.arange(4, dtype=F)
        xy[4:8, 1] = F(15.0) + F(10.0) * np.arange(4, dtype=F)
    octave = rng.integers(0, 8, n).astype(np.int32)
    minx, miny = F(0.0), F(0.0)
    gwi = F(64) / F(w - 0.0)
    ghi = F(48) / F(h - 0.0)
    return xy, octave, float(minx), float(miny), float(gwi), float(ghi)


def twin_assign(xy, minx, miny, gwi, ghi):
    cells = [[] for _ in range(64 * 48)]
    for i, (x, y) in enumerate(xy):
        vx = F(F(x) - F(minx)) * F(gwi)
        vy = F(F(y) - F(miny)) * F(ghi)
        px = int(math.copysign(math.floor(abs(float(vx)) + 0.5), float(vx)))  # C round(): half away from zero
        py = int(math.copysign(math.floor(abs(float(vy)) + 0.5), float(vy)))
        if 0 <= px < 64 and 0 <= py < 48:
            cells[px * 48 + py].append(i)
    return cells


def twin_area(xy, octave, cells, minx, miny, gwi, ghi, x, y, r, minL, maxL):
    x, y, r = F(x), F(y), F(r)
    out = []
    a = max(0, int(math.floor(float(F(F(F(x - F(minx)) - r) * F(gwi))))))
    if a >= 64:
        return out
    b = min(63, int(math.ceil(float(F(F(F(x - F(minx)) + r) * F(gwi))))))
    if b < 0:
        return out
    c = max(0, int(math.floor(float(F(F(F(y - F(miny)) - r) * F(ghi))))))
    if c >= 48:
        return out
    d = min(47, int(math.ceil(float(F(F(F(y - F(miny)) + r) * F(ghi))))))
    if d < 0:
        return out
    chk = minL > 0 or maxL >= 0
    for ix in range(a, b + 1):
        for iy in range(c, d + 1):
            for k in cells[ix * 48 + iy]:
                if chk:
                    if octave[k] < minL:
                        continue
                    if maxL >= 0 and octave[k] > maxL:
                        continue
                if abs(F(xy[k, 0] - x)) < r and abs(F(xy[k, 1] - y)) < r:
                    out.append(k)
    return out


def queries(seed, nq):
    rng = np.random.default_rng(1000 + seed)
    q = np.stack([rng.uniform(-60, 700, nq), rng.uniform(-60, 540, nq), rng.uniform(1, 60, nq)], 1).astype(F)
    lv = np.stack([rng.integers(-1, 4, nq), rng.integers(-1, 8, nq)], 1).astype(np.int32)
    if nq > 4:
        q[0] = (-500, 100, 10)   # entirely left of the grid
        q[1] = (5000, 100, 10)   # entirely right
        q[2] = (100, -900, 10)
        q[3] = (100, 9000, 10)
    return q, lv


@pytest.mark.parametrize("seed,n", [(0, 1000), (1, 2000), (2, 9), (3, 0), (4, 1)])
def test_oracle_grid_vs_twin(oracle, seed, n):
    xy, octave, minx, miny, gwi, ghi = grid_case(seed, n)
    off, idx = oracle.assign_grid(xy, minx, miny, gwi, ghi)
    cells = twin_assign(xy, minx, miny, gwi, ghi)
    assert off[0] == 0 and off[-1] == sum(len(c) for c in cells)
    for c in range(64 * 48):
        assert list(idx[off[c]:off[c + 1]]) == cells[c]
    q, lv = queries(seed, 40)
    for (x, y, r), (a, b) in zip(q, lv):
        got = oracle.features_in_area(xy, octave, off, idx, minx, miny, gwi, ghi, float(x), float(y), float(r), int(a),
                                      int(b))
        assert list(got) == twin_area(xy, octave, cells, minx, miny, gwi, ghi, x, y, r, int(a), int(b))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,nq", [(0, 1000, 1000), (1, 2000, 3000), (2, 9, 5), (3, 0, 7), (4, 1, 1), (5, 8000, 100),
                                       (6, 500, 0)])
def test_gpu_grid_parity(oracle, seed, n, nq):
    from orb_slam2_ssd_semantic_amd import FrameGrid, ORBmatcher
    mt = ORBmatcher(0.9, True)
    xy, octave, minx, miny, gwi, ghi = grid_case(seed, n)
    g = FrameGrid(mt, xy, octave, minx, miny, gwi, ghi)
    off, idx = oracle.assign_grid(xy, minx, miny, gwi, ghi)
    assert np.array_equal(g.cell_off, off)
    assert np.array_equal(g.cell_idx, idx)
    q, lv = queries(seed, nq)
    for levels in (lv, None):
        qoff, cand = g.query(q, levels, cap=16)  # small cap: exercises the ORBFE_ERR_CAP retry
        assert qoff[0] == 0 and len(cand) == qoff[-1]
        for i in range(nq):
            a, b = (int(lv[i, 0]), int(lv[i, 1])) if levels is not None else (-1, -1)
            ref = oracle.features_in_area(xy, octave, off, idx, minx, miny, gwi, ghi, float(q[i, 0]), float(q[i, 1]),
                                          float(q[i, 2]), a, b)
            assert np.array_equal(cand[qoff[i]:qoff[i + 1]], ref)


@pytest.mark.gpu
def test_gpu_projection_search_pipeline(oracle):
    """grid query -> hamming_csr == the oracle's two stages chained (SearchByProjection inner loop)."""
    from orb_slam2_ssd_semantic_amd import FrameGrid, ORBmatcher
    rng = np.random.default_rng(7)
    mt = ORBmatcher(0.9, True)
    xy, octave, minx, miny, gwi, ghi = grid_case(11, 2000)
    desc = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    qd = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    g = FrameGrid(mt, xy, octave, minx, miny, gwi, ghi)
    q, lv = queries(11, 500)
    off, cand = g.query(q, lv)
    bi, b, s = mt.HammingCSR(qd, desc, off, cand)
    rbi, rb, rs = oracle.hamming_csr(qd, desc, off, cand)
    assert np.array_equal(bi, rbi) and np.array_equal(b, rb) and np.array_equal(s, rs)


def test_grid_bad_args_cpu():
    """argument validation happens before any device work"""
    from orb_slam2_ssd_semantic_amd import _ffi
    L = _ffi.lib()
    assert L.orbfe_assign_grid(None, None, 0, 0.0, 0.0, 1.0, 1.0, None, None, None) == _ffi.ORBFE_ERR_ARG
    assert L.orbfe_features_in_area(None, None, None, 0, None, None, 0.0, 0.0, 1.0, 1.0, None, None, 0, None, None,
                                    0) == _ffi.ORBFE_ERR_ARG


@pytest.mark.gpu
def test_gpu_grid_batch_device_on_extractor_output(oracle):
    """Device-resident: a batched extractor call, then AssignFeaturesToGrid for every frame on its output block
    (orbfe_assign_grid_batch_device), nothing through the host; per frame against the oracle on the same keypoints
    (cell offsets, cell lists in keypoint order, number of keypoints inside the grid)."""
    import torch
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor, ORBmatcher
    from orb_slam2_ssd_semantic_amd.synth import synth_frame
    w, h, B = 640, 480, 9
    frames = np.stack([synth_frame(300 + i, h, w, sparse=(i % 2 == 1)) for i in range(B)])
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = e.capacity()
    dg = torch.from_numpy(frames).cuda()
    dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
    dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    dn = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), st)
    d_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    d_idx = torch.full((B, cap), -1, dtype=torch.int32, device="cuda")
    d_nin = torch.zeros(B, dtype=torch.int32, device="cuda")
    # mnMinX = 0 ... as Frame::ComputeImageBounds gives without distortion (src/Frame.cc:533-543); a grid that does not
    # cover the whole image for the second half of the checks (keypoints outside the grid are skipped, :327)
    for (minx, miny, maxx, maxy) in ((0.0, 0.0, float(w), float(h)), (100.0, 50.0, 500.0, 400.0)):
        gwi, ghi = float(F(64) / F(maxx - minx)), float(F(48) / F(maxy - miny))
        ORBmatcher(0.9, True).AssignFeaturesToGrid_batch_device(dk.data_ptr(), dn.data_ptr(), cap, B, minx, miny, gwi, ghi,
                                                                d_off.data_ptr(), d_idx.data_ptr(), d_nin.data_ptr(), st)
        torch.cuda.synchronize()
        n, kp = dn.cpu().numpy(), dk.cpu().numpy()
        off, idx, nin = d_off.cpu().numpy().view(np.uint32), d_idx.cpu().numpy().view(np.uint32), d_nin.cpu().numpy()
        for i in range(B):
            k = kp[i, :n[i]].copy().view(KP_DTYPE).reshape(-1)
            xy = np.stack([k["x"], k["y"]], 1)
            roff, ridx = oracle.assign_grid(xy, minx, miny, gwi, ghi)
            assert nin[i] == len(ridx) and (minx > 0 or nin[i] == n[i])
            assert np.array_equal(off[i], roff) and np.array_equal(idx[i, :nin[i]], ridx), i


@pytest.mark.gpu
def test_gpu_projection_search_chain_device_resident(oracle):
    """The SearchByProjection inner loop without leaving the device: extractor block -> grid of every frame -> for frame f,
    GetFeaturesInArea of 400 queries (orbfe_features_in_area_device) -> best / second-best Hamming over the candidate lists
    (orbfe_hamming_csr_device, query descriptors = the previous frame's).  Candidate lists and matches against the oracle's
    two stages on the same keypoints; a too small `cap` reports the required size and writes nothing."""
    import torch
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor, ORBmatcher, _ffi
    from orb_slam2_ssd_semantic_amd.synth import synth_frame
    w, h, B, nq = 640, 480, 4, 400
    frames = np.stack([synth_frame(700 + i, h, w) for i in range(B)])
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    mt = ORBmatcher(0.9, True)
    cap = e.capacity()
    st = torch.cuda.current_stream().cuda_stream
    dg = torch.from_numpy(frames).cuda()
    dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
    dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    dn = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), st)
    minx, miny = 0.0, 0.0
    gwi, ghi = float(F(64) / F(w)), float(F(48) / F(h))
    g_off = torch.zeros((B, 64 * 48 + 1), dtype=torch.int32, device="cuda")
    g_idx = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    g_nin = torch.zeros(B, dtype=torch.int32, device="cuda")
    mt.AssignFeaturesToGrid_batch_device(dk.data_ptr(), dn.data_ptr(), cap, B, minx, miny, gwi, ghi, g_off.data_ptr(), g_idx.data_ptr(),
                                         g_nin.data_ptr(), st)
    q, lv = queries(21, nq)
    dq, dlv = torch.from_numpy(q).cuda(), torch.from_numpy(lv).cuda()
    ccap = 60000
    c_off = torch.zeros(nq + 1, dtype=torch.int32, device="cuda")
    c_cand = torch.full((ccap,), -1, dtype=torch.int32, device="cuda")
    res = [torch.zeros(nq, dtype=torch.int32, device="cuda") for _ in range(4)]
    f = 2
    mt.GetFeaturesInArea_device(dk[f].data_ptr(), g_off[f].data_ptr(), g_idx[f].data_ptr(), minx, miny, gwi, ghi, dq.data_ptr(),
                                dlv.data_ptr(), nq, c_off.data_ptr(), c_cand.data_ptr(), ccap, st)
    rc = _ffi.lib().orbfe_hamming_csr_device(mt.handle, dd[f - 1].data_ptr(), nq, dd[f].data_ptr(), c_off.data_ptr(), c_cand.data_ptr(),
                                             res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    n = dn.cpu().numpy()
    k = dk[f, :n[f]].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
    xy = np.stack([k["x"], k["y"]], 1)
    roff, ridx = oracle.assign_grid(xy, minx, miny, gwi, ghi)
    off = c_off.cpu().numpy().view(np.uint32)
    cand = c_cand.cpu().numpy().view(np.uint32)[:off[nq]]
    ro, rc_ = [0], []
    for i in range(nq):
        c = oracle.features_in_area(xy, k["octave"], roff, ridx, minx, miny, gwi, ghi, q[i, 0], q[i, 1], q[i, 2], lv[i, 0], lv[i, 1])
        rc_.extend(c.tolist())
        ro.append(len(rc_))
    assert np.array_equal(off, np.array(ro, np.uint32)) and np.array_equal(cand, np.array(rc_, np.uint32)) and off[nq] > 2000
    desc = dd.cpu().numpy()
    rbi, rb, rs = oracle.hamming_csr(desc[f - 1, :nq], desc[f, :n[f]], off, cand)
    assert np.array_equal(res[0].cpu().numpy(), rbi) and np.array_equal(res[1].cpu().numpy(), rb) and np.array_equal(res[2].cpu().numpy(), rs)
    # cap too small: the required size comes back in off[nq], the candidate buffer is not touched
    c_cand.fill_(-7)
    mt.GetFeaturesInArea_device(dk[f].data_ptr(), g_off[f].data_ptr(), g_idx[f].data_ptr(), minx, miny, gwi, ghi, dq.data_ptr(),
                                dlv.data_ptr(), nq, c_off.data_ptr(), c_cand.data_ptr(), 100, st)
    torch.cuda.synchronize()
    assert c_off.cpu().numpy().view(np.uint32)[nq] == off[nq] and bool((c_cand == -7).all())


def test_host_grid_builder_equals_the_oracle_and_the_sliced_reference():
    """orbfe_assign_grid_host (what the matcher shim uses to rebuild a KeyFrame's protected mGrid from its public mvKeysUn)
    == the oracle's AssignFeaturesToGrid == the reference's own sliced Frame::AssignFeaturesToGrid; CPU only, no handle."""
    import ctypes as C
    from orb_slam2_ssd_semantic_amd import _ffi
    from oracle import oracle_ffi as O
    from oracle import ref_ffi as R
    L = _ffi.lib()
    for sd, n in ((1, 0), (2, 1), (3, 1000), (4, 3000), (5, 2000)):
        xy, octave, minx, miny, gwi, ghi = grid_case(sd, n)
        if sd == 5:   # lens-distorted bounds: non-integer minimum, points outside the grid on every side
            minx, miny = np.float32(-7.3), np.float32(-4.9)
            xy = xy + np.float32([-30.0, -20.0])
        off, idx = np.zeros(64 * 48 + 1, np.uint32), np.zeros(max(n, 1), np.uint32)
        nin = C.c_int32()
        assert L.orbfe_assign_grid_host(_ffi.ptr(np.ascontiguousarray(xy, np.float32)), n, minx, miny, gwi, ghi, _ffi.ptr(off), _ffi.ptr(idx),
                                        C.byref(nin)) == 0
        ooff, oidx = O.assign_grid(xy, minx, miny, gwi, ghi)
        assert np.array_equal(off, ooff) and np.array_equal(idx[:nin.value], oidx) and nin.value == len(oidx)
        if R.available():
            roff, ridx = R.assign_grid(xy, minx, miny, gwi, ghi)
            assert np.array_equal(off, roff) and np.array_equal(idx[:nin.value], ridx)
