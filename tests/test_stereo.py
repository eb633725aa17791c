"""Frame::ComputeStereoMatches (reference src/Frame.cc:642-846).

GPU: extractor + stereo matcher through the C-ABI against the oracle chain on the same synthetic stereo pair:
mvuRight / mvDepth bit-exact (float results of identical operation sequences)."""
import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd.synth import synth_frame


def stereo_pair(seed, w=640, h=480):
    """left = S(seed); right = left seen with a disparity that grows towards the bottom, plus mild noise."""
    rng = np.random.default_rng(seed)
    left = synth_frame(seed, h, w)
    right = np.empty_like(left)
    for y in range(h):
        d = 4 + (20 * y) // h
        right[y] = np.roll(left[y], -d)
    noise = rng.integers(-3, 4, left.shape)
    right = np.clip(right.astype(np.int32) + noise, 0, 255).astype(np.uint8)
    return left, right


def test_oracle_stereo_sane(oracle):
    left, right = stereo_pair(5)
    exL, exR = oracle.OracleExtractor(), oracle.OracleExtractor()
    kL, dL = exL(left)
    kR, dR = exR(right)
    u, dep, sad = oracle.stereo_matches(exL, exR, kL, dL, kR, dR, 40.0, 0.08)
    ok = u >= 0
    assert ok.sum() > 100
    disp = kL["x"][ok] - u[ok]
    expect = 4 + (20 * kL["y"][ok].astype(np.int64)) // 480
    assert np.median(np.abs(disp - expect)) < 1.0  # recovers the synthetic disparity
    assert np.allclose(dep[ok], np.float32(40.0) / disp.astype(np.float32), rtol=1e-6)
    assert np.all(dep[~ok] == -1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mbf,mb", [(5, 40.0, 0.08), (6, 386.1448, 0.537), (7, 40.0, 4.0)])
def test_gpu_stereo_parity(oracle, seed, mbf, mb):
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher
    left, right = stereo_pair(seed)
    exL, exR = oracle.OracleExtractor(), oracle.OracleExtractor()
    kL, dL = exL(left)
    kR, dR = exR(right)
    ru, rd, _ = oracle.stereo_matches(exL, exR, kL, dL, kR, dR, mbf, mb)
    gl = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    gr = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    gkL, gdL = gl(left)
    gkR, gdR = gr(right)
    assert np.array_equal(gdL, dL) and np.array_equal(gdR, dR)
    mt = ORBmatcher(0.9, True)
    u, d = mt.ComputeStereoMatches(gl, gr, gkL, gdL, gkR, gdR, mbf, mb)
    assert (ru >= 0).sum() > 50
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32))
    assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))


@pytest.mark.gpu
def test_gpu_stereo_no_matches(oracle):
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher
    left = synth_frame(8, 480, 640)
    right = synth_frame(9, 480, 640)  # unrelated image: few or no accepted matches
    gl = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    gr = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    kL, dL = gl(left)
    kR, dR = gr(right)
    exL, exR = oracle.OracleExtractor(), oracle.OracleExtractor()
    okL, odL = exL(left)
    okR, odR = exR(right)
    ru, rd, _ = oracle.stereo_matches(exL, exR, okL, odL, okR, odR, 40.0, 0.08)
    u, d = ORBmatcher(0.9, True).ComputeStereoMatches(gl, gr, kL, dL, kR, dR, 40.0, 0.08)
    assert np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))


@pytest.mark.gpu
def test_gpu_stereo_batch_device_chain(oracle):
    """Device-resident chain: two batched extractor calls (left / right images of 10 stereo pairs, one of them an unrelated
    pair) and orbfe_stereo_matches_batch_device on their output blocks, on one stream, no host buffer in between; every
    frame pair against the oracle chain (mvuRight / mvDepth bit patterns)."""
    import torch
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor, ORBmatcher
    w, h, B = 640, 480, 10
    pairs = [stereo_pair(20 + i) for i in range(B)]
    pairs[4] = (synth_frame(8, h, w), synth_frame(9, h, w))
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    gl = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    gr = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = gl.capacity()
    out = {}
    st = torch.cuda.current_stream().cuda_stream
    for name, e, img in (("L", gl, L), ("R", gr, R)):
        dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
        dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        dn = torch.zeros(B, dtype=torch.int32, device="cuda")
        e.extract_batch_device(img.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), st)
        out[name] = (dk, dd, dn)
    du = torch.full((B, cap), 7.0, dtype=torch.float32, device="cuda")
    dz = torch.full((B, cap), 7.0, dtype=torch.float32, device="cuda")
    mbf, mb = 40.0, 0.08
    ORBmatcher(0.9, True).ComputeStereoMatches_batch_device(gl, gr, out["L"][0].data_ptr(), out["L"][1].data_ptr(), out["L"][2].data_ptr(),
                                                            out["R"][0].data_ptr(), out["R"][1].data_ptr(), out["R"][2].data_ptr(), cap, B,
                                                            mbf, mb, du.data_ptr(), dz.data_ptr(), st)
    torch.cuda.synchronize()
    nL = out["L"][2].cpu().numpy()
    u, z = du.cpu().numpy(), dz.cpu().numpy()
    total = 0
    for i, (left, right) in enumerate(pairs):
        exL, exR = oracle.OracleExtractor(), oracle.OracleExtractor()
        kL, dL = exL(left)
        kR, dR = exR(right)
        assert nL[i] == len(kL)
        assert np.array_equal(out["L"][0][i, :nL[i]].cpu().numpy().copy().view(KP_DTYPE).reshape(-1).view(np.uint8), kL.view(np.uint8))
        ru, rd, _ = oracle.stereo_matches(exL, exR, kL, dL, kR, dR, mbf, mb)
        assert np.array_equal(u[i, :nL[i]].view(np.uint32), ru.view(np.uint32)), i
        assert np.array_equal(z[i, :nL[i]].view(np.uint32), rd.view(np.uint32)), i
        assert np.all(u[i, nL[i]:] == 7.0) and np.all(z[i, nL[i]:] == 7.0)   # slots past the count are not touched
        total += int((ru >= 0).sum())
    assert total > 1000
