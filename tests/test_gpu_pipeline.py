"""The sequence pipeline of the C-ABI (include/orbfe.h orbfe_pipeline_*; the reference's frame loop
perfect/Examples/RGB-D/rgbd_tum.cc:77-119: extractor on every frame, match against the previous frame -- BASELINE config 3).

CPU: the symbols are exported and the C++ host links.  GPU: every frame of a sequence equals the oracle (count, keypoint bit
patterns, descriptors, order) and every match row equals the oracle's brute-force match of frame k against frame k - 1 --
across sub-batch boundaries, across calls (ORBFE_PIPE_CONTINUE, with the caller re-using its output blocks), with and
without the closing join, for both all-pairs kernels; and a C++ host (tests/cpp/test_pipeline.cpp: HIP runtime + liborbfe.so,
no Python in the data path) drives a 3 000-frame sequence."""
import os
import struct
import subprocess
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_frames_parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_pipeline")


def build_host():
    from orb_slam2_ssd_semantic_amd import _build
    _build.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_pipeline.cpp")
    deps = [src, os.path.join(ROOT, "include", "orbfe.h"), _build.LIB]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           src, "-L", os.path.join(ROOT, "orb_slam2_ssd_semantic_amd"), "-lorbfe", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../../orb_slam2_ssd_semantic_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_pipeline_host_compiles_and_links():
    assert os.path.exists(build_host())


def test_pipeline_fails_loudly_without_a_device(have_gpu):
    import ctypes as C
    from orb_slam2_ssd_semantic_amd import _ffi
    if have_gpu:
        pytest.skip("a GPU is present: the no-device path cannot be taken here")
    L = _ffi.lib()
    p = _ffi.OrbfeParams(1000, 1.2, 8, 20, 7, 640, 480, 16, -1, 0)
    h = C.c_void_p()
    assert L.orbfe_pipeline_create(C.byref(p), 3, C.byref(h)) == _ffi.ORBFE_ERR_NODEVICE and not h.value
    assert L.orbfe_pipeline_create(C.byref(p), 0, C.byref(h)) == _ffi.ORBFE_ERR_ARG


# ---- oracle side (a pool of processes: the oracle takes ~20 ms per frame) -------------------------------------------
_OE = {}


def _oracle_frame(args):
    nf, img = args
    from oracle import oracle_ffi as O
    if nf not in _OE:
        _OE[nf] = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    k, d = _OE[nf](img)
    return k.copy(), d.copy()


def oracle_sequence(frames, nf, workers=None):
    workers = workers or min(32, os.cpu_count() or 1)
    if len(frames) <= 8 or workers <= 1:
        return [_oracle_frame((nf, f)) for f in frames]
    with ProcessPoolExecutor(workers) as ex:
        return list(ex.map(_oracle_frame, [(nf, f) for f in frames], chunksize=max(1, len(frames) // (4 * workers))))


def check_sequence(oracle, ref, n, kps, desc, match, nm, first_has_pred=None, th=100, label=""):
    """ref: oracle (keypoints, descriptors) per frame; device results as numpy blocks"""
    from orb_slam2_ssd_semantic_amd import KP_DTYPE
    for k, (ok, od) in enumerate(ref):
        nk = int(n[k])
        assert nk == len(ok), (label, k, nk, len(ok))
        gk = kps[k, :nk].copy().view(KP_DTYPE).reshape(-1)
        assert np.array_equal(gk.view(np.uint8), ok.view(np.uint8)), (label, "keypoints", k)
        assert np.array_equal(desc[k, :nk], od), (label, "descriptors", k)
        if match is None:
            continue
        pred = ref[k - 1] if k > 0 else first_has_pred
        if pred is None:
            assert int(nm[k]) == 0 and np.all(match[k, :nk] == -1), (label, "first frame has no predecessor", k)
            continue
        rm, _, _, rn = oracle.match_bf(od, pred[1], ok["angle"], pred[0]["angle"], 0.9, th, True)
        assert int(nm[k]) == rn, (label, "match count", k, int(nm[k]), rn)
        assert np.array_equal(match[k, :nk], rm), (label, "match row", k)
        assert np.all(match[k, nk:] == -1), (label, "padding slots", k)


@pytest.mark.gpu
@pytest.mark.parametrize("npipes,sub,nframes", [(3, 16, 100), (1, 32, 70), (4, 8, 33), (2, 64, 64), (3, 16, 1)])
def test_sequence_equals_oracle_across_sub_batches_and_calls(oracle, npipes, sub, nframes):
    import torch
    from orb_slam2_ssd_semantic_amd import FramePipeline
    w, h, nf = 640, 480, 1000
    frames = np.stack([synth_frame(8000 + i, h, w, sparse=(i % 5 == 3)) for i in range(nframes)])
    ref = oracle_sequence(frames, nf)
    pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=sub, npipes=npipes)
    cap = pl.capacity()
    dg = torch.from_numpy(frames).cuda()
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    dk, dd, dn = z((nframes, cap, 7), torch.int32), z((nframes, cap, 32), torch.uint8), z(nframes, torch.int32)
    dm, dnm = z((nframes, cap), torch.int32), z(nframes, torch.int32)
    st = torch.cuda.current_stream().cuda_stream

    def run(lo, hi, flags, out_off=None):
        o = lo if out_off is None else out_off
        pl.extract_match_device(dg[lo].data_ptr(), hi - lo, w, h, w, w * h, dk[o].data_ptr(), dd[o].data_ptr(), cap, dn[o:].data_ptr(),
                                dm[o].data_ptr(), dnm[o:].data_ptr(), flags=flags, stream=st)

    def host():
        torch.cuda.synchronize()
        return dn.cpu().numpy(), dk.cpu().numpy(), dd.cpu().numpy(), dm.cpu().numpy(), dnm.cpu().numpy()

    # (1) one call: a new sequence, joined
    run(0, nframes, 0)
    n, kps, desc, match, nm = host()
    assert pl.overflow() == 0
    check_sequence(oracle, ref, n, kps, desc, match, nm, label="one call")
    if nframes < 4:
        return
    # (2) the same sequence in three calls that CONTINUE; every call writes to the SAME output slots [0, ...) (a host that
    # re-uses its buffers): the carried last frame must survive that
    cuts = [0, nframes // 3, 2 * nframes // 3 + 1, nframes]
    for c in range(3):
        lo, hi = cuts[c], cuts[c + 1]
        if c == 0:
            pl.reset_sequence()
        for t in (dk, dd, dn, dm, dnm):
            t.zero_()
        run(lo, hi, pl.CONTINUE if c else 0, out_off=0)
        n, kps, desc, match, nm = host()
        check_sequence(oracle, ref[lo:hi], n, kps, desc, match, nm, first_has_pred=ref[lo - 1] if c else None, label=f"call {c}")
    # (3) NO_JOIN: two calls back to back, results only after synchronize(); popcount kernel; same answers
    pl.set_bf_kernel(1)
    pl.reset_sequence()
    for t in (dk, dd, dn, dm, dnm):
        t.zero_()
    half = nframes // 2
    run(0, half, pl.NO_JOIN)
    run(half, nframes, pl.NO_JOIN | pl.CONTINUE)
    pl.synchronize()
    n, kps, desc, match, nm = host()
    check_sequence(oracle, ref, n, kps, desc, match, nm, label="no_join + popc")
    pl.set_bf_kernel(0)
    # (4) CONTINUE after reset_sequence behaves like a new sequence; extract only leaves the match blocks alone
    pl.reset_sequence()
    dm.fill_(-7)
    dnm.fill_(-7)
    pl.extract_match_device(dg.data_ptr(), nframes, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), None, None,
                            flags=pl.CONTINUE, stream=st)
    n, kps, desc, match, nm = host()
    check_sequence(oracle, ref, n, kps, desc, None, None, label="extract only")
    assert np.all(match == -7) and np.all(nm == -7)
    assert pl.overflow() == 0


@pytest.mark.gpu
def test_pipeline_equals_single_handle_calls(oracle):
    """the pipeline is the plain entry points in another launch order: same bytes as orbfe_extract_batch_device +
    orbfe_match_bf_frames_device on one handle, 2000 features"""
    import torch
    from orb_slam2_ssd_semantic_amd import FramePipeline, ORBextractor, ORBmatcher, _ffi
    w, h, nf, N = 640, 480, 2000, 48
    frames = np.stack([synth_frame(8600 + i, h, w) for i in range(N)])
    dg = torch.from_numpy(frames).cuda()
    pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=16, npipes=3)
    cap = pl.capacity()
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    out = [(z((N, cap, 7), torch.int32), z((N, cap, 32), torch.uint8), z(N, torch.int32), z((N, cap), torch.int32), z(N, torch.int32))
           for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    k, d, n, m, nm = out[0]
    pl.extract_match_device(dg.data_ptr(), N, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), m.data_ptr(), nm.data_ptr(),
                            stream=st)
    e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=N)
    mt = ORBmatcher(0.9, True)
    assert e.capacity() == cap
    k2, d2, n2, m2, nm2 = out[1]
    e.extract_batch_device(dg.data_ptr(), N, w, h, w, w * h, k2.data_ptr(), d2.data_ptr(), cap, n2.data_ptr(), st)
    qf = torch.arange(1, N, dtype=torch.int32, device="cuda")
    tf = qf - 1
    _ffi.check(_ffi.lib().orbfe_match_bf_frames_device(mt.handle, k2.data_ptr(), d2.data_ptr(), n2.data_ptr(), cap, qf.data_ptr(),
                                                       tf.data_ptr(), N - 1, 0.9, 100, 1, m2[1].data_ptr(), nm2[1:].data_ptr(), st), "match")
    m2[0].fill_(-1)
    torch.cuda.synchronize()
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert int(n.min()) > 1500


@pytest.mark.gpu
def test_cpp_host_drives_a_3000_frame_sequence(oracle, tmp_path):
    """a C++ host calls orbfe_pipeline_extract_match_device on a 3 000-frame sequence (5 calls of 600 frames that
    continue each other, 3 pipes x 256-frame sub-batches, NO_JOIN + explicit join, device output blocks re-used by every call)
    and every frame and every match row equals the oracle."""
    exe = build_host()
    w, h, nf, N = 640, 480, 1000, 3000
    nseed = 250   # distinct generator seeds; the rest are lossless roll / flip transforms (every frame a different image)
    base = synth_frames_parallel("S", nseed, h, w, 91000)
    frames = np.empty((N, h, w), np.uint8)
    for i in range(N):
        b, k = base[i % nseed], i // nseed
        f = np.roll(b, ((37 * k) % h, (101 * k) % w), axis=(0, 1)) if k else b
        frames[i] = f[:, ::-1] if k & 1 else f
    raw, out = tmp_path / "seq.raw", tmp_path / "seq.out"
    frames.tofile(raw)
    r = subprocess.run([exe, str(raw), str(w), str(h), str(N), str(nf), "256", "3", "5", str(out), "1"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = oracle_sequence(frames, nf, workers=min(64, os.cpu_count() or 1))
    blob = open(out, "rb").read()
    pos = 0
    KP = oracle.KP_DTYPE
    for k, (ok, od) in enumerate(ref):
        n = struct.unpack_from("<i", blob, pos)[0]
        pos += 4
        assert n == len(ok), (k, n, len(ok))
        gk = np.frombuffer(blob, KP, n, pos)
        pos += 28 * n
        gd = np.frombuffer(blob, np.uint8, 32 * n, pos).reshape(n, 32)
        pos += 32 * n
        nm = struct.unpack_from("<i", blob, pos)[0]
        pos += 4
        gm = np.frombuffer(blob, np.int32, n, pos)
        pos += 4 * n
        assert np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(gd, od), ("frame", k)
        if k == 0:
            assert nm == 0 and np.all(gm == -1)
        elif k % 7 == 0 or (k % 600) % 256 == 0:   # every seventh pair + every sub-batch / call boundary
            pk, pd = ref[k - 1]
            rm, _, _, rn = oracle.match_bf(od, pd, ok["angle"], pk["angle"], 0.9, 100, True)
            assert nm == rn and np.array_equal(gm, rm), ("matches", k)
    assert pos == len(blob)


def _read_host_dump(blob, nframes, KP):
    pos, out = 0, []
    for _ in range(nframes):
        n = struct.unpack_from("<i", blob, pos)[0]
        pos += 4
        gk = np.frombuffer(blob, KP, n, pos)
        pos += 28 * n
        gd = np.frombuffer(blob, np.uint8, 32 * n, pos).reshape(n, 32)
        pos += 32 * n
        nm = struct.unpack_from("<i", blob, pos)[0]
        pos += 4
        gm = np.frombuffer(blob, np.int32, n, pos)
        pos += 4 * n
        out.append((gk, gd, nm, gm))
    assert pos == len(blob)
    return out


@pytest.mark.gpu
def test_host_entry_point_from_cpp_and_python(oracle, tmp_path):
    """orbfe_pipeline_extract_match: host frames in, host results out (BASELINE config 3 as SURVEY 8(d) words it), copies and
    pipes overlapped inside the library.  (a) a C++ host with plain malloc'ed (pageable) buffers, 2 calls x 150 frames, 4 pipes x
    32-frame chunks; (b) the ctypes mirror with strided page-locked frames and a `cap` larger than the pipeline's.  Every frame and
    every match row against the oracle."""
    import ctypes as C
    import torch
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, FramePipeline
    exe = build_host()
    w, h, nf, N = 640, 480, 1000, 300
    frames = np.stack([synth_frame(9700 + i, h, w, sparse=(i % 7 == 2)) for i in range(N)])
    ref = oracle_sequence(frames, nf)
    raw, out = tmp_path / "h.raw", tmp_path / "h.out"
    frames.tofile(raw)
    r = subprocess.run([exe, str(raw), str(w), str(h), str(N), str(nf), "32", "4", "2", str(out), "0", "1"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for k, ((gk, gd, nm, gm), (ok, od)) in enumerate(zip(_read_host_dump(open(out, "rb").read(), N, oracle.KP_DTYPE), ref)):
        assert len(gk) == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(gd, od), ("frame", k)
        if k == 0:
            assert nm == 0 and np.all(gm == -1)
        else:
            rm, _, _, rn = oracle.match_bf(od, ref[k - 1][1], ok["angle"], ref[k - 1][0]["angle"], 0.9, 100, True)
            assert nm == rn and np.array_equal(gm, rm), ("matches", k)
    # (b) ctypes, frames with a row pitch of w + 64 in page-locked memory, cap = capacity + 64
    pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=16, npipes=3)
    M, stride = 70, w + 64
    buf = torch.zeros((M, h, stride), dtype=torch.uint8).pin_memory()
    buf[:, :, :w] = torch.from_numpy(frames[:M])
    cap = pl.capacity() + 64
    hk = torch.zeros((M, cap, 7), dtype=torch.int32).pin_memory()
    hd = torch.zeros((M, cap, 32), dtype=torch.uint8).pin_memory()
    hn, hnm = torch.zeros(M, dtype=torch.int32).pin_memory(), torch.zeros(M, dtype=torch.int32).pin_memory()
    hm = torch.zeros((M, cap), dtype=torch.int32).pin_memory()
    ptrs = (C.c_void_p * M)(*[buf[i].data_ptr() for i in range(M)])
    pl.extract_match(ptrs, M, w, h, stride, hk.data_ptr(), hd.data_ptr(), cap, hn.data_ptr(), hm.data_ptr(), hnm.data_ptr())
    pc = pl.capacity()   # the library fills the pipeline's own capacity of every row; the caller's extra columns stay untouched
    check_sequence(oracle, ref[:M], hn.numpy(), hk.numpy(), hd.numpy(), hm.numpy()[:, :pc], hnm.numpy(), label="host, strided")
    assert not hm.numpy()[:, pc:].any() and not hk.numpy()[:, pc:].any()


@pytest.mark.gpu
def test_many_short_continuing_calls_take_turns_on_the_pipes(oracle):
    """40 calls of 1 - 7 frames (sub-batch 4, 3 pipes: consecutive calls start on different pipes), each continuing the sequence
    and each writing into the SAME output slots, nothing synchronised between the calls but the final read-back of each call's
    rows on the launch stream: the carried last frame, the frame-0 match against it and the re-use of the caller's blocks are
    ordered by the pipeline's events alone."""
    import torch
    from orb_slam2_ssd_semantic_amd import FramePipeline
    w, h, nf = 640, 480, 1000
    rng = np.random.default_rng(5)
    sizes = [int(v) for v in rng.integers(1, 8, 40)]
    N = sum(sizes)
    frames = np.stack([synth_frame(8800 + i, h, w, sparse=(i % 4 == 1)) for i in range(N)])
    ref = oracle_sequence(frames, nf)
    pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=4, npipes=3)
    cap = pl.capacity()
    dg = torch.from_numpy(frames).cuda()
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    dk, dd, dn, dm, dnm = z((8, cap, 7), torch.int32), z((8, cap, 32), torch.uint8), z(8, torch.int32), z((8, cap), torch.int32), z(8, torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    got, lo = [], 0
    for c, n in enumerate(sizes):
        pl.extract_match_device(dg[lo].data_ptr(), n, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), dm.data_ptr(), dnm.data_ptr(),
                                flags=pl.CONTINUE if c else 0, stream=st)
        # asynchronous copies on the launch stream (ordered behind the call's join); the next call overwrites the blocks at once
        got.append(tuple(t[:n].to("cpu", non_blocking=False).numpy().copy() for t in (dn, dk, dd, dm, dnm)))
        lo += n
    lo = 0
    for c, n in enumerate(sizes):
        cn, ck, cd, cm, cnm = got[c]
        check_sequence(oracle, ref[lo:lo + n], cn, ck, cd, cm, cnm, first_has_pred=ref[lo - 1] if c else None, label=f"call {c} ({n} frames)")
        lo += n
    assert pl.overflow() == 0


@pytest.mark.gpu
def test_pipeline_at_config5_and_config4_shapes_equals_one_batched_call():
    """BASELINE config 5 (1920x1080, 4000 features) and config 4 (640x480, 2000 features) through the pipeline, extract only:
    byte-identical to one batched call on one handle (which tests/test_gpu_extract.py pins to the oracle at these shapes)."""
    import torch
    from orb_slam2_ssd_semantic_amd import FramePipeline, ORBextractor
    for (w, h, nf, N, sub, pipes) in ((1920, 1080, 4000, 40, 8, 3), (640, 480, 2000, 100, 16, 4)):
        frames = np.stack([synth_frame(9900 + i, h, w, sparse=(i % 6 == 5)) for i in range(N)])
        dg = torch.from_numpy(frames).cuda()
        e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=N)
        pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=sub, npipes=pipes)
        cap = e.capacity()
        assert pl.capacity() == cap
        st = torch.cuda.current_stream().cuda_stream
        outs = []
        for use_pl in (False, True):
            k = torch.zeros((N, cap, 7), dtype=torch.int32, device="cuda")
            d = torch.zeros((N, cap, 32), dtype=torch.uint8, device="cuda")
            n = torch.zeros(N, dtype=torch.int32, device="cuda")
            if use_pl:
                pl.extract_match_device(dg.data_ptr(), N, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), None, None, stream=st)
            else:
                e.extract_batch_device(dg.data_ptr(), N, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append((k, d, n))
        assert e.overflow() == 0 and pl.overflow() == 0
        for a, b in zip(*outs):
            assert torch.equal(a, b), (w, h)
        assert int(outs[0][2].min()) > 0.9 * nf
        pl.close()
        e.close()


@pytest.mark.gpu
def test_unjoined_calls_over_shifted_and_differently_sized_slices_are_ordered_by_address(oracle):
    """ORBFE_PIPE_NO_JOIN calls whose output slices do not line up with the previous call's sub-batch indices: a ring of
    output slots written by calls of varying size at a moving offset -- slices recorded under index i of one call are overwritten
    under index j != i of a later one -- and extract-only calls in between (no frame-0 match: the carry slot's write-after-write).
    Nothing is joined until the end of a lap; every frame's rows must be the oracle's, whichever call wrote them last."""
    import torch
    from orb_slam2_ssd_semantic_amd import FramePipeline
    w, h, nf, sub = 640, 480, 1000, 4
    rng = np.random.default_rng(9)
    N = 48
    frames = np.stack([synth_frame(9100 + i, h, w, sparse=(i % 3 == 1)) for i in range(N)])
    ref = oracle_sequence(frames, nf)
    pl = FramePipeline(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=sub, npipes=3)
    cap = pl.capacity()
    dg = torch.from_numpy(frames).cuda()
    R = 20   # ring of output slots
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    dk, dd, dn, dm, dnm = z((R, cap, 7), torch.int32), z((R, cap, 32), torch.uint8), z(R, torch.int32), z((R, cap), torch.int32), z(R, torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    for lap in range(3):
        owner = {}          # ring slot -> (frame index, has match row, predecessor frame or None)
        lo, off, c = 0, int(rng.integers(0, R)), 0
        pl.reset_sequence()
        while lo < N:
            n = min(int(rng.integers(1, 11)), N - lo)
            if off + n > R:
                off = int(rng.integers(0, 3))          # wrap: the new slices straddle old ones at another alignment
            with_match = (c % 4) != 2                  # every fourth call is extract only
            pl.extract_match_device(dg[lo].data_ptr(), n, w, h, w, w * h, dk[off].data_ptr(), dd[off].data_ptr(), cap, dn[off:].data_ptr(),
                                    dm[off].data_ptr() if with_match else None, dnm[off:].data_ptr() if with_match else None,
                                    flags=pl.NO_JOIN | (pl.CONTINUE if c else 0), stream=st)
            for i in range(n):
                prev = lo + i - 1
                # a call behind an extract-only call has no carried predecessor for its frame 0?  No: the carry is the previous
                # call's last frame whether or not that call matched; only the very first call of the lap has none
                owner[off + i] = (lo + i, with_match, prev if prev >= 0 else None)
            lo += n
            off += n
            c += 1
        pl.synchronize()
        torch.cuda.synchronize()
        hn, hk, hd, hm, hnm = (t.cpu().numpy() for t in (dn, dk, dd, dm, dnm))
        for slot, (f, with_match, prev) in owner.items():
            pred = ref[prev] if prev is not None else None
            check_sequence(oracle, [ref[f]], hn[slot:slot + 1], hk[slot:slot + 1], hd[slot:slot + 1], hm[slot:slot + 1] if with_match else None,
                           hnm[slot:slot + 1] if with_match else None, first_has_pred=pred, label=f"lap {lap} slot {slot} frame {f}")
        assert pl.overflow() == 0
    pl.close()
