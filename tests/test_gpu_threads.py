"""The three SLAM threads use the matcher at the same time ("the C-ABI must be thread-safe").

In the reference Tracking (src/Tracking.cc:1346 SearchByProjection), LocalMapping (src/LocalMapping.cc:422 SearchForTriangulation,
:697 Fuse) and LoopClosing (src/LoopClosing.cc:342 SearchByBoW, :418 SearchBySim3) each construct ORBmatcher objects and call
them concurrently, while the Frame constructor runs the extractor(s).  Here: three Python threads (ctypes releases the GIL for
the duration of a call) loop over those members through oracle/_ref/libshim_full.so -- the product's matcher shim as the ONLY
ORBmatcher translation unit, one thread_local C-ABI matcher handle per thread -- while two more threads drive two extractor
shim objects through the reference's Frame::ExtractORB.  Every single result must equal the reference's compiled body,
computed beforehand on one thread."""
import threading

import numpy as np
import pytest

import proj_cases as PC
from oracle import ref_ffi as R

needs = pytest.mark.skipif(not (R.available() and R.shim_available()), reason="oracle/_ref libraries not built and /root/reference absent")


def _eq(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_eq(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, np.asarray(b))
    return a == b


@needs
@pytest.mark.gpu
def test_three_matcher_threads_and_two_extractor_threads_run_concurrently():
    from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like
    from test_ref_pin import _bow_case
    rng = np.random.default_rng(77_000)
    ITER = 220
    R.configure(bump=True, canonical_trig=True, blur_mode=1)

    # ---- the work items of every thread: (callable(shim) -> result) and the reference's answer, computed here ----------------
    def items_tracking():
        out = []
        for k in range(6):
            cur, last = PC.last_frame_case(rng, 1000, 1000, ["small", "forward", "backward"][k % 3], stereo=(k % 2 == 0))
            out.append(lambda sh, c=cur, l=last, m=(k % 2 == 1): R.search_by_projection_last_frame(c, l, 15.0, m, shim=sh))
            cur2, mps = PC.local_map_case(rng, 1000, 2500)
            out.append(lambda sh, c=cur2, m=mps: R.search_by_projection_local_map(c, m, 3.0, 0.8, shim=sh))
        return out

    def items_mapping():
        out = []
        for k in range(5):
            k1, k2, F12 = PC.triangulation_case(rng, 1000, 1000, 80)
            out.append(lambda sh, a=k1, b=k2, f=F12: R.search_for_triangulation(a, b, f, False, shim=sh))
            kf2, mps2 = PC.fuse_case(rng, 1000, 1500)
            out.append(lambda sh, a=kf2, b=mps2: R.fuse(a, b, 3.0, shim=sh))
        return out

    def items_loop():
        out = []
        for k in range(5):
            (d1, v1, a1, fv1), (d2, v2, a2, fv2) = _bow_case(rng, 1000, 1000, 100, 0.8, k % 2)
            out.append(lambda sh, A=(d1, v1, a1, fv1, d2, a2, fv2): R.search_by_bow_kf_f(*A, 0.75, True, shim=sh))
            out.append(lambda sh, A=(d1, v1, a1, fv1, d2, v2, a2, fv2): R.search_by_bow_kf_kf(*A, 0.75, True, shim=sh))
            s1, s2, s12, R12, t12, m_in = PC.sim3_pair_case(rng, 1000, 1000)
            out.append(lambda sh, A=(s1, s2, s12, R12, t12), M=m_in: R.search_by_sim3(*A, 7.5, M, shim=sh))
        return out

    work = {"tracking": items_tracking(), "mapping": items_mapping(), "loop": items_loop()}
    want = {name: [fn(False) for fn in fns] for name, fns in work.items()}
    for name, fns in work.items():          # one warm call per item on this thread: libraries loaded, handles of THIS thread made
        for fn, w in zip(fns, want[name]):
            assert _eq(fn("full"), w), name

    frames = [synth_frame(900), synth_tum_like(901), synth_frame(902, sparse=True), synth_frame(903)]
    rext = R.RefExtractor(1000, 1.2, 8, 20, 7)
    want_ext = [rext(f, cap=1200) for f in frames]
    shims = [R.ShimExtractor(1000, 1.2, 8, 20, 7), R.ShimExtractor(1000, 1.2, 8, 20, 7)]   # two instances: the stereo case
    for s in shims:
        k, d = s.extract_via_frame(frames[0], cap=1200)
        assert np.array_equal(k.view(np.uint8), want_ext[0][0].view(np.uint8)) and np.array_equal(d, want_ext[0][1])

    errors, counts = [], {}
    start = threading.Barrier(5)

    def matcher_thread(name):
        fns, ws = work[name], want[name]
        start.wait()
        n = 0
        try:
            for it in range(ITER):
                i = it % len(fns)
                if not _eq(fns[i]("full"), ws[i]):
                    errors.append((name, it, i))
                    break
                n += 1
        except BaseException as e:   # noqa: BLE001
            errors.append((name, "exception", repr(e)))
        counts[name] = n

    def extractor_thread(idx):
        s = shims[idx]
        start.wait()
        n = 0
        try:
            for it in range(ITER):
                i = (it + idx) % len(frames)
                k, d = s.extract_via_frame(frames[i], left=(idx == 0), cap=1200)
                if not (np.array_equal(k.view(np.uint8), want_ext[i][0].view(np.uint8)) and np.array_equal(d, want_ext[i][1])):
                    errors.append(("extractor", idx, it))
                    break
                n += 1
        except BaseException as e:   # noqa: BLE001
            errors.append(("extractor", idx, repr(e)))
        counts[f"extractor{idx}"] = n

    threads = [threading.Thread(target=matcher_thread, args=(n,)) for n in work] + \
              [threading.Thread(target=extractor_thread, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors
    assert all(v == ITER for v in counts.values()) and len(counts) == 5, counts
    R.configure(bump=True, canonical_trig=True, blur_mode=0)
