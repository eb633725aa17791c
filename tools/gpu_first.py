"""First-contact GPU check: stage-by-stage diff of the HIP path against the oracle on one frame."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle_ffi as O
from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher
from orb_slam2_ssd_semantic_amd.synth import synth_frame

def main():
    sparse = "--sparse" in sys.argv
    img = synth_frame(0, sparse=sparse)
    oe = O.OracleExtractor()
    ok, od = oe(img)
    ge = ORBextractor()
    t = time.time(); gk, gd = ge(img); print("gpu first call s", time.time() - t)
    t = time.time(); gk, gd = ge(img); print("gpu second call s", time.time() - t)
    print("n oracle", len(ok), "n gpu", len(gk))
    for l in range(8):
        a, b = oe.level(l), ge.pyramid_level(l)
        print("level", l, a.shape, "pyr diff", int((a != b).sum()))
    for l in range(8):
        a, b = oe.blurred(l), ge.blurred_level(l)
        print("level", l, "blur diff", -1 if a is None else int((a != b).sum()))
    for l in range(8):
        a = oe.candidates(l); b = ge.candidates(l)
        aa = np.stack([a["x"], a["y"], a["response"]], 1) if len(a) else np.zeros((0, 3), np.float32)
        same = aa.shape == b.shape and np.array_equal(aa, b)
        print("level", l, "cands", len(a), len(b), "same", same)
        if not same and len(a) and len(b):
            k = min(len(aa), len(b)); d = np.nonzero((aa[:k] != b[:k]).any(1))[0]
            print("   first diffs", d[:5], aa[d[:3]], b[d[:3]])
    for l in range(8):
        a = oe.selected(l); b = ge.selected(l)
        aa = np.stack([a["x"], a["y"], a["response"]], 1) if len(a) else np.zeros((0, 3), np.float32)
        same = aa.shape == b.shape and np.array_equal(aa, b)
        print("level", l, "selected", len(a), len(b), "same", same)
        if not same:
            k = min(len(aa), len(b)); d = np.nonzero((aa[:k] != b[:k]).any(1))[0]
            print("   first diffs", d[:5], aa[d[:3]], b[d[:3]])
    if len(ok) == len(gk):
        for f in ok.dtype.names:
            print("kp field", f, "diff", int((ok[f].view(np.uint32) != gk[f].view(np.uint32)).sum()))
        print("desc diff rows", int((od != gd).any(1).sum()))
    ge.set_profiling(True); ge(img); print(ge.stage_ms())
    # matcher
    m = ORBmatcher(0.9, True)
    img2 = synth_frame(1, sparse=sparse); gk2, gd2 = ge(img2)
    r = m.MatchBruteForce(gd, gd2, gk["angle"], gk2["angle"])
    ro = O.match_bf(gd, gd2, gk["angle"], gk2["angle"], 0.9, 100, True)
    print("bf match same", [np.array_equal(x, y) for x, y in zip(r[:3], ro[:3])], r[3], ro[3])
    r = m.MatchBruteForce(gd, gd, gk["angle"], gk["angle"])
    ro = O.match_bf(gd, gd, gk["angle"], gk["angle"], 0.9, 100, True)
    print("bf self match same", [np.array_equal(x, y) for x, y in zip(r[:3], ro[:3])], r[3], ro[3])

main()
