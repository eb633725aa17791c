"""Direct sweep on the GPU box: B frames (S, sparse S, S_tum; 1000 and 2000 features; both blur roundings) through one batched
device call each, every frame compared with oracle/_ref -- the UNMODIFIED reference ORBextractor.cc compiled against the cv
stub (bump allocator, canonical cos/sin) -- without the oracle in between.  The reference side runs in a process pool.
usage: python tools/ref_sweep.py [B=320] [seed0=5000] [procs=32] [fast_mode=0]     (fast_mode: orbfe_set_fast_mode, 2 = lane-compacting)"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

w, h = 640, 480
_REF = {}


def frame(seed, i):
    from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like
    return synth_frame(seed + i, h, w, sparse=(i % 3 == 1)) if i % 3 else synth_tum_like(seed + i, h, w)


def ref_job(args):
    seed, i, nf, mode = args
    from oracle import ref_ffi as R
    R.configure(bump=True, canonical_trig=True, blur_mode=mode)
    ref = _REF.get(nf)
    if ref is None:
        ref = _REF[nf] = R.RefExtractor(nf, 1.2, 8, 20, 7)
    rk, rd = ref(frame(seed, i), cap=nf + 128)
    return i, rk.view(np.uint8).tobytes(), rd.tobytes()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    fast_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    pool = mp.get_context("spawn").Pool(procs)   # spawned before torch touches the GPU
    import torch
    from oracle import ref_ffi as R
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
    assert R.available()
    frames = np.stack(pool.starmap(frame, [(seed, i) for i in range(B)], chunksize=8))
    bad = total = 0
    for nf in (1000, 2000):
        for mode in (0, 1):
            e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, blur_rounding=mode)
            e.set_fast_mode(fast_mode)
            cap = e.capacity()
            dg = torch.from_numpy(frames).cuda()
            dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
            dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
            dn = torch.zeros(B, dtype=torch.int32, device="cuda")
            e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert e.overflow() == 0
            n, kps, desc = dn.cpu().numpy(), dk.cpu().numpy(), dd.cpu().numpy()
            del e, dg, dk, dd, dn
            t0 = time.time()
            for i, rk, rd in pool.imap_unordered(ref_job, [(seed, i, nf, mode) for i in range(B)], chunksize=4):
                gk = kps[i, :n[i]].tobytes()
                ok = gk == rk and desc[i, :n[i]].tobytes() == rd
                bad += 0 if ok else 1
                total += 1
            print("nfeatures", nf, "blur_rounding", mode, "frames", B, "mismatching frames so far", bad,
                  "ref time %.1f s on %d processes" % (time.time() - t0, procs), flush=True)
    print("TOTAL extractions", total, "mismatches", bad, "fast_mode", fast_mode)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
