"""Direct sweep on the GPU box: B frames (S, sparse S, S_tum; 1000 and 2000 features) through one batched device call each,
every frame compared with oracle/_ref -- the UNMODIFIED reference ORBextractor.cc compiled against the cv stub (bump
allocator, canonical cos/sin) -- without the oracle in between.  usage: python tools/ref_sweep.py [B=320]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_ffi as R
from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like
assert R.available()
R.configure(bump=True, canonical_trig=True, blur_mode=0)
w, h, B = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 320
frames = np.stack([synth_frame(5000 + i, h, w, sparse=(i % 3 == 1)) if i % 3 else synth_tum_like(5000 + i, h, w) for i in range(B)])
bad = 0
for nf in (1000, 2000):
    e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = e.capacity()
    dg = torch.from_numpy(frames).cuda()
    dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda"); dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); dn = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert e.overflow() == 0
    n, kps, desc = dn.cpu().numpy(), dk.cpu().numpy(), dd.cpu().numpy()
    ref = R.RefExtractor(nf, 1.2, 8, 20, 7)
    t0 = time.time()
    for i in range(B):
        rk, rd = ref(frames[i], cap=nf + 128)
        gk = kps[i, :n[i]].copy().view(KP_DTYPE).reshape(-1)
        ok = n[i] == len(rk) and np.array_equal(gk.view(np.uint8), rk.view(np.uint8)) and np.array_equal(desc[i, :n[i]], rd)
        bad += 0 if ok else 1
    print(nf, "frames", B, "mismatching frames so far", bad, "ref time %.1f s" % (time.time() - t0), flush=True)
print("TOTAL mismatches", bad)
