"""Randomised parity sweep of the DEVICE batch entry point: odd widths, row strides, frame strides, batch sizes, exactly
sized input buffers (GPU box).  usage: python tools/fuzz_device_api.py [ncases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle_ffi as O
from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
from orb_slam2_ssd_semantic_amd.synth import synth_frame

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for c in range(ncases):
    w, h = int(rng.integers(200, 900)), int(rng.integers(180, 700))
    B = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 24]))
    nf = int(rng.integers(100, 1500))
    stride = w + int(rng.choice([0, 0, 1, 3, 4, 7, 64]))
    fstride = stride * (h - 1) + w + int(rng.choice([0, 0, 1, 5, 64, 4096]))
    buf = np.zeros(fstride * (B - 1) + stride * (h - 1) + w, np.uint8)  # exactly sized: not one byte of slack
    imgs = []
    for i in range(B):
        img = synth_frame(int(rng.integers(0, 1 << 30)), h, w, sparse=bool(rng.integers(0, 2)))
        imgs.append(img)
        for y in range(h):
            buf[i * fstride + y * stride:i * fstride + y * stride + w] = img[y]
    try:
        e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    except Exception as ex:
        print(f"case {c}: {w}x{h} rejected: {str(ex)[:60]}")
        continue
    cap = e.capacity()
    d_gray = torch.from_numpy(buf).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_gray.data_ptr(), B, w, h, stride, fstride, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                           d_n.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    kps, desc = d_kps.cpu().numpy(), d_desc.cpu().numpy()
    oe = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    okk = True
    for i in sorted(set([0, B - 1, int(rng.integers(0, B))])):
        ok, od = oe(imgs[i], cap=cap + 64)
        gk = kps[i, :n[i]].copy().view(KP_DTYPE).reshape(-1)
        okk = okk and len(gk) == len(ok) and np.array_equal(desc[i, :n[i]], od) and all(
            np.array_equal(gk[f].view(np.uint32), ok[f].view(np.uint32)) for f in ok.dtype.names)
    if not okk:
        bad += 1
        print(f"case {c}: {w}x{h} stride {stride} fstride {fstride} B {B} nf {nf} -> MISMATCH")
print("device-api fuzz done:", ncases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
