"""Probe: does running two half-batch extractor pipelines on two streams beat one full-batch pipeline?  (GPU box)
usage: python tools/two_stream_probe.py"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import base_frames, expand_frames

w, h, B = 640, 480, 1024
fr = expand_frames(torch.from_numpy(base_frames("S", 32, w, h, 10000)).cuda(), B)


def make(nb):
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nb)
    cap = e.capacity()
    return e, cap, torch.zeros((nb, cap, 7), dtype=torch.int32, device="cuda"), torch.zeros((nb, cap, 32), dtype=torch.uint8, device="cuda"), \
        torch.zeros(nb, dtype=torch.int32, device="cuda")


def timed(f, reps=20):
    f(3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f(reps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


one = make(B)


def run_one(n):
    e, cap, dk, dd, dn = one
    for _ in range(n):
        e.extract_batch_device(fr.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), None)


print("one stream, 1024 frames per call: %.4f ms per 1024 frames" % timed(run_one))
for parts in (2, 4):
    nb = B // parts
    hs = [make(nb) for _ in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]

    def run_many(n):
        cur = torch.cuda.current_stream()
        for s in ss:
            s.wait_stream(cur)
        for _ in range(n):
            for i, (hh, s) in enumerate(zip(hs, ss)):
                e, cap, dk, dd, dn = hh
                e.extract_batch_device(fr[i * nb:].data_ptr(), nb, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), s.cuda_stream)
        for s in ss:
            cur.wait_stream(s)

    print("%d streams, %d frames per call each: %.4f ms per 1024 frames" % (parts, nb, timed(run_many)))
