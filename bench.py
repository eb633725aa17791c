#!/usr/bin/env python3
"""bench.py -- ORB extract + match frames/sec on 640x480 TUM-shaped frames (BASELINE.json metric).

One "step" = one pass of the hot path over one HBM-resident batch of synthetic frames on every rank:
    orbfe_extract_batch_device  (pyramid -> FAST -> quadtree -> blur -> IC-angle + rBRIEF)
    orbfe_match_bf_frames_device (every frame against its predecessor in the batch; config 3 parameters
                                  nnratio 0.9, TH_HIGH 100, rotation histogram on)
    [N > 1] one all-gather (RCCL over xGMI) of counts + keypoints + descriptors (config 4's exchange step)
Frames are independent, so ranks shard by construction (weak scaling: every rank owns --frames frames);
there is no collective on the data path itself.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from orb_slam2_ssd_semantic_amd.distributed import OverlappedKeyframeGather  # noqa: E402
from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher, _ffi  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frame  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def level_sizes(w, h, nlevels=8, sf=1.2):
    s = np.float32(1.0)
    out = []
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(s * np.float32(sf))
    return out


def algorithmic_bytes(w, h, nfeat, ncand):
    """bytes per frame each pass must move, independent of the implementation."""
    P = [a * b for a, b in level_sizes(w, h)]
    sp = sum(P)
    return {
        "pyramid": (sp - P[-1]) + (sp - P[0]),
        "fast": sp + 8 * ncand,
        "octree": 12 * ncand,                  # every candidate read once (4 B key) + written/read once more (8 B)
        "blur": 2 * sp,
        "describe": nfeat * (749 + 37 * 37 + 28 + 32),
    }


def make_frames(nframes, w, h, seed0):
    """Distinct synthetic frames S(seed).  A base set comes from the canonical generator; the rest
    are derived by cheap lossless transforms (roll + flip) so that every frame is a different image."""
    nbase = min(nframes, 32)
    base = [synth_frame(seed0 + i, h, w) for i in range(nbase)]
    out = np.empty((nframes, h, w), np.uint8)
    for i in range(nframes):
        b = base[i % nbase]
        k = i // nbase
        if k:
            b = np.roll(b, (37 * k) % h, axis=0)
            b = np.roll(b, (101 * k) % w, axis=1)
            if k & 1:
                b = b[:, ::-1]
        out[i] = b
    return out


def cpu_baseline(w, h, nfeat, budget_s=12.0, max_frames=200):
    """Oracle restatement ('port') of ORBextractor::operator() + BF match, 1 thread, bounded sample."""
    from oracle import oracle_ffi as O
    e = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    frames = [synth_frame(10000 + i, h, w) for i in range(8)]
    prev = None
    e(frames[0])  # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < max_frames and time.perf_counter() - t0 < budget_s:
        k, d = e(frames[n % len(frames)])
        if prev is not None:
            O.match_bf(d, prev[1], k["angle"], prev[0]["angle"], 0.9, 100, True)
        prev = (k, d)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} synthetic {w}x{h} frames, {nfeat} features, extract + BF match to previous frame, "
                      f"oracle/orb_oracle.c single thread ({dt:.1f} s)",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per step per GPU (HBM-resident batch)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU-oracle baseline and the single-frame latency probe (used for rocprof runs)")
    ap.add_argument("--include-h2d", action="store_true", help="also report the PCIe-inclusive rate (not `value`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"WORLD_SIZE={world} does not match --gpus {args.gpus}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w, h, B, nf = args.width, args.height, args.frames, args.nfeatures
    ext = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=local_rank)
    mat = ORBmatcher(0.9, True, device=local_rank)
    cap = ext.capacity()
    L = _ffi.lib()

    frames = make_frames(B, w, h, 10000 + rank * B)
    d_gray = torch.from_numpy(frames).cuda()
    # two output sets: with N > 1 the all-gather of step k overlaps the kernels of step k+1, so the set being gathered
    # must not be overwritten (N = 1 just alternates)
    outs = [(torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda"),
             torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(2)]
    d_kps, d_desc, d_n = outs[0]
    qf = torch.arange(0, B, dtype=torch.int32, device="cuda")
    tf = (torch.arange(0, B, dtype=torch.int32, device="cuda") + (B - 1)) % B  # predecessor (wraps at frame 0)
    d_match = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # (n, kps, desc) order of distributed.all_gather_keyframes
    gather = OverlappedKeyframeGather([(o[2], o[0], o[1]) for o in outs]) if world > 1 else None
    counter = [0]

    def step():
        k = counter[0] & 1
        counter[0] += 1
        kps, desc, n = outs[k]
        if gather:
            gather.acquire(k)  # set k is free again once its previous gather (two steps ago) has read it (stream-level wait)
        ext.extract_batch_device(d_gray.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap,
                                 n.data_ptr(), stream)
        if not args.no_match:
            rc = L.orbfe_match_bf_frames_device(mat.handle, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), cap,
                                                qf.data_ptr(), tf.data_ptr(), B, 0.9, 100, 1, d_match.data_ptr(),
                                                d_nm.data_ptr(), stream)
            _ffi.check(rc, "orbfe_match_bf_frames_device")
        if world > 1:
            # the one exchange step of the batched keyframe mode (distributed.all_gather_keyframes, asynchronous here):
            # RCCL runs on its own stream after the kernels above and overlaps the next step's kernels
            gather.launch(k)
        return None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ext.set_profiling(True)  # HIP events on the launch stream around every stage of every timed call
    m0 = torch.cuda.Event(enable_timing=True)
    m1 = torch.cuda.Event(enable_timing=True)
    match_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if gather:  # retire the work handles of the last two gathers (already complete: fence() synchronised the device)
        gather.acquire(0)
        gather.acquire(1)
    stage = ext.stage_ms()
    ext.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # matcher kernel time (same stream as torch's current stream, so torch events bracket it correctly)
    if not args.no_match:
        torch.cuda.synchronize()
        m0.record()
        for _ in range(5):
            L.orbfe_match_bf_frames_device(mat.handle, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                           qf.data_ptr(), tf.data_ptr(), B, 0.9, 100, 1, d_match.data_ptr(),
                                           d_nm.data_ptr(), stream)
        m1.record()
        torch.cuda.synchronize()
        match_ms = m0.elapsed_time(m1) / 5

    n_host = d_n.cpu().numpy()
    ncand = 0
    if rank == 0:
        ncand = int(sum(len(ext.candidates(l, frame=0)) for l in range(8)))

    total_frames = B * world * args.steps
    value = total_frames / elapsed
    result = None
    if rank == 0:
        ab = algorithmic_bytes(w, h, float(n_host.mean()), ncand)
        stage_k = {k: stage[k] for k in ("pyramid", "fast", "octree", "blur", "describe")}
        dom = max(stage_k, key=stage_k.get)
        # the pass the north star prices against the HBM roofline is pyramid + FAST; the dominant stage is
        # reported as `roofline`, the per-stage table and the pyramid/FAST pass ride along in `stages`.
        def gbs(name):
            return ab[name] * B / (stage_k[name] * 1e-3) / 1e9 if stage_k[name] > 0 else 0.0
        roof = {"bound": "hbm", "kernel": {"pyramid": "k_pyr_resize (7 launches)", "fast": "k_fast_map",
                                           "octree": "k_octree", "blur": "k_blur7 (8 launches)",
                                           "describe": "k_orient_describe"}[dom],
                "achieved": round(gbs(dom), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs(dom) / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_launch": int(ab[dom] * B), "launch_ms": round(stage_k[dom], 4)}
        pf_ms = stage_k["pyramid"] + stage_k["fast"]
        pf_gbs = (ab["pyramid"] + ab["fast"]) * B / (pf_ms * 1e-3) / 1e9
        stages = {k: {"ms": round(v, 4), "GBps": round(gbs(k), 2), "frac": round(gbs(k) / HBM_PEAK_GBS, 5)}
                  for k, v in stage_k.items()}
        stages["pyramid+fast"] = {"ms": round(pf_ms, 4), "GBps": round(pf_gbs, 2), "frac": round(pf_gbs / HBM_PEAK_GBS, 5)}
        stages["extract_total_ms"] = round(stage["total"], 4)
        stages["match_ms"] = round(match_ms, 4)
        if match_ms > 0:
            nn = n_host.astype(np.float64)
            evals = float((nn * np.roll(nn, 1)).sum())
            stages["match_Gdist_per_s"] = round(evals / (match_ms * 1e-3) / 1e9, 2)
        result = {
            "metric": "ORB extract+match frames/sec on 640x480 TUM RGB-D; bit-exact kp/desc",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("BASELINE config 3 shape: %dx%d synthetic TUM-shaped frames S(seed), %d features, "
                                    "8 levels, %s, HBM-resident batch of %d frames per GPU%s") %
                                   (w, h, nf, "extract only" if args.no_match else
                                    "extract + brute-force Hamming match to previous frame (nnratio 0.9, TH_HIGH 100, rot. hist.)",
                                    B, ", all-gather of counts/keypoints/descriptors" if world > 1 else ""),
                       "frames_per_gpu_per_step": B, "width": w, "height": h, "nfeatures": nf,
                       "parallelism": f"frames sharded over {world} GPU(s), one process per GPU",
                       "mean_keypoints_per_frame": round(float(n_host.mean()), 1),
                       "fast_candidates_frame0": ncand},
            "roofline": roof, "stages": stages,
        }
        if args.include_h2d:
            pin = torch.from_numpy(frames).pin_memory()
            hk = torch.empty((B, cap, 7), dtype=torch.int32).pin_memory()
            hd = torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                d_gray.copy_(pin, non_blocking=True)
                step()
                hk.copy_(d_kps, non_blocking=True)
                hd.copy_(d_desc, non_blocking=True)
            torch.cuda.synchronize()
            result["pcie_inclusive_frames_per_s"] = round(3 * B / (time.perf_counter() - t1), 2)
        if world == 1 and not args.no_cpu_baseline:
            # online (single-frame, host buffers in / out) latency of ORBextractor::operator(): replicas-only path
            e1 = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=local_rank)
            img = np.ascontiguousarray(frames[0])
            for _ in range(5):
                e1(img)
            lat = []
            for _ in range(50):
                t2 = time.perf_counter()
                e1(img)
                lat.append(time.perf_counter() - t2)
            result["single_frame_host_latency_ms"] = round(float(np.median(lat)) * 1e3, 4)
            pm = os.path.join(ROOT, "profiles", "r01_v8_pmc_hbm.json")
            if os.path.exists(pm):  # HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes
                try:
                    pj = json.load(open(pm))
                    kn = roof["kernel"].split(" ")[0]
                    # gfx950: FETCH_SIZE reports half of the read bytes (calibrated on this repo's access shapes,
                    # profiles/r01_fetch_calibration.txt), WRITE_SIZE is exact
                    tb = (2 * pj["FETCH_SIZE_KB"][kn]["mean_per_launch"] + pj["WRITE_SIZE_KB"][kn]["mean_per_launch"]) * 1024
                    roof["traffic"] = int(tb)
                    roof["traffic_note"] = ("2 x FETCH_SIZE + WRITE_SIZE per launch from profiles/r01_v8_pmc_hbm.json "
                                            "(separate --pmc passes; x2 read correction per profiles/r01_fetch_calibration.txt)")
                except Exception:
                    pass
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(w, h, nf)
            result["cpu_baseline"] = cb
            result["speedup_vs_cpu_1thread"] = round(value / cb["value"], 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
