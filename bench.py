#!/usr/bin/env python3
"""bench.py -- ORB extract + match frames/sec on 640x480 TUM-shaped frames (BASELINE.json metric).

One "step" = one pass of the hot path over one HBM-resident batch of synthetic frames on every rank, issued as
`--launches` back-to-back sub-batches of `--frames` frames:
    orbfe_extract_batch_device   (pyramid -> FAST -> quadtree -> blur -> IC-angle + rBRIEF)
    orbfe_match_bf_frames_device (every frame against its predecessor in the sub-batch; config 3 parameters
                                  nnratio 0.9, TH_HIGH 100, rotation histogram on)
    [N > 1] one asynchronous all-gather (RCCL over xGMI) of counts + keypoints + descriptors of the whole step
Frames are independent, so ranks shard by construction (weak scaling: every rank owns its own batch); there is no
collective on the data path itself.

`value` is the HBM-resident rate (inputs in HBM when the timed region starts); the brute-force matcher of sub-batch j runs on a
second stream behind an event, so it overlaps the pyramid of sub-batch j + 1 (+1.6 %).  The same JSON line also carries
  pcie_inclusive        the contract's config 3 as SURVEY 8(d) words it: pinned host frames -> H2D -> kernels -> D2H of
                        counts / keypoints / descriptors / matches, double-buffered on three streams (never `value`)
  workloads             S(seed) (corner-saturated) and S_tum(seed) (camera-like corner statistics), both FAST variants
  config4               the batched-keyframe configuration: 2000 features, a 1024-frame batch sharded over the ranks,
                        all-gather bytes and bus bandwidth, strong and weak figures
  roofline / cpu_baseline / cpu_baseline_all_cores

    python bench.py --gpus N --steps K --warmup W        (N > 1 re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  oracle/ is used only for the cpu_baseline legs (rank 0, N = 1).
`--fake` replaces the extractor by a CPU stand-in over gloo: it exists so that tests/ can drive the spawn /
rendezvous / gather / reporting path on a box without GPUs; its line is marked "fake": true.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# Every extractor handle runs its blur on a side stream and this script adds matcher, copy and collective streams of its
# own (a dozen over all legs); the ROCm runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues
# (default 4), and two streams that land on one queue serialise -- measured here: the PCIe-inclusive leg drops from 177 k
# to 95-118 k frames/s when a copy stream shares a queue with a kernel stream (4 and 8 queues), 16 keep them apart
# ("streams").  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CTL = {}   # control-plane process groups of a multi-GPU run (see main) and the `abandon` flag of the guarded group leg
from orb_slam2_ssd_semantic_amd.distributed import OverlappedKeyframeGather, all_gather_keyframes, shard_range  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import regular_vocabulary, synth_frame, synth_frames_parallel, synth_tum_like  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_SIMD = 1024            # 256 CUs x 4 SIMDs
# The round in which a kernel of the extractor chain last changed (round 6: k_blur7, k_pyr_walk).  Counter files under profiles/
# are named rNN_*: figures from a file older than this round are NOT attached to a line -- the field says "stale" instead.
KERNEL_ROUND = 6


def profile_round(name):
    try:
        return int(name[1:3]) if name and name[0] == "r" else 0
    except ValueError:
        return 0


def profile_for_shape(suffix, shape):
    """Newest committed profiles/rNN_*<suffix> whose recorded shape (width, height, nfeatures, workload,
    frames_per_launch) equals `shape`.  Files of rounds 1-2 carry no "shape" key: they were taken with the default command
    (640x480, 1000 features, S, their own frames_per_launch).  Counter figures are never rescaled to another shape."""
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir), reverse=True):
        if not name.endswith(suffix):
            continue
        try:
            pj = json.load(open(os.path.join(pdir, name)))
        except Exception:
            continue
        have = pj.get("shape") or {"width": 640, "height": 480, "nfeatures": 1000, "workload": "S",
                                   "frames_per_launch": pj.get("frames_per_launch")}
        if all(have.get(k) == shape[k] for k in ("width", "height", "nfeatures", "workload", "frames_per_launch")):
            return name, pj
    return None, None


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


def expand_frames(base, total):
    """`total` distinct frames from a base set [nb, h, w] by lossless transforms (roll + flip), on whatever device
    `base` lives on.  Every frame is a different image with the statistics of its generator."""
    nb, h, w = base.shape
    out = torch.empty((total, h, w), dtype=torch.uint8, device=base.device)
    k = 0
    while k * nb < total:
        blk = base
        if k:
            blk = torch.roll(blk, shifts=((37 * k) % h, (101 * k) % w), dims=(1, 2))
            if k & 1:
                blk = blk.flip(2)
        n = min(nb, total - k * nb)
        out[k * nb:k * nb + n] = blk[:n]
        k += 1
    return out


def base_frames(gen, n, w, h, seed0):
    if gen == "TUM":   # the real sequence, when $TUM_FR3_WALKING_XYZ points at it (gray conversion of src/Tracking.cc:342)
        from orb_slam2_ssd_semantic_amd import tum
        fr = tum.load_gray_frames(limit=n)
        assert fr.shape[1:] == (h, w), f"TUM frames are {fr.shape[1:]}, bench asked for {(h, w)}"
        return fr
    return synth_frames_parallel(gen, n, h, w, seed0)   # n distinct generator seeds, a pool of host processes


# ----------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only).  oracle/ is test infrastructure: it is timed here, never used by the product.
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, h, nfeat, budget_s=16.0, nframes=200, warmup=20):
    """ORBextractor::operator() + BF match to the previous frame, 1 thread, as SURVEY 8(d) asks: 20 warm-up frames, then the
    MEDIAN per-frame time of 200 frames (bounded by budget_s of wall time; the sample says how many were timed).
    kind "reference": the extractor is oracle/_ref/libref_orb.so = the UNMODIFIED reference src/ORBextractor.cc compiled
    against a cv stub whose five OpenCV primitives are scalar C restatements (so slower than a real OpenCV build); the
    brute-force match (not a reference function) is the oracle's.  Falls back to kind "port" (oracle/orb_oracle.c) without the
    library."""
    from oracle import oracle_ffi as O
    kind, e = "port", None
    try:
        from oracle import ref_ffi as R
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_orb.so")):
            R.configure(bump=True, canonical_trig=True, blur_mode=0)
            e = R.RefExtractor(nfeat, 1.2, 8, 20, 7)
            kind = "reference"
    except Exception:
        e = None
    if e is None:
        e = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    frames = [synth_frame(10000 + i, h, w) for i in range(8)]
    prev = None
    times = []
    t_start = time.perf_counter()
    n = 0
    while len(times) < nframes and time.perf_counter() - t_start < budget_s:
        t0 = time.perf_counter()
        k, d = e(frames[n % len(frames)])
        if prev is not None:
            O.match_bf(d, prev[1], k["angle"], prev[0]["angle"], 0.9, 100, True)
        dt = time.perf_counter() - t0
        prev = (k, d)
        n += 1
        if n > warmup:
            times.append(dt)
    med = float(np.median(times))
    what = ("oracle/_ref (unmodified reference ORBextractor.cc, cv stub with scalar OpenCV primitives) + oracle BF match"
            if kind == "reference" else "oracle/orb_oracle.c")
    return {"value": round(1.0 / med, 3), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"median per-frame time of {len(times)} synthetic {w}x{h} frames S(seed) after {warmup} warm-up frames, {nfeat} "
                      f"features, extract + BF match to previous frame, {what}, single thread "
                      f"({time.perf_counter() - t_start:.1f} s; mean {1.0 / float(np.mean(times)):.2f} frames/s)",
            "host_cpus": os.cpu_count()}


def host_cpu_report():
    """what the CPU figures ran on: logical CPUs, the CPUs this process may run on, the cgroup quota"""
    rep = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            rep["cgroup_" + os.path.basename(path)] = open(path).read().strip()
        except OSError:
            pass
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")]
        rep["cpu_model"] = model[0] if model else None
    except OSError:
        pass
    q = rep.get("cgroup_cpu.max", "max").split()
    rep["usable_cpus"] = rep["affinity_cpus"] or rep["logical_cpus"]
    if q and q[0] != "max" and len(q) == 2:
        rep["usable_cpus"] = max(1, min(rep["usable_cpus"], int(float(q[0]) / float(q[1]))))
    return rep


def cpu_baseline_vectorised(w, h, nfeat, budget_s=8.0, nframes=120, warmup=10):
    """The same measurement as cpu_baseline on oracle/_ref/libref_orb_vec.so: the same unmodified reference ORBextractor.cc and the
    same five OpenCV stand-ins, built -O3 -mavx2 (auto-vectorised, -ffp-contract=off) -- the nearest thing to "the reference
    over a real SIMD OpenCV" this box allows.  Its outputs are asserted equal to the -O2 build's before it is timed."""
    from oracle import oracle_ffi as O
    try:
        from oracle import ref_ffi as R
        if not (R.vectorised_available() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_orb.so"))):
            return None
    except Exception:
        return None
    frames = [synth_frame(10000 + i, h, w) for i in range(8)]
    R.configure(bump=True, canonical_trig=True, blur_mode=0)
    k0, d0 = R.RefExtractor(nfeat, 1.2, 8, 20, 7)(frames[0])
    with R.use_vectorised():
        R.configure(bump=True, canonical_trig=True, blur_mode=0)
        e = R.RefExtractor(nfeat, 1.2, 8, 20, 7)
        k1, d1 = e(frames[0])
        if not (np.array_equal(k0.view(np.uint8), k1.view(np.uint8)) and np.array_equal(d0, d1)):
            raise SystemExit("bench.py: the -O3 -mavx2 build of the reference differs from the -O2 build")
        prev, times, n = None, [], 0
        t_start = time.perf_counter()
        while len(times) < nframes and time.perf_counter() - t_start < budget_s:
            t0 = time.perf_counter()
            k, d = e(frames[n % len(frames)])
            if prev is not None:
                O.match_bf(d, prev[1], k["angle"], prev[0]["angle"], 0.9, 100, True)
            dt = time.perf_counter() - t0
            prev = (k, d)
            n += 1
            if n > warmup:
                times.append(dt)
        del e
    med = float(np.median(times))
    return {"value": round(1.0 / med, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": f"median per-frame time of {len(times)} synthetic {w}x{h} frames S(seed) after {warmup} warm-up frames, {nfeat} features, "
                      "extract + BF match to previous frame, oracle/_ref/libref_orb_vec.so (unmodified reference ORBextractor.cc + the cv "
                      "stub's OpenCV stand-ins, g++ -O3 -mavx2 -ffp-contract=off; outputs asserted identical to the -O2 build) + oracle BF "
                      "match, single thread", "host": host_cpu_report()}


def cpu_baseline_all_cores(w, h, nfeat, duration_s=6.0, max_procs=None):
    """One frame stream per host CPU -- ALL of them, as SURVEY 8(d) asks (the reference extractor is serial per call):
    independent worker processes (oracle/cpu_worker.py), all started on a common wall-clock tick.  kind "reference" when
    oracle/_ref/libref_orb.so is there (the unmodified reference ORBextractor.cc over the cv stub), else "port"."""
    host = host_cpu_report()
    procs = host["usable_cpus"] or os.cpu_count() or 1   # the CPUs this process may actually use (affinity mask, cgroup quota)
    if max_procs:
        procs = min(procs, max_procs)
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_orb.so")) else "port"
    t_go = time.time() + 8.0 + procs / 32.0   # every interpreter has to be up before the tick
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), str(w), str(h), str(nfeat), repr(t_go),
           repr(duration_s)]
    os.environ["ORBFE_CPU_WORKER_KIND"] = kind
    ps = [subprocess.Popen(cmd + [str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
    frames, late = 0, 0
    for p in ps:
        out, _ = p.communicate(timeout=duration_s + 180)
        try:
            tok = out.strip().split()
            frames += int(tok[-1])
            late += int(len(tok) > 1 and tok[0] == "late")
        except Exception:
            pass
    return {"value": round(frames / duration_s, 2), "unit": "frames/s", "cores": procs, "kind": kind, "late_starters": late,
            "per_process_frames_per_s": round(frames / duration_s / procs, 3), "host": host,
            "sample": f"{procs} processes (one per CPU this process may use: affinity mask / cgroup quota, see host) x {duration_s:.0f} s, each its own stream of {w}x{h} S(seed) "
                      f"frames, extract + BF match to previous frame, "
                      + ("oracle/_ref (unmodified reference ORBextractor.cc, cv stub) + oracle BF match" if kind == "reference"
                         else "oracle/orb_oracle.c"), "host_cpus": os.cpu_count()}


# ----------------------------------------------------------------------------------------------------------------
# the engines: HIP (the product) and the CPU stand-in used by the spawn-path test
# ----------------------------------------------------------------------------------------------------------------
class HipEngine:
    """One step = ONE call of the library's sequence pipeline (include/orbfe.h orbfe_pipeline_extract_match_device): the B
    resident frames of the step are a sequence; the library cuts it into sub-batches of F frames on P pipes (extractor +
    matcher + stream each), matches frame k against frame k - 1 across sub-batch boundaries, and -- ORBFE_PIPE_CONTINUE --
    frame 0 of a step against the last frame of the previous step.  Nothing of the pipeline lives in this file."""

    def __init__(self, args, local_rank, nfeatures, frames_per_launch, launches, world, device=None):
        from orb_slam2_ssd_semantic_amd import FramePipeline, _ffi
        self.ffi = _ffi
        self.L = _ffi.lib()
        self.w, self.h, self.F, self.nl = args.width, args.height, frames_per_launch, launches
        self.match = not args.no_match
        dev = local_rank if device is None else device
        self.P = max(1, args.pipes)
        self.pl = FramePipeline(nfeatures, 1.2, 8, 20, 7, max_width=self.w, max_height=self.h, sub_batch=self.F, npipes=self.P,
                                device=dev, blur_rounding=args.blur_rounding, nnratio=0.9, th=100, check_ori=True)
        self.exts, self.mats = self.pl.extractors, self.pl.matchers
        for opt in ("overlap", "rows_fast", "rows_blur"):
            if getattr(args, opt, None) is not None:
                for e in self.exts:
                    e.set_option(opt, getattr(args, opt))
        self.ext, self.mat = self.exts[0], self.mats[0]
        # One GPU: nothing consumes a step's outputs before the timed region's closing fence, so the steps run back to back
        # (ORBFE_PIPE_NO_JOIN: the pipes are not drained at step boundaries; the library orders the re-use of the output blocks
        # with events).  With N > 1 the gather consumes every step: the call joins the launch stream.
        self.free_run = world == 1 and not args.step_join
        self.cap = self.pl.capacity()
        B = self.F * self.nl
        self.B = B
        nsets = 2 if world > 1 else 1   # with N > 1 the gather of step k overlaps the kernels of step k+1
        self.outs = [(torch.zeros((B, self.cap, 7), dtype=torch.int32, device="cuda"),
                      torch.zeros((B, self.cap, 32), dtype=torch.uint8, device="cuda"),
                      torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(nsets)]
        self.d_match = torch.zeros((B, self.cap), dtype=torch.int32, device="cuda")
        self.d_nm = torch.zeros(B, dtype=torch.int32, device="cuda")
        # pairs (i, i - 1) of one sub-batch, for the matcher-alone timing of the stage table
        self.qf = torch.arange(1, self.F, dtype=torch.int32, device="cuda")
        self.tf = self.qf - 1

    def step(self, d_gray, out_set, stream, nframes=None, flags=None):
        """the whole resident batch d_gray [B, h, w] -> output set `out_set` (+ match rows)"""
        kps, desc, n = self.outs[out_set]
        B = self.B if nframes is None else nframes
        if flags is None:
            flags = self.pl.CONTINUE | (self.pl.NO_JOIN if self.free_run else 0)
        self.pl.extract_match_device(d_gray.data_ptr(), B, self.w, self.h, self.w, self.w * self.h, kps.data_ptr(), desc.data_ptr(),
                                     self.cap, n.data_ptr(), self.d_match.data_ptr() if self.match else None,
                                     self.d_nm.data_ptr() if self.match else None, flags=flags, stream=stream)

    def one_pipe_pass(self, d_gray, stream):
        """every sub-batch of the step through pipe 0's extractor ALONE on `stream` (nothing beside it on the chip): the
        exclusive stage times of the report"""
        kps, desc, n = self.outs[0]
        F, w, h = self.F, self.w, self.h
        for j in range(self.nl):
            lo = j * F
            self.ext.extract_batch_device(d_gray[lo].data_ptr(), F, w, h, w, w * h, kps[lo].data_ptr(), desc[lo].data_ptr(), self.cap,
                                          n[lo:].data_ptr(), stream)


class FakeEngine:
    """CPU stand-in with the same launch interface: fills the padded outputs with deterministic numbers.  No kernels,
    no oracle -- it only lets the distributed plumbing of this file run where there is no GPU."""

    def __init__(self, args, local_rank, nfeatures, frames_per_launch, launches, world):
        self.w, self.h, self.F, self.nl = args.width, args.height, frames_per_launch, launches
        self.cap = 64
        self.B = self.F * self.nl
        nsets = 2 if world > 1 else 1
        self.outs = [(torch.zeros((self.B, self.cap, 7), dtype=torch.int32), torch.zeros((self.B, self.cap, 32), dtype=torch.uint8),
                      torch.zeros(self.B, dtype=torch.int32)) for _ in range(nsets)]

    def step(self, d_gray, out_set, stream):
        kps, desc, n = self.outs[out_set]
        s = d_gray.reshape(self.B, -1)[:, :self.cap].to(torch.int32)
        n[:] = 1 + (s[:, 0] % (self.cap - 1))
        kps[:, :, 0] = s
        desc[:, :, 0] = s.to(torch.uint8)


class c_stdout_to_stderr:
    """RCCL prints a version banner on the C stdout when a communicator is created; this script's stdout carries ONE JSON
    line.  Inside the block, file descriptor 1 points at stderr (C stdio is flushed on both sides)."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per sub-batch of the library's pipeline")
    ap.add_argument("--launches", type=int, default=24, help="sub-batches per step: a step is ONE pipeline call over frames x "
                    "launches resident frames per GPU (default 24 576: twenty steps are > 1.5 s of GPU work)")
    ap.add_argument("--pipes", type=int, default=int(os.environ.get("ORBFE_BENCH_PIPES", "12")), help="pipes of the pipeline (round 6, 16 hardware queues: 8: 327 k, 12: 336-338 k, 16-24: 312-314 k frames/s; with 24-32 queues 16-20 pipes reach 335-337 k)")
    ap.add_argument("--step-join", action="store_true", help="join the launch stream after every step even on one GPU")
    ap.add_argument("--rows-fast", type=int, default=None, help="ORBFE_OPT_ROWS_FAST of the pipes' extractors (rows a FAST wave walks)")
    ap.add_argument("--rows-blur", type=int, default=None, help="ORBFE_OPT_ROWS_BLUR of the pipes' extractors")
    ap.add_argument("--overlap", type=int, default=None, help="ORBFE_OPT_OVERLAP of the pipes' extractors (blur on the handle's side "
                    "stream: 0 never, 1 / 2 beside FAST / the quadtree; default: the library's choice)")
    ap.add_argument("--blur-rounding", type=int, default=0, help="GaussianBlur column rounding: 0 canonical half-up, 1 = the x86 "
                    "SSE2 kernel's (what an x86-64 OpenCV <= 3.3 binary computes); value_blur_mode1 reports the other one")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--workload", choices=["S", "S_tum", "TUM"], default="S",
                    help="frames `value` is timed on: S(seed), S_tum(seed), or the TUM sequence at $TUM_FR3_WALKING_XYZ")
    ap.add_argument("--seeds", type=int, default=1024, help="distinct generator seeds in the resident batch (the rest of the "
                    "batch are roll / flip transforms of them)")
    ap.add_argument("--fast-mode", type=int, default=3, help="FAST variant of the timed region (3 auto = the library's default: dense or "
                    "lane-compacting per launch by the pass rate the kernel reports; 0 dense, 1 sparse shortcuts, 2 lane-compacting)")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baselines and the latency probe")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (rocprof runs): no PCIe leg, no "
                    "S_tum leg, no config 4, no CPU baselines")
    ap.add_argument("--path", choices=["extract_match", "bow", "config5"], default="extract_match",
                    help="bow: print the line of the device-resident ComputeBoW -> SearchByBoW chain instead; config5: the "
                         "line of BASELINE config 5 (512 resident 1920x1080 frames, 4000 features, one batched call)")
    ap.add_argument("--fake", action="store_true", help="CPU stand-in over gloo (spawn-path test only)")
    ap.add_argument("--same-device", action="store_true", help="TEST MODE: all ranks of an N > 1 run share GPU 0 (the exchange goes "
                    "through a host-staged gloo group, RCCL refuses two ranks per device).  Exercises every N > 1 code path with "
                    "real kernels on a one-GPU box; the line says same_device: true and is never a scaling figure")
    ap.add_argument("--config4-batch", type=int, default=1024, help="global batch of the config-4 leg (tests: an uneven 1023)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the exchange step out (compute only): with the default line's "
                    "ms_per_step this decomposes scaling efficiency into compute and exchange")
    ap.add_argument("--gather-full", action="store_true", help="N > 1: all `cap` slots of every frame travel (default: the valid prefix, "
                    "the maximum count of a probe step rounded up to 64: 1024 of 1088 slots at 1000 features)")
    ap.add_argument("--rccl-channels", type=int, default=None, help="N > 1: NCCL_MAX_NCHANNELS for RCCL (its copy kernels take compute "
                    "units from a VALU-bound step; fewer channels = fewer of them)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}: refusing to report a line for the "
              f"wrong number of GPUs", file=sys.stderr)
        sys.exit(2)
    fake = args.fake
    same = bool(args.same_device) and world > 1 and not fake
    if same:
        local_rank = 0
    args.same = same
    if not fake:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rccl_channels:
            os.environ["NCCL_MAX_NCHANNELS"] = str(int(args.rccl_channels))   # read by RCCL when the communicator is made
        from datetime import timedelta
        if fake:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif same:
            # every rank on device 0: CPU-side groups only; "gather" is used by the gather worker thread alone
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=900))
            CTL["gather"] = dist.new_group(backend="gloo", timeout=timedelta(seconds=900))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if not fake:
            # two CPU-side control groups for the guarded config-4 leg (config4_group_guarded): one for the main thread, one for
            # the worker thread that drives the library's own RCCL communicator
            CTL["main"] = dist.new_group(backend="gloo", timeout=timedelta(seconds=900))
            CTL["leg"] = dist.new_group(backend="gloo", timeout=timedelta(seconds=300))
    dev = "cpu" if fake else "cuda"

    def fence():
        if not fake:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not fake:
            torch.cuda.synchronize()

    if args.path == "bow":
        if world != 1 or fake:
            raise SystemExit("--path bow is a single-GPU line")
        print(json.dumps(bow_leg(args, local_rank, steps=args.steps, warmup=args.warmup, standalone=True)), flush=True)
        return
    if args.path == "config5":
        if world != 1 or fake:
            raise SystemExit("--path config5 is a single-GPU line")
        print(json.dumps(config5_leg(args, local_rank, warmup=args.warmup, reps=args.steps, standalone=True,
                                     check=not args.no_extras, pipeline=not args.no_extras)), flush=True)
        return
    w, h, F, NL, nf = args.width, args.height, args.frames, args.launches, args.nfeatures
    B = F * NL
    Engine = FakeEngine if fake else HipEngine
    eng = Engine(args, local_rank, nf, F, NL, world)
    if not fake:
        eng.pl.set_fast_mode(args.fast_mode)
        eng.fast_mode = args.fast_mode
    # the resident batch: `--seeds` DISTINCT generator seeds (default 1024 = one whole sub-batch of different images), expanded to
    # the B frames of a step by lossless roll / flip transforms of that set (every frame a different image)
    nbase = min(B, args.seeds)
    base = torch.from_numpy(base_frames(args.workload, nbase, w, h, 10000 + rank * 4096)).to(dev)
    d_gray = expand_frames(base, B)
    stream = None if fake else torch.cuda.current_stream().cuda_stream
    # (n, kps, desc) order of distributed.all_gather_keyframes
    gather, gcap = None, None
    if world > 1 and not args.no_gather:
        if not fake and not args.gather_full:
            # the valid prefix travels: counts of a probe step, maximum over all ranks, rounded up to 64 slots (the counts are
            # gathered whole every step, so a later frame with more keypoints than that is seen and reported: gather_truncated_frames)
            eng.step(d_gray, 0, stream)
            torch.cuda.synchronize()
            mx = torch.tensor([int(eng.outs[0][2].max().item())], dtype=torch.int64)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=CTL.get("main"))
            gcap = min(eng.cap, (int(mx.item()) + 63) // 64 * 64)
        gather = OverlappedKeyframeGather([(o[2], o[0], o[1]) for o in eng.outs], group=CTL.get("gather"), host_staged=same,
                                          gather_cap=gcap)
    counter = [0]
    use_gather = [True]

    def step():
        k = counter[0] % len(eng.outs)
        counter[0] += 1
        if gather:
            gather.acquire(k)  # set k is free once its previous gather (two steps ago) has read it (stream-level wait)
        eng.step(d_gray, k, stream)   # ONE call of the library's pipeline; with N > 1 it joins the launch stream
        if gather and use_gather[0]:
            # the one exchange step of the batched keyframe mode, asynchronous: RCCL runs on the gather stream after the
            # kernels above and overlaps the next step's kernels
            gather.launch(k)

    def timed(nsteps):
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        if not fake and world == 1:
            eng.pl.synchronize()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if (fake or same) else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    exch = None
    if gather:  # retire the work handles of the last gathers (already complete: fence() synchronised the device)
        for k in range(len(eng.outs)):
            gather.acquire(k)
        # the exchange in numbers: HIP events on the gather stream around the collectives of every timed step, the bytes one rank
        # contributes, the bus bandwidth an all-gather of that size reached ((N - 1) x bytes per rank leave and enter every GPU), and
        # the same K steps WITHOUT the exchange: what of the gather the kernels did not hide is the difference
        tm = gather.timing(last=args.steps)
        trunc = 0 if fake else sum(gather.truncated(k) for k in range(len(eng.outs)))
        use_gather[0] = False
        el_ng = timed(args.steps)
        use_gather[0] = True
        step()      # the last output set gathered again (the checks below read the gathered blocks of the last step)
        fence()
        for k in range(len(eng.outs)):
            gather.acquire(k)
        ms_g, ms_ng = elapsed / args.steps * 1e3, el_ng / args.steps * 1e3
        exch = {"gather_ms": round(tm["mean_ms"], 4) if tm else None, "gather_ms_max": round(tm["max_ms"], 4) if tm else None,
                "gather_timed_by": ("wall clock of the host-staged worker (D2H + gloo + H2D)" if same else
                                    "HIP events on the gather stream around the three collectives (RCCL)") if not fake else "not timed (CPU stand-in)",
                "gather_bytes_per_rank": int(gather.bytes_per_rank), "gather_slots_per_frame": int(gcap or eng.cap),
                "gather_slots_full": int(eng.cap), "gather_truncated_frames": int(trunc),
                "gather_bus_GBps": round(gather.bytes_per_rank * (world - 1) / (tm["mean_ms"] * 1e-3) / 1e9, 2) if tm and tm["mean_ms"] > 0 else None,
                "ms_per_step_no_gather": round(ms_ng, 4), "ms_per_step_with_gather": round(ms_g, 4),
                "value_no_gather": round(B * world * args.steps / el_ng, 2),
                "gather_hidden_frac": (round(max(0.0, min(1.0, 1.0 - max(0.0, ms_g - ms_ng) / tm["mean_ms"])), 4) if tm and tm["mean_ms"] > 0 else None),
                "rccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}
    total_frames = B * world * args.steps
    value = total_frames / elapsed

    result = None
    if rank == 0:
        result = {
            "metric": "ORB extract+match frames/sec on 640x480 TUM RGB-D; bit-exact kp/desc",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("BASELINE config 3 shape, HBM-RESIDENT inputs (the PCIe-inclusive rate of the same pipeline "
                                    "is `value_pcie_inclusive`): %dx%d synthetic frames %s(seed) (S / S_tum: no TUM data on the box), %d "
                                    "features, 8 levels, %s; a step = ONE call of the library's sequence pipeline "
                                    "(orbfe_pipeline_extract_match_device) over %d resident frames per GPU = %d sub-batches of %d on %d "
                                    "pipes, frame k matched against frame k - 1 across sub-batches and steps%s; exactness is "
                                    "checked against the in-repo oracle, which is pinned to the compiled reference "
                                    "(tests/test_ref_pin.py)") %
                                   (w, h, args.workload, nf, "extract only" if args.no_match else
                                    "extract + brute-force Hamming match to the previous frame (nnratio 0.9, TH_HIGH 100, rot. hist.)",
                                    B, NL, F, getattr(eng, "P", 1),
                                    ", asynchronous all-gather of counts/keypoints/descriptors per step" if world > 1 else ""),
                       "frames_per_gpu_per_step": B, "frames_per_launch": F, "width": w, "height": h, "nfeatures": nf,
                       "workload_name": args.workload, "pipes": getattr(eng, "P", 1), "blur_rounding": args.blur_rounding,
                       "fast_mode": {0: "dense", 1: "dense + wave-uniform shortcuts", 2: "lane-compacting",
                                     3: "auto (library default): lane-compacting or dense per launch from the pass rate the kernel reports; "
                                        "dense on these frames after the first probe (value_fast_dense: the same step pinned to dense)"}[args.fast_mode],
                       "value_is": "value_hbm_resident (bench contract: inputs resident in HBM when the timed region starts); SURVEY "
                                   "8(d) row 3 as worded -- host frames in, host results out -- is value_pcie_inclusive",
                       "match_kernel": None if args.no_match else "mfma_i8 (k_match_bf: exact int8 dot product on the matrix cores; "
                                       "the north star's xor/popcount formulation k_match_popc gives identical results: "
                                       "value_match_popc)",
                       "generator_seeds": int(min(B, args.seeds)),
                       "parallelism": f"frames sharded over {world} GPU(s), one process per GPU",
                       "streams": ("the library's pipeline: %d pipes (extractor + matcher + stream each), sub-batch j on pipe j mod %d; %s"
                                   % (getattr(eng, "P", 1), getattr(eng, "P", 1),
                                      "steps run back to back (ORBFE_PIPE_NO_JOIN), joined by the closing fence only (one GPU: no consumer "
                                      "between steps)" if getattr(eng, "free_run", False) else
                                      "every step joins the launch stream (the gather consumes it)"))},
        }
        result["value_hbm_resident"] = result["value"]
        if world > 1:
            result["exchange"] = exch if exch is not None else {"gather": "disabled (--no-gather): compute only"}
            if exch:
                for k in ("gather_ms", "gather_bytes_per_rank", "gather_bus_GBps", "gather_hidden_frac", "ms_per_step_no_gather", "value_no_gather"):
                    result[k] = exch[k]
        if same:
            result["same_device"] = True
            result["config"]["workload"] = ("SAME-DEVICE TEST MODE -- %d ranks share ONE GPU, host-staged gloo exchange: exercises the N > 1 "
                                            "code paths with real kernels, NOT a scaling measurement.  " % world) + result["config"]["workload"]
        if fake:
            result["fake"] = True
            result["config"]["workload"] = "FAKE CPU stand-in (spawn-path test), not a measurement"
            result["gathered_frames"] = int(gather.result(0)[0].shape[0]) if gather else B

    if not fake and rank == 0:
        if args.no_extras:
            result["exact_checked"] = False   # profiling runs: no oracle in the process
        else:
            result.update(self_check(eng, d_gray, (counter[0] - 1) % len(eng.outs), nf))
    if not fake and world > 1 and not args.no_extras and gather:
        chk = gathered_check(eng, gather, d_gray, (counter[0] - 1) % len(eng.outs), nf, rank, world)
        if rank == 0:
            result.update(chk)
    if not fake:
        extras(args, eng, d_gray, value, result, rank, local_rank, world, fence, step, timed)
    if gather:
        gather.close()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)   # anything a native library left in the C stdout buffer goes out BEFORE the line
        print(json.dumps(result), flush=True)
    if CTL.get("abandon"):
        # a rank's worker thread is stuck inside the group leg's communicator: the line above is complete and says so; tearing
        # the process groups down would wait for that thread, so every rank leaves directly
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if world > 1:
        with c_stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()
    os.dup2(2, 1)   # nothing may follow the JSON line on stdout (buffers flushed at exit by native libraries)


def gathered_check(eng, gather, d_gray, out_set, nf, rank, world):
    """N > 1: frames of ANOTHER rank's shard, as they arrived through the all-gather, against the oracle.  Every rank sends
    rank 0 two of its input frames (CPU-side group); rank 0 extracts them with the oracle and compares them with the rows of
    the gathered blocks that belong to that rank."""
    from orb_slam2_ssd_semantic_amd import KP_DTYPE
    B = eng.B
    n_all, k_all, d_all = gather.result(out_set)
    torch.cuda.synchronize()
    picks = [0, B // 2 + 7]
    mine = torch.stack([d_gray[f] for f in picks]).cpu()
    bucket = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bucket, dst=0, group=CTL.get("main"))
    if rank != 0:
        return {}
    from oracle import oracle_ffi as O
    oe = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    checked = []
    for r in range(world):
        if r == 0 and world > 1:
            continue
        for i, f in enumerate(picks):
            ok, od = oe(bucket[r][i].numpy())
            row = r * B + f
            nd = int(n_all[row].item())
            gk = k_all[row, :nd].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
            if not (nd == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(d_all[row, :nd].cpu().numpy(), od)):
                raise SystemExit(f"bench.py: frame {f} of rank {r}, read from the gathered blocks on rank 0, differs from the oracle")
            checked.append({"rank": r, "frame": f, "keypoints": nd})
    return {"gathered_exact_checked": True, "gathered_exact_checked_frames": checked}


def self_check(eng, d_gray, out_set, nf, seed=2026, blur_mode=None, nrandom=8):
    """The output of the TIMED region itself against the oracle (pinned to the compiled reference, tests/test_ref_pin.py):
    a dozen frames of the last timed step's output set -- count, every keypoint field bit pattern, every descriptor byte,
    order -- and their brute-force matches against the predecessor frame, including the pairs that straddle a sub-batch
    boundary and the step boundary.  Checker use of oracle/ only."""
    from oracle import oracle_ffi as O
    from orb_slam2_ssd_semantic_amd import KP_DTYPE
    kps, desc, n = eng.outs[out_set]
    F, B = eng.F, eng.B
    oe = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    bm = getattr(getattr(eng, "pl", None), "blur_rounding", 0) if blur_mode is None else blur_mode
    if bm:
        oe.set_blur_mode(bm)
    # frame 0 (its predecessor is the LAST frame of the previous step: ORBFE_PIPE_CONTINUE), the first frames of two sub-batches
    # (predecessor extracted on another pipe), the last frame, and eight random ones
    frames = sorted(set([0, B - 1] + [j * F for j in (1, 2) if j * F < B] +
                        [int(v) for v in np.random.default_rng(seed).integers(1, B, nrandom)])) if B > 1 else [0]
    cache = {}

    def oracle_frame(f):
        if f not in cache:
            cache[f] = oe(d_gray[f].cpu().numpy())
        return cache[f]
    checked = []
    for f in frames:
        ok, od = oracle_frame(f)
        nd = int(n[f].item())
        gk = kps[f, :nd].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
        gd = desc[f, :nd].cpu().numpy()
        same = nd == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(gd, od)
        if not same:
            raise SystemExit(f"bench.py self-check: frame {f} of the timed region differs from the oracle "
                             f"({nd} vs {len(ok)} keypoints)")
        row = {"frame": f, "keypoints": nd}
        if eng.match:
            pf = (f - 1) % B   # a real sequence; the steps repeat the same resident batch, so frame 0 follows frame B - 1
            pk, pd = oracle_frame(pf)
            rm, _, _, rn = O.match_bf(od, pd, ok["angle"], pk["angle"], 0.9, 100, True)
            gm = eng.d_match[f, :nd].cpu().numpy()
            if not (np.array_equal(gm, rm) and int(eng.d_nm[f].item()) == rn):
                raise SystemExit(f"bench.py self-check: matches of frame {f} (against frame {pf}) differ from the oracle")
            row["matches"] = int(rn)
        checked.append(row)
    return {"exact_checked": True, "exact_checked_frames": checked}


def stage_report(ext, stage, mean_kp, w, h, nf, F, workload, local_rank):
    """Roofline object of the dominant kernel + per-stage table from the library's HIP-event stage times (`stage`, ms per
    launch of F frames, measured live on the launch stream) and SURVEY 8(d)'s algorithmic bytes.  Counter-derived fields
    (`traffic`, `valu_ceiling`) are attached only from a committed profile of exactly this shape."""
    ncand = int(sum(len(ext.candidates(l, frame=0)) for l in range(8)))
    ab = algorithmic_bytes(w, h, mean_kp, ncand)
    ab_px = algorithmic_bytes(w, h, mean_kp, 0)
    stage_k = {k: stage[k] for k in ("pyramid", "fast", "octree", "blur", "describe")}
    dom = max(stage_k, key=stage_k.get)

    def gbs(name, tab=ab):
        return tab[name] * F / (stage_k[name] * 1e-3) / 1e9 if stage_k[name] > 0 else 0.0
    kname = {"pyramid": "k_pyr_walk (7 launches)", "fast": "k_fast_map_c" if stage.get("fast_form") == "lane-compacting" else "k_fast_map_u",   # the dense kernel of batch handles (cell-row runs)
             "octree": "k_octree", "blur": "k_blur7", "describe": "k_orient_describe"}
    # k_fast_map is bound by VALU issue (VALUBusy 0.92: profiles/*_pmc_valubusy.json), not by HBM: the label says so; achieved /
    # peak / frac stay the HBM figures the contract defines (algorithmic bytes / kernel time against 8 TB/s), valu_frac beside them
    roof = {"bound": "valu" if dom == "fast" else "hbm", "frac_is": "hbm: achieved / peak", "kernel": kname[dom],
            "achieved": round(gbs(dom), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs(dom) / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes_per_launch": int(ab[dom] * F), "launch_ms": round(stage_k[dom], 4),
            "frames_per_launch": F}
    if dom == "fast":  # SURVEY 8(d) calls the 8 B / candidate term negligible; on S it is not: both forms are reported
        roof["frac_pixels_only"] = round(gbs("fast", ab_px) / HBM_PEAK_GBS, 5)
        roof["algorithmic_bytes_per_launch_pixels_only"] = int(ab_px["fast"] * F)
    # The FAST pass is bound by VALU issue, not by HBM: print that ceiling next to the HBM one.  Measured, not
    # modelled: VALUBusy = share of the kernel's cycles in which the VALU was issuing, SQ_INSTS_VALU = wave
    # instructions executed (rocprofv3 --pmc passes of this command, committed under profiles/).
    clk = getattr(torch.cuda.get_device_properties(local_rank), "clock_rate", 2.4e6) * 1e3  # kHz -> Hz (2.4 GHz: profiles/r02_valu_mix.json)
    P = [a * b for a, b in level_sizes(w, h)]
    shape = {"width": w, "height": h, "nfeatures": nf, "workload": workload, "frames_per_launch": F}
    kn = roof["kernel"].split(" ")[0]
    name, pj = profile_for_shape("_pmc_valubusy.json", shape)
    if pj and profile_round(name) < KERNEL_ROUND:
        roof["valu_ceiling"] = {"stale": f"profiles/{name} predates the round-{KERNEL_ROUND} kernels: not attached"}
        pj = None
    if pj and kn in pj.get("VALUBusy_percent", {}):
        vbusy = pj["VALUBusy_percent"][kn] / 100.0
        vc = {"kernel": kn, "valu_busy_frac": round(vbusy, 4), "frac": round(vbusy, 4),
              "min_ms_at_this_instruction_count": round(stage_k[dom] * vbusy, 4), "measured_ms": round(stage_k[dom], 4),
              "source": f"profiles/{name} (rocprofv3 --pmc VALUBusy / SQ_INSTS_VALU of this command and shape; committed "
                        "file, not measured in this run)"}
        if kn in pj.get("SQ_INSTS_VALU_per_launch", {}):
            nv = pj["SQ_INSTS_VALU_per_launch"][kn]
            vc["wave_valu_insts_per_launch"] = int(nv)
            vc["lane_valu_insts_per_pixel"] = round(nv * 64 / (F * sum(P)), 2)
            vc["clk_per_wave_valu_inst_per_simd"] = round(stage_k[dom] * 1e-3 * clk * N_SIMD / nv, 3)
        roof["valu_ceiling"] = vc
        roof["valu_frac"] = vc["valu_busy_frac"]
    name, pj = profile_for_shape("_pmc_hbm.json", shape)
    stale = bool(pj) and profile_round(name) < KERNEL_ROUND
    if stale:
        roof["traffic_source"] = f"stale: profiles/{name} predates the round-{KERNEL_ROUND} kernels; traffic omitted"
    elif pj and kn in pj.get("FETCH_SIZE_KB", {}):
        # gfx950: FETCH_SIZE reports half of the read bytes (calibrated on this repo's access shapes,
        # profiles/r01_fetch_calibration.txt), WRITE_SIZE is exact
        tb = (2 * pj["FETCH_SIZE_KB"][kn]["mean_per_launch"] + pj.get("WRITE_SIZE_KB", {}).get(kn, {"mean_per_launch": 0})["mean_per_launch"]) * 1024
        roof["traffic"] = int(tb)
        roof["traffic_source"] = (f"profiles/{name}: 2 x FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 "
                                  "--pmc passes of this command and shape (committed file, not measured in this run)")
    elif not stale:
        roof["traffic_source"] = ("no committed counter file for this shape (width, height, nfeatures, workload, "
                                  "frames_per_launch): traffic omitted rather than rescaled")
    try:  # what a plain device copy reaches on this part (tools/hbm_rate.py), next to the 8 TB/s datasheet peak
        hr = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_rate.json")))
        roof["hbm_rate_measured"] = {"copy_GBps": round(hr["copy_read_plus_write_TBps"] * 1e3, 1),
                                     "read_GBps": round(hr["read_only_sum_TBps"] * 1e3, 1),
                                     "write_GBps": round(hr["write_only_fill_TBps"] * 1e3, 1),
                                     "source": "profiles/r02_hbm_rate.json (tools/hbm_rate.py, committed file)"}
    except Exception:
        pass
    pf_ms = stage_k["pyramid"] + stage_k["fast"]
    pf_gbs = (ab["pyramid"] + ab["fast"]) * F / (pf_ms * 1e-3) / 1e9
    pf_gbs_px = (ab_px["pyramid"] + ab_px["fast"]) * F / (pf_ms * 1e-3) / 1e9
    stages = {k: {"ms": round(v, 4), "GBps": round(gbs(k), 2), "frac": round(gbs(k) / HBM_PEAK_GBS, 5)}
              for k, v in stage_k.items()}
    stages["fast"]["frac_pixels_only"] = round(gbs("fast", ab_px) / HBM_PEAK_GBS, 5)
    stages["pyramid+fast"] = {"ms": round(pf_ms, 4), "GBps": round(pf_gbs, 2), "frac": round(pf_gbs / HBM_PEAK_GBS, 5),
                              "frac_pixels_only": round(pf_gbs_px / HBM_PEAK_GBS, 5)}
    stages["extract_total_ms"] = round(stage["total"], 4)
    stages["per"] = f"launch of {F} frames"
    return roof, stages, ncand


def exclusive_stage_pass(eng, d_gray, stream, local_rank):
    """ONE table of stage times, ONE mode: every sub-batch of a step through pipe 0's extractor alone, ORBFE_OPT_OVERLAP = 0
    (every kernel in line on one stream, nothing beside it on the chip), the library's HIP events on that stream around every
    stage, averaged over the step's sub-batches; then the matcher alone on the same stream.  ms per sub-batch of F frames."""
    ext = eng.ext
    ext.set_option("overlap", 0)
    form = getattr(eng, "fast_mode", 0)
    if form == 3:   # auto: the table is taken in the form the mode has settled on for these frames (no probe call inside it)
        ps = ext.fast_stats()
        form = 0 if ps["parked_pairs"] > 0.25 * 128.0 * max(ps["row_steps"], 1) else 2
        ext.set_fast_mode(form)
    eng.one_pipe_pass(d_gray, stream)       # warm-up in this mode
    torch.cuda.synchronize()
    ext.set_profiling(True)
    eng.one_pipe_pass(d_gray, stream)
    torch.cuda.synchronize()
    st = ext.stage_ms()
    ext.set_profiling(False)
    ext.set_option("overlap", -1)
    if getattr(eng, "fast_mode", 0) == 3:
        ext.set_fast_mode(3)
    st["fast_form"] = {0: "dense", 1: "dense + shortcuts", 2: "lane-compacting"}[form]
    match_ms = 0.0
    if eng.match:
        kps, desc, n = eng.outs[0]
        F = eng.F
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        m0.record()
        for _ in range(5):   # F - 1 pairs (i, i - 1) of one sub-batch
            eng.L.orbfe_match_bf_frames_device(eng.mat.handle, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), eng.cap,
                                               eng.qf.data_ptr(), eng.tf.data_ptr(), F - 1, 0.9, 100, 1, eng.d_match[1].data_ptr(),
                                               eng.d_nm[1:].data_ptr(), stream)
        m1.record()
        torch.cuda.synchronize()
        match_ms = m0.elapsed_time(m1) / 5
    st["match"] = match_ms
    return st


def extras(args, eng, d_gray, value, result, rank, local_rank, world, fence, step, timed):
    """Everything besides the timed region: stage table + roofline, PCIe-inclusive leg, S_tum leg, config 4, CPU legs."""
    from orb_slam2_ssd_semantic_amd import ORBextractor
    w, h, F, NL, nf = args.width, args.height, args.frames, args.launches, args.nfeatures
    B = F * NL
    stream = torch.cuda.current_stream().cuda_stream
    ext = eng.ext
    torch.cuda.synchronize()
    n_host = eng.outs[0][2].cpu().numpy()

    if rank == 0:
        stage = exclusive_stage_pass(eng, d_gray, stream, local_rank)
        # the taps (candidate lists) now belong to frame 0 of the LAST sub-batch pipe 0's extractor ran
        assert eng.pl.overflow() == 0, "device-side capacity overflow during the timed region"
        roof, stages, ncand = stage_report(ext, stage, float(n_host.mean()), w, h, nf, F, args.workload, local_rank)
        stages["match_ms"] = round(stage["match"], 4)
        if stage["match"] > 0:
            nn = n_host[:F].astype(np.float64)
            stages["match_Gdist_per_s"] = round(float((nn[1:] * nn[:-1]).sum()) / (stage["match"] * 1e-3) / 1e9, 2)
        stages["mode"] = ("exclusive: one stream, ORBFE_OPT_OVERLAP = 0, pipe 0's extractor alone on the chip, HIP events of the library "
                          f"around every stage, mean over the {NL} sub-batches of a step; the matcher alone on the same stream.  Sum = the "
                          "one-pipe in-line cost of a sub-batch; the timed region overlaps stages of different sub-batches, so "
                          "ms_per_step / launches is smaller than the sum")
        result["roofline"] = roof
        result["stages"] = stages
        # the driver's record keeps top-level scalars: the table once more, flat
        for k in ("pyramid", "fast", "octree", "blur", "describe"):
            result[f"stage_ms_exclusive_{k}"] = round(stage[k], 4)
        result["stage_ms_exclusive_match"] = round(stage["match"], 4)
        result["stage_table_fast_form"] = stage.get("fast_form")
        result["stage_ms_exclusive_sum"] = round(sum(stage[k] for k in ("pyramid", "fast", "octree", "blur", "describe", "match")), 4)
        result["stage_ms_per_sub_batch_in_timed_region"] = round(result["ms_per_step"] / NL, 4)
        vc = roof.get("valu_ceiling") or {}
        result["valu_frac_fast"] = vc.get("valu_busy_frac")
        result["roofline_frac_hbm"] = roof["frac"]
        result["config"]["mean_keypoints_per_frame"] = round(float(n_host.mean()), 1)
        result["config"]["fast_candidates_frame0"] = ncand

    if args.no_extras:
        return

    if world == 1:
        # ---- the same step in the other modes, each timed like `value` (fence, K steps back to back, fence) -------------------
        nsteps = max(2, args.steps // 4)

        def rate(run_step, nst=nsteps):
            run_step()
            torch.cuda.synchronize()
            eng.pl.synchronize()
            t1 = time.perf_counter()
            for _ in range(nst):
                run_step()
            eng.pl.synchronize()
            torch.cuda.synchronize()
            return B * nst / (time.perf_counter() - t1)

        if eng.match:
            # the north star's formulation of the all-pairs matcher (xor + v_bcnt popcount, k_match_popc): identical results
            eng.pl.set_bf_kernel(1)
            result["value_match_popc"] = round(rate(step), 2)
            eng.pl.set_bf_kernel(0)
            # BASELINE config 2: extract only
            eng.match = False
            result["value_extract_only"] = round(rate(step), 2)
            eng.match = True
        # joined steps (what a consumer between the steps costs)
        fr = eng.free_run
        eng.free_run = not fr
        result["value_step_joined" if fr else "value_free_run"] = round(rate(step), 2)
        eng.free_run = fr
        # the other blur rounding: blur_rounding = 1 is what an x86-64 OpenCV <= 3.3 BINARY computes (SSE2 column kernel), 0 the
        # canonical integer formula; the same pipeline, switched with ORBFE_OPT_BLUR_ROUNDING
        other_mode = 1 - args.blur_rounding
        for e in eng.exts:
            e.set_option("blur_rounding", other_mode)
        eng.pl.blur_rounding = other_mode
        eng.pl.reset_sequence()
        result["value_blur_mode%d" % other_mode] = round(rate(step), 2)
        if not args.no_cpu_baseline:
            result["blur_mode%d_exact_checked" % other_mode] = self_check(eng, d_gray, 0, nf, blur_mode=other_mode, nrandom=2)["exact_checked"]
        for e in eng.exts:
            e.set_option("blur_rounding", args.blur_rounding)
        eng.pl.blur_rounding = args.blur_rounding
        eng.pl.reset_sequence()
        # camera-like frames: the same step on S_tum(seed) (256 distinct seeds, expanded like the main batch)
        other = "S_tum" if args.workload == "S" else "S"
        d_other = expand_frames(torch.from_numpy(base_frames(other, min(B, 256), w, h, 10000)).cuda(), B)
        eng.pl.set_fast_mode(args.fast_mode)   # another workload: the auto mode starts over (its dense runs last up to 256 calls)
        result["value_%s" % other.lower()] = round(rate(lambda: eng.step(d_other, 0, stream)), 2)
        if not args.no_cpu_baseline:
            # the output of THIS leg (on S_tum: the lane-compacting FAST kernel's timed output) against the oracle
            chk = self_check(eng, d_other, 0, nf, seed=2027, nrandom=4)
            result["%s_exact_checked" % other.lower()] = chk["exact_checked"]
            result["%s_exact_checked_frames" % other.lower()] = chk["exact_checked_frames"]
        # ... and its own stage table and roofline object (the kernel that dominates THESE frames)
        n_other = eng.outs[0][2].cpu().numpy()
        st2 = exclusive_stage_pass(eng, d_other, stream, local_rank)
        roof2, stages2, ncand2 = stage_report(ext, st2, float(n_other.mean()), w, h, nf, F, other, local_rank)
        stages2["match_ms"] = round(st2["match"], 4)
        stages2["fast_form"] = st2.get("fast_form")
        result["roofline_%s" % other.lower()] = roof2
        result["stages_%s" % other.lower()] = stages2
        result["roofline_frac_hbm_%s" % other.lower()] = roof2["frac"]
        for k in ("pyramid", "fast", "octree", "blur", "describe"):
            result["stage_ms_exclusive_%s_%s" % (other.lower(), k)] = round(st2[k], 4)
        # north_star: ">= 60 % of HBM roofline on the pyramid/FAST pass" -- the pass as a whole (algorithmic bytes of the resize
        # chain and of the FAST pass over the sum of their exclusive times), on both workloads
        s_main = result["stages"]["pyramid+fast"]
        s_oth = stages2["pyramid+fast"]
        by = {args.workload: s_main, other: s_oth}
        result["roofline_pass"] = {"what": "pyramid + FAST pass: algorithmic bytes (SURVEY 8(d)) / (exclusive pyramid ms + exclusive FAST ms) "
                                           "against the 8 TB/s HBM peak; north_star asks for >= 0.60",
                                   "target": 0.60, "unit": "fraction of 8000 GB/s",
                                   **{k: {"frac": v["frac"], "frac_pixels_only": v["frac_pixels_only"], "ms": v["ms"], "GBps": v["GBps"]}
                                      for k, v in by.items()},
                                   "bound": "VALU issue, not HBM: the exact per-pixel FAST score costs more lane instructions than 0.60 leaves "
                                            "room for (<= 22 per pixel for resize + FAST together; DESIGN.md)"}
        result["roofline_pass_frac_S"] = by["S"]["frac"]
        result["roofline_pass_frac_S_tum"] = by["S_tum"]["frac"]
        # the same two workloads with the FAST form pinned to dense (what rounds 1-4 shipped): the library's default, auto, picks
        # the lane-compacting kernel where few pixel pairs pass the necessary test (S_tum: 18 %) and dense elsewhere (S: 84 %)
        eng.pl.set_fast_mode(0)
        result["value_%s_fast_dense" % other.lower()] = round(rate(lambda: eng.step(d_other, 0, stream)), 2)
        eng.pl.reset_sequence()
        result["value_fast_dense"] = round(rate(step), 2)
        eng.pl.set_fast_mode(args.fast_mode)
        result["workloads"] = workload_legs(args, eng, d_gray[:F], d_other[:F], local_rank)
        del d_other
        rp = real_photo_leg(args, eng, stream, rate, check=not args.no_cpu_baseline)
        if rp is not None:
            result["real_photo"] = rp
            result["value_real_photo"] = rp["frames_per_s"]
            result["real_photo_exact_checked"] = rp.get("exact_checked", False)
        eng.pl.reset_sequence()
        step()   # the resident batch's results back in the output set
        eng.pl.synchronize()

    # ---- config 4: 2000 features, a 1024-frame batch sharded over the ranks + all-gather ---------------------------
    c4 = config4_leg(args, rank, local_rank, world, fence, global_batch=args.config4_batch)
    if rank == 0:
        result["config4"] = c4
        # one-number summaries at the top level (the driver's record keeps top-level scalars)
        if isinstance(c4, dict) and "strong" in c4:
            result["config4_frames_per_s"] = c4["strong"]["frames_per_s"]           # the 1024-frame batch sharded over the ranks
            result["config4_weak_frames_per_s"] = c4["weak"]["frames_per_s"]
            result["config4_pipeline_frames_per_s"] = c4.get("pipeline_frames_per_s")
            result["config4_allgather_ms"] = c4["strong"]["allgather_ms"]
            result["config4_allgather_bus_GBps"] = c4["strong"]["allgather_bus_GBps"]
            cg = c4.get("cabi_group") or {}
            result["config4_cabi_group_frames_per_s"] = cg.get("frames_per_s")
            result["config4_cabi_group_status"] = "ok" if "frames_per_s" in cg else ("error: " + str(cg.get("error")))
            if "roofline" in c4:
                result["config4_roofline_frac_hbm"] = c4["roofline"]["frac"]

    if world == 1 and not args.no_cpu_baseline:
        result["projection_chain"] = projection_leg(local_rank)
    if world == 1:
        result["config5"] = config5_leg(args, local_rank, check=not args.no_cpu_baseline)
        result["config5_frames_per_s"] = result["config5"]["frames_per_s"]
        result["config5_pipeline_frames_per_s"] = result["config5"]["frames_per_s_pipeline"]
        result["config5_roofline_frac_hbm"] = result["config5"]["roofline"]["frac"]
        result["bow_chain"] = bow_leg(args, local_rank)
        result["bow_search_pairs_per_s"] = round(1e6 / result["bow_chain"]["search_by_bow_us_per_pair"], 1)
        result["stereo_chain"] = stereo_leg(args, local_rank)
        result["stereo_pairs_per_s"] = result["stereo_chain"]["stereo_pairs_per_s"]
        result["host_api"] = host_api_leg(args, local_rank, d_gray[:F])
        result["host_api_frames_per_s"] = result["host_api"]["frames_per_s"]

    # ---- PCIe-inclusive leg (SURVEY 8(d) config 3 as worded: host frames in, host results out) through the library's host entry
    # point.  The host path is bound by the PCIe link, not by the kernels, and every stream of a process shares 16 hardware queues
    # with the two copy streams: it runs on a pipeline of its own with 3 pipes (what a host application streaming frames would
    # create; 0.96 of the link against 0.88 through the 12-pipe pipeline of the resident loop), made after that one is closed.
    if world == 1:
        import copy
        eng.pl.synchronize()
        eng.pl.close()
        torch.cuda.empty_cache()
        a3 = copy.copy(args)
        a3.pipes = 3
        eng3 = HipEngine(a3, local_rank, nf, F, 2, world)
        eng3.pl.set_fast_mode(args.fast_mode)
        result["pcie_inclusive"] = pcie_leg(eng3, d_gray[:F], w, h, F)
        result["pcie_inclusive"]["pipes"] = 3
        result["value_pcie_inclusive"] = result["pcie_inclusive"]["frames_per_s"]
        result["pcie_inclusive_frac_of_link"] = result["pcie_inclusive"]["frac_of_link_bound"]
        eng3.pl.close()
        del eng3
        torch.cuda.empty_cache()

    if world == 1 and not args.no_cpu_baseline:
        # online (single-frame, host buffers in / out) latency of ORBextractor::operator(): replicas-only path
        e1 = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=local_rank)
        img = d_gray[0].cpu().numpy().copy()
        for _ in range(5):
            e1(img)
        lat = []
        for _ in range(50):
            t2 = time.perf_counter()
            e1(img)
            lat.append(time.perf_counter() - t2)
        result["single_frame_host_latency_ms"] = round(float(np.median(lat)) * 1e3, 4)
        del e1
        sl = shim_latency_leg(img, nf)
        if sl:
            result["shim_single_frame"] = sl
            result["shim_operator_ms_keep_pyramid"] = sl["keep_pyramid_true_ms"]
            result["shim_operator_ms_no_pyramid"] = sl["keep_pyramid_false_ms"]
            result["shim_operator_ms_identical_input"] = sl["identical_input_second_call_ms"]
        cb = cpu_baseline(w, h, nf)
        result["cpu_baseline"] = cb
        result["speedup_vs_cpu_1thread"] = round(value / cb["value"], 1)
        cv = cpu_baseline_vectorised(w, h, nf)
        if cv is not None:
            result["cpu_baseline_vectorised"] = cv
            result["cpu_baseline_vectorised_frames_per_s"] = cv["value"]
            result["speedup_vs_cpu_1thread_vectorised"] = round(value / cv["value"], 1)
        result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(w, h, nf)
        result["cpu_baseline_all_cores_frames_per_s"] = result["cpu_baseline_all_cores"]["value"]


def shim_latency_leg(img, nf, reps=40):
    """ORB_SLAM2::ORBextractor::operator() of the product's C++ shim (shim/ORBextractor.cc), called the way the reference's Frame
    calls it (Frame::ExtractORB, src/Frame.cc:337-343), median ms per 640x480 frame:
      keep_pyramid true   the reference-faithful default: mvImagePyramid is filled after every call (one kernel + a 1.2 MB device-to-host
                          copy + eight cv::Mat headers) -- only the stereo matcher reads it
      keep_pyramid false  what INTEGRATION.md section 2 recommends for the RGB-D / monocular TUM configurations
      identical input     ORBFE_OPT_REUSE_IDENTICAL_INPUT (on in a -DORBFE_SHIM_PERFECT build): the fork builds two Frames from one
                          mImGray per image (perfect/src/Tracking.cc:685, :716); the second operator() is answered from the first
    The shim is product source; the build that binds it to Python sits with the checkers (oracle/_ref/libshim_ext.so)."""
    try:
        from oracle import ref_ffi as RF
        if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libshim_ext.so")):
            return None
        out = {}
        for keep in (True, False):
            e = RF.ShimExtractor(nf, 1.2, 8, 20, 7)
            for _ in range(5):
                e.extract_via_frame(img, keep_pyramid=keep)
            lat = []
            for _ in range(reps):
                t = time.perf_counter()
                e.extract_via_frame(img, keep_pyramid=keep)
                lat.append(time.perf_counter() - t)
            out["keep_pyramid_%s_ms" % str(keep).lower()] = round(float(np.median(lat)) * 1e3, 4)
            del e
        e = RF.ShimExtractor(nf, 1.2, 8, 20, 7)
        e.set_reuse(True)
        other = np.ascontiguousarray(img[::-1])
        first, second = [], []
        for i in range(reps + 3):
            a = img if i % 2 == 0 else other      # a new image every pair of calls, as Tracking sees it
            t = time.perf_counter()
            e.extract_via_frame(a, keep_pyramid=False)
            t1 = time.perf_counter()
            e.extract_via_frame(a, keep_pyramid=False)
            t2 = time.perf_counter()
            if i >= 3:
                first.append(t1 - t)
                second.append(t2 - t1)
        out["identical_input_first_call_ms"] = round(float(np.median(first)) * 1e3, 4)
        out["identical_input_second_call_ms"] = round(float(np.median(second)) * 1e3, 4)
        out["identical_input_reused_calls"] = e.reused_calls()
        out["identical_input_pairs"] = reps + 3
        out["includes"] = "the binding's copies of keypoints / descriptors into numpy arrays (both settings alike)"
        return out
    except Exception as ex:   # the checker-side build is optional on a box
        return {"error": repr(ex)}


def real_photo_leg(args, eng, stream, rate, check=True):
    """The same step on REAL photographs: the 640 x 480 frame set of tests/golden/real (scikit-image / scipy images incl. the
    motorcycle stereo pair, colour ones through Tracking's gray conversion with both Camera.RGB settings, JPEG re-encodes: the
    nearest thing to TUM frames the boxes hold), tiled to the resident batch by the same lossless roll / flip transforms as the
    synthetic sets.  FAST mode as shipped (auto).  Checked against the oracle like `value`."""
    from orb_slam2_ssd_semantic_amd import photos
    w, h, nf = args.width, args.height, args.nfeatures
    if not photos.available() or (w, h) != (640, 480):
        return None
    fr = photos.vga_gray_frames()
    base = torch.from_numpy(np.stack([a for _, a in fr])).cuda()
    d_ph = expand_frames(base, eng.B)
    eng.pl.set_fast_mode(args.fast_mode)
    eng.pl.reset_sequence()
    fps = rate(lambda: eng.step(d_ph, 0, stream))
    out = {"frames_per_s": round(fps, 2), "distinct_photographs": len(fr), "frames": [t for t, _ in fr],
           "resident_frames": int(eng.B), "fast_mode": "auto (library default)",
           "mean_keypoints_per_frame": round(float(eng.outs[0][2].float().mean().item()), 1),
           "source": "tests/golden/real (made by tests/golden/make_real_images.py from the build container's scikit-image / scipy "
                     "images; golden vectors from the compiled reference in tests/golden/real/golden.npz)"}
    st = eng.ext.fast_stats(reset=False)
    if st.get("row_steps"):
        out["fast_pass_rate_last_probe"] = round(st["parked_pairs"] / max(128.0 * st["row_steps"], 1), 4)
    if check:
        chk = self_check(eng, d_ph, 0, nf, seed=2028, nrandom=6)
        out["exact_checked"] = chk["exact_checked"]
        out["exact_checked_frames"] = chk["exact_checked_frames"]
    del d_ph
    return out


def pcie_leg(eng, d_src, w, h, F, nbatches=24):
    """Host frames in, host results out -- BASELINE config 3 as SURVEY 8(d) words it -- through the PRODUCT's host entry point
    orbfe_pipeline_extract_match: page-locked frames -> H2D -> pipes -> D2H of counts, padded keypoints / descriptors / matches, the
    three overlapped inside liborbfe.so (three device buffer sets, two copy streams, chunks of one sub-batch taking turns on the
    pipes).  One blocking call over nbatches x F frames; nothing of the pipeline lives in this file."""
    import ctypes as C
    pin_in = d_src.cpu().pin_memory()
    cap = eng.cap
    N = nbatches * F
    hk = torch.empty((N, cap, 7), dtype=torch.int32, pin_memory=True)
    hd = torch.empty((N, cap, 32), dtype=torch.uint8, pin_memory=True)
    hn = torch.empty(N, dtype=torch.int32, pin_memory=True)
    hm = torch.empty((N, cap), dtype=torch.int32, pin_memory=True)
    hnm = torch.empty(N, dtype=torch.int32, pin_memory=True)
    ptrs = (C.c_void_p * N)(*[pin_in[i % F].data_ptr() for i in range(N)])
    eng.pl.synchronize()
    eng.pl.reset_sequence()

    def link_rate(fn, nbytes, reps=12):
        for _ in range(3):   # first touches of the pinned pages / warm-up of the copy engines
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return nbytes * reps / (time.perf_counter() - t) / 1e9

    d_tmp = torch.empty_like(d_src)
    h2d = link_rate(lambda: d_tmp.copy_(pin_in, non_blocking=True), pin_in.numel())
    kps, desc, n = eng.outs[0]
    d_out = (kps[:F], desc[:F], n[:F], eng.d_match[:F], eng.d_nm[:F])
    h_out = (hk[:F], hd[:F], hn[:F], hm[:F], hnm[:F])
    out_bytes = sum(t.numel() * t.element_size() for t in d_out)
    d2h = link_rate(lambda: [hh.copy_(dd, non_blocking=True) for hh, dd in zip(h_out, d_out)], out_bytes)

    def run(nfr):
        eng.pl.extract_match(ptrs, nfr, w, h, w, hk.data_ptr(), hd.data_ptr(), cap, hn.data_ptr(),
                             hm.data_ptr() if eng.match else None, hnm.data_ptr() if eng.match else None)

    run(N)   # warm-up over the whole result arrays (first DMA into every page-locked page)
    t = time.perf_counter()
    run(N)
    dt = time.perf_counter() - t
    fps = N / dt
    link_fps = h2d * 1e9 / (w * h)
    res = {"frames_per_s": round(fps, 2), "h2d_GBps_measured": round(h2d, 2), "d2h_GBps_measured": round(d2h, 2),
           "bytes_in_per_frame": w * h, "bytes_out_per_frame": int(out_bytes // F),
           "link_bound_frames_per_s": round(link_fps, 1), "frac_of_link_bound": round(fps / link_fps, 4),
           "sample": f"ONE orbfe_pipeline_extract_match call over {nbatches} x {F} page-locked host frames: H2D -> extract + match -> D2H of "
                     f"counts, padded keypoints / descriptors / matches, overlapped inside the library"}
    # the host results of the call against the device-resident pipeline's on the same frames (frames F .. 2F - 1 repeat 0 .. F - 1)
    torch.cuda.synchronize()
    res["host_equals_resident_counts"] = bool(torch.equal(hn[:F], hn[F:2 * F]))
    return res


def workload_legs(args, eng, d_S, d_other, local_rank):
    """Stage times (overlap as shipped, one pipe) and rate of one sub-batch of F frames on both workloads, dense and sparse FAST."""
    w, h, F = args.width, args.height, args.frames
    stream = torch.cuda.current_stream().cuda_stream
    other = "S_tum" if args.workload == "S" else "S"
    sets = {args.workload: d_S, other: d_other}
    out = {}
    kps, desc, n = eng.outs[0]
    for name in sets:
        g = sets[name]
        row = {}
        for mode, label in ((0, "dense"), (1, "sparse"), (2, "compact")):
            eng.ext.set_fast_mode(mode, collect_stats=(mode >= 1))
            if mode >= 1:
                eng.ext.fast_stats(reset=True)
            for _ in range(2):
                eng.ext.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), eng.cap,
                                             n.data_ptr(), stream)
            torch.cuda.synchronize()
            if mode == 1:
                st = eng.ext.fast_stats(reset=True)
                row["sparse_arc_skip_frac"] = round(st["arc_skips"] / max(st["row_steps"], 1), 4)
                row["sparse_nms_skip_frac"] = round(st["nms_skips"] / max(st["row_steps"], 1), 4)
                eng.ext.set_fast_mode(1, collect_stats=False)
            if mode == 2:   # {row steps, batches, parked pairs} of the sampled waves
                st = eng.ext.fast_stats(reset=True)
                row["compact_pass_rate"] = round(st["parked_pairs"] / max(128.0 * st["row_steps"], 1), 4)
                row["compact_batch_fill"] = round(st["parked_pairs"] / max(64.0 * st["batches"], 1), 4)
                eng.ext.set_fast_mode(2, collect_stats=False)
            eng.ext.set_profiling(True)
            t = time.perf_counter()
            reps = 8
            for _ in range(reps):
                eng.ext.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), eng.cap,
                                             n.data_ptr(), stream)
                if eng.match:
                    eng.L.orbfe_match_bf_frames_device(eng.mat.handle, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), eng.cap,
                                                       eng.qf.data_ptr(), eng.tf.data_ptr(), F - 1, 0.9, 100, 1,
                                                       eng.d_match[1].data_ptr(), eng.d_nm[1:].data_ptr(), stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            sm = eng.ext.stage_ms()
            eng.ext.set_profiling(False)
            row[f"fps_{label}"] = round(reps * F / dt, 1)
            row[f"stage_ms_{label}"] = {k: round(sm[k], 4) for k in ("pyramid", "fast", "octree", "blur", "describe")}
        row["fast_candidates_frame0"] = int(sum(len(eng.ext.candidates(l, frame=0)) for l in range(8)))
        row["mean_keypoints_per_frame"] = round(float(n[:F].float().mean().item()), 1)
        row["note"] = "one pipe, blur beside the quadtree as shipped (ORBFE_OPT_OVERLAP built-in): stage intervals here overlap"
        out[name] = row
    eng.ext.set_fast_mode(args.fast_mode)
    return out


def host_api_leg(args, local_rank, d_src, nframes=8192, chunk=512):
    """The library's own host entry point (orbfe_extract_batch: host frames in, host keypoints / descriptors out, extract
    only) on page-locked buffers: chunks of `chunk` frames pipelined inside liborbfe.so on three streams."""
    import ctypes as C
    from orb_slam2_ssd_semantic_amd import ORBextractor, _ffi
    w, h, nf = args.width, args.height, args.nfeatures
    nb = d_src.shape[0]
    ext = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=chunk, device=local_rank)
    cap = ext.capacity()
    pf = d_src.cpu().pin_memory()
    pk = torch.empty((nframes, cap, 7), dtype=torch.int32).pin_memory()
    pd = torch.empty((nframes, cap, 32), dtype=torch.uint8).pin_memory()
    pn = torch.zeros(nframes, dtype=torch.int32).pin_memory()
    arr = (C.c_void_p * nframes)(*[pf[i % nb].data_ptr() for i in range(nframes)])
    L = _ffi.lib()

    def run():
        _ffi.check(L.orbfe_extract_batch(ext.handle, arr, nframes, w, h, w, pk.data_ptr(), pd.data_ptr(), cap, pn.data_ptr()),
                   "orbfe_extract_batch")
    run()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t) / reps
    return {"frames_per_s": round(nframes / dt, 1), "frames_per_call": nframes, "chunk": chunk, "mean_keypoints": round(float(pn.float().mean()), 1),
            "what": "orbfe_extract_batch on page-locked host frames / outputs (extract only; the pipeline is inside liborbfe.so)"}


def stereo_leg(args, local_rank, npairs=256, steps=10, warmup=3):
    """The stereo Frame constructor as a device-resident chain (src/Frame.cc:58-116): left and right images of `npairs`
    stereo pairs through two batched extractor calls, then Frame::ComputeStereoMatches for every pair on their output
    blocks (orbfe_stereo_matches_batch_device), one stream, nothing leaves HBM.  Right image = left image shifted by a
    disparity that grows towards the bottom, plus noise.  CPU side: the oracle's stereo matcher on one pair."""
    from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher
    w, h, nf = args.width, args.height, args.nfeatures
    stream = torch.cuda.current_stream().cuda_stream
    gl = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=npairs, device=local_rank)
    gr = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=npairs, device=local_rank)
    mat = ORBmatcher(0.9, True, device=local_rank)
    cap = gl.capacity()
    left = expand_frames(torch.from_numpy(base_frames("S", 32, w, h, 50000)).cuda(), npairs)
    disp = 4 + (20 * torch.arange(h, device="cuda")) // h                       # per row
    idx = (torch.arange(w, device="cuda")[None, :] + disp[:, None]) % w        # right[y, x] = left[y, x + d(y)]
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    right = torch.gather(left, 2, idx[None].expand(npairs, h, w))
    right = (right.to(torch.int16) + torch.randint(-3, 4, right.shape, device="cuda", generator=g, dtype=torch.int16)).clamp(0, 255).to(torch.uint8)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    oL = (z((npairs, cap, 7), torch.int32), z((npairs, cap, 32), torch.uint8), z(npairs, torch.int32))
    oR = (z((npairs, cap, 7), torch.int32), z((npairs, cap, 32), torch.uint8), z(npairs, torch.int32))
    du, dz = z((npairs, cap), torch.float32), z((npairs, cap), torch.float32)
    mbf, mb = 40.0, 0.08

    def extract():
        gl.extract_batch_device(left.data_ptr(), npairs, w, h, w, w * h, oL[0].data_ptr(), oL[1].data_ptr(), cap, oL[2].data_ptr(), stream)
        gr.extract_batch_device(right.data_ptr(), npairs, w, h, w, w * h, oR[0].data_ptr(), oR[1].data_ptr(), cap, oR[2].data_ptr(), stream)

    def stereo():
        mat.ComputeStereoMatches_batch_device(gl, gr, oL[0].data_ptr(), oL[1].data_ptr(), oL[2].data_ptr(), oR[0].data_ptr(), oR[1].data_ptr(),
                                              oR[2].data_ptr(), cap, npairs, mbf, mb, du.data_ptr(), dz.data_ptr(), stream)

    def timed(fn):
        for _ in range(warmup):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    t_ex = timed(extract)
    t_st = timed(stereo)
    nL = oL[2].cpu().numpy()
    u = du.cpu().numpy()
    matched = float(np.mean([(u[i, :nL[i]] >= 0).sum() for i in range(npairs)]))
    out = {"pairs": npairs, "features_per_image": round(float(nL.mean()), 1), "mean_stereo_matches_per_pair": round(matched, 1),
           "extract_left_right_ms_per_batch": round(t_ex, 4), "stereo_matches_ms_per_batch": round(t_st, 4),
           "stereo_matches_us_per_pair": round(t_st * 1e3 / npairs, 3),
           "stereo_pairs_per_s": round(npairs / ((t_ex + t_st) * 1e-3), 1)}
    if not args.no_cpu_baseline:
        from oracle import oracle_ffi as O
        from orb_slam2_ssd_semantic_amd import KP_DTYPE
        l0, r0 = left[0].cpu().numpy(), right[0].cpu().numpy()
        exL, exR = O.OracleExtractor(nf, 1.2, 8, 20, 7), O.OracleExtractor(nf, 1.2, 8, 20, 7)
        kL, dL = exL(l0)
        kR, dR = exR(r0)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ru, rd, _ = O.stereo_matches(exL, exR, kL, dL, kR, dR, mbf, mb)
        out["cpu_oracle_stereo_matches_us_per_pair"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        out["pair0_equals_oracle"] = bool(np.array_equal(u[0, :nL[0]].view(np.uint32), ru.view(np.uint32)) and nL[0] == len(kL)
                                          and np.array_equal(oL[0][0, :nL[0]].cpu().numpy().copy().view(KP_DTYPE).reshape(-1).view(np.uint8), kL.view(np.uint8)))
    return out


def bow_leg(args, local_rank, npairs=256, steps=10, warmup=3, standalone=False):
    """The device-resident chain behind Tracking / LoopClosing's BoW matching: extractor output block -> ComputeBoW for
    every frame -> SearchByBoW(KeyFrame, Frame) for a batch of pairs, one stream, nothing leaves HBM in between.
    256 pairs of 1000 x 1000 features (each frame against a shifted, re-noised view of itself), ORBvoc-shaped tree
    (k = 10, L = 6, levelsup = 4 -> 100 FeatureVector nodes).  CPU side: the oracle's SearchByBoW / transform on one pair."""
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor, ORBmatcher, ORBVocabulary
    w, h, nf = args.width, args.height, args.nfeatures
    B = 2 * npairs
    stream = torch.cuda.current_stream().cuda_stream
    ext = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=local_rank)
    mat = ORBmatcher(0.7, True, device=local_rank)
    voc = regular_vocabulary(10, 6, seed=3)
    V = ORBVocabulary(mat, **voc)
    cap = ext.capacity()
    a = expand_frames(torch.from_numpy(base_frames("S", 32, w, h, 30000)).cuda(), npairs)
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    b = torch.roll(a, shifts=(2, 3), dims=(1, 2)).to(torch.float32) + 2.5 * torch.randn(a.shape, device="cuda", generator=g)
    frames = torch.cat([a, b.round().clamp(0, 255).to(torch.uint8)])
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    d_kps, d_desc, d_n = z((B, cap, 7), torch.int32), z((B, cap, 32), torch.uint8), z(B, torch.int32)
    bl = dict(f_word=z((B, cap), torch.int32), f_node=z((B, cap), torch.int32), f_weight=z((B, cap), torch.float64),
              bow_id=z((B, cap), torch.int32), bow_val=z((B, cap), torch.float64), fv_node=z((B, cap), torch.int32),
              fv_off=z((B, cap + 1), torch.int32), fv_idx=z((B, cap), torch.int32), counts=z((B, 4), torch.int32))
    d_kf = torch.arange(0, npairs, dtype=torch.int32, device="cuda")
    d_f = d_kf + npairs
    d_match, d_nm = z((npairs, cap), torch.int32), z(npairs, torch.int32)
    ext.extract_batch_device(frames.data_ptr(), B, w, h, w, w * h, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), stream)

    def transform():
        V.transform_batch_device(d_desc.data_ptr(), d_n.data_ptr(), B, cap, 4, bl["f_word"].data_ptr(), bl["f_node"].data_ptr(),
                                 bl["f_weight"].data_ptr(), bl["bow_id"].data_ptr(), bl["bow_val"].data_ptr(),
                                 bl["fv_node"].data_ptr(), bl["fv_off"].data_ptr(), bl["fv_idx"].data_ptr(), bl["counts"].data_ptr(), stream)

    def search():
        mat.SearchByBoW_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), cap, None, bl["fv_node"].data_ptr(), bl["fv_off"].data_ptr(),
                                     bl["fv_idx"].data_ptr(), bl["counts"].data_ptr(), d_kf.data_ptr(), d_f.data_ptr(), npairs,
                                     d_match.data_ptr(), d_nm.data_ptr(), kf_kf=False, stream=stream)

    def timed(fn):
        for _ in range(warmup):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    t_tr = timed(transform)
    t_se = timed(search)
    n = d_n.cpu().numpy()
    nm = d_nm.cpu().numpy()
    counts = bl["counts"].cpu().numpy()
    out = {"pairs": npairs, "features_per_frame": round(float(n.mean()), 1), "vocabulary": "k=10 L=6 (1 111 111 nodes), levelsup 4",
           "fv_nodes_per_frame": round(float(counts[:, 1].mean()), 1), "mean_matches_per_pair": round(float(nm.mean()), 1),
           "bow_transform_ms_per_batch": round(t_tr, 4), "bow_transform_us_per_frame": round(t_tr * 1e3 / B, 3),
           "search_by_bow_ms_per_batch": round(t_se, 4), "search_by_bow_us_per_pair": round(t_se * 1e3 / npairs, 3)}
    if not args.no_cpu_baseline:
        from oracle import oracle_ffi as O
        desc = d_desc.cpu().numpy()
        kps = d_kps.cpu().numpy().view(KP_DTYPE).reshape(B, cap)
        fvs, t0 = [], time.perf_counter()
        for fidx in (0, npairs):
            r = O.bow_transform(voc, desc[fidx, :n[fidx]], 4)
            fvs.append((r["fv_node"], r["fv_off"], r["fv_idx"]))
        t_cpu_tr = (time.perf_counter() - t0) / 2
        t_cpu_se = 1e9
        for _ in range(20):   # best of 20: the host is busy with the other legs' worker processes now and then
            t0 = time.perf_counter()
            om, on = O.search_by_bow(desc[0, :n[0]], None, kps["angle"][0, :n[0]], fvs[0], desc[npairs, :n[npairs]], None,
                                     kps["angle"][npairs, :n[npairs]], fvs[1], 0.7, 50, False, True)
            t_cpu_se = min(t_cpu_se, time.perf_counter() - t0)
        assert np.array_equal(d_match[0, :n[npairs]].cpu().numpy(), om) and int(nm[0]) == on, "bow chain differs from the oracle"
        out["cpu_oracle_search_by_bow_us_per_pair"] = round(t_cpu_se * 1e6, 1)
        out["cpu_oracle_bow_transform_us_per_frame"] = round(t_cpu_tr * 1e6, 1)
        out["search_speedup_vs_cpu_1thread"] = round(t_cpu_se * 1e6 / (t_se * 1e3 / npairs), 1)
    if standalone:
        return {"metric": "SearchByBoW (KeyFrame, Frame) pairs/sec, device-resident after ComputeBoW", "unit": "pairs/s",
                "value": round(npairs / (t_se * 1e-3), 1), "n_gpus": 1, "steps": steps, "warmup": warmup,
                "ms_per_step": round(t_se, 4), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "256 (KeyFrame, Frame) pairs of 1000 x 1000 ORB features, ~100 vocabulary nodes each"},
                "bow_chain": out}
    return out


def projection_leg(local_rank, reps=40):
    """The per-frame matchers of Tracking under the reference's own signatures: SearchByProjection(Frame&,
    const Frame&, th, bMono) (TrackWithMotionModel, src/ORBmatcher.cc:1578-1724) on 1000 x 1000 features and
    SearchByProjection(Frame&, vector<MapPoint*>&, th) (SearchLocalPoints, :63-157) on 1000 features x 3000 map points.
    GPU side: the product's shim member called through the reference's class (oracle/_ref/libshim_ref.so: host gating on
    cv::Mat, ONE orbfe_search_by_projection call, replay) and that C-ABI call alone; cpu_baseline: the reference's compiled
    body (oracle/_ref/libref_orb.so) on the same mock Frames, one thread.  Times are the member call alone."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import proj_cases as PC
    from oracle import oracle_ffi as O
    from oracle import ref_ffi as R
    from orb_slam2_ssd_semantic_amd import ORBmatcher
    if not (R.available() and R.shim_available()):
        return {"skipped": "oracle/_ref libraries not present"}
    mat = ORBmatcher(0.9, True, device=local_rank)
    rng = np.random.default_rng(2026)
    out = {}

    def med(fn, n):
        v = []
        for _ in range(n):
            v.append(fn())
        return float(np.median(v))
    # ---- last frame ----
    cur, last = PC.last_frame_case(rng, 1000, 1000, "small")
    ra, rn = R.search_by_projection_last_frame(cur, last, 15.0, False)
    sa, sn = R.search_by_projection_last_frame(cur, last, 15.0, False, shim=True)
    assert sn == rn and np.array_equal(sa, ra), "projection_chain: shim differs from the reference body"
    q, valid = O.proj_queries_last_frame(cur["Tcw"], last["Tcw"], cur["K"], cur["bounds"], cur["scale_factors"], last["has_mp"],
                                         last["outlier"], last["world_pos"], last["octave"], last["obs_gt0"], 15.0, False)
    sel = valid.astype(bool)
    ci = PC.core_inputs(cur)
    qq, qd = q[sel], last["mpdesc"][sel]   # selected once: the fancy indexing is not part of the call

    def core():
        t = time.perf_counter()
        mat.SearchByProjectionCore(queries=qq, qdesc=qd, th=100, nnratio=0.0, ratio_rule=0, **ci)
        return (time.perf_counter() - t) * 1e3
    core()
    cpu = med(lambda: (R.search_by_projection_last_frame(cur, last, 15.0, False), R.last_call_ms())[1], 15)
    gpu = med(lambda: (R.search_by_projection_last_frame(cur, last, 15.0, False, shim=True), R.last_call_ms(shim=True))[1], reps)
    out["last_frame"] = {"frame_features": 1000, "last_frame_features": 1000, "queries": int(sel.sum()), "matches": int(rn), "th": 15,
                         "shim_member_ms": round(gpu, 4), "cabi_call_ms": round(med(core, reps), 4),
                         "cpu_reference_member_ms": round(cpu, 4), "speedup_vs_cpu_1thread": round(cpu / gpu, 2)}
    # ---- local map ----
    cur, mps = PC.local_map_case(rng, 1000, 3000)
    ra, rn = R.search_by_projection_local_map(cur, mps, 3.0, 0.8)
    sa, sn = R.search_by_projection_local_map(cur, mps, 3.0, 0.8, shim=True)
    assert sn == rn and np.array_equal(sa, ra), "projection_chain: shim differs from the reference body (local map)"
    cpu = med(lambda: (R.search_by_projection_local_map(cur, mps, 3.0, 0.8), R.last_call_ms())[1], 15)
    gpu = med(lambda: (R.search_by_projection_local_map(cur, mps, 3.0, 0.8, shim=True), R.last_call_ms(shim=True))[1], reps)
    out["local_map"] = {"frame_features": 1000, "map_points": 3000, "matches": int(rn), "th": 3,
                        "shim_member_ms": round(gpu, 4), "cpu_reference_member_ms": round(cpu, 4),
                        "speedup_vs_cpu_1thread": round(cpu / gpu, 2)}
    # ---- the other members of the class (mapping / loop-closing / initialisation threads), same two sides -------------------
    def pair(call, same):
        r, s_ = call(False), call(True)
        assert same(r, s_), "projection_chain: shim differs from the reference body"
        cpu = med(lambda: (call(False), R.last_call_ms())[1], 9)
        gpu = med(lambda: (call(True), R.last_call_ms(shim=True))[1], 25)
        return {"shim_member_ms": round(gpu, 4), "cpu_reference_member_ms": round(cpu, 4), "speedup_vs_cpu_1thread": round(cpu / gpu, 2)}
    eq = lambda a, b: all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))
    members = {}
    cur3, kfp = PC.frame_kf_case(rng, 1000, 1000)
    members["SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) 1000 x 1000"] = pair(
        lambda sh: R.search_by_projection_frame_kf(cur3, kfp, 10.0, 100, True, shim=sh), eq)
    kf, Scw, pts, mi = PC.kf_sim3_case(rng, 1000, 2000)
    members["SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) 1000 x 2000"] = pair(
        lambda sh: R.search_by_projection_kf_sim3(kf, Scw, pts, mi, 10, shim=sh), eq)
    m2 = dict(pts, null=np.zeros(len(pts["bad"]), np.uint8))
    members["Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) 1000 x 2000"] = pair(lambda sh: R.fuse_sim3(kf, Scw, m2, 4.0, shim=sh), eq)
    kf2, mps2 = PC.fuse_case(rng, 1000, 2000)
    members["Fuse(KeyFrame*, vpMapPoints, th) 1000 x 2000"] = pair(lambda sh: R.fuse(kf2, mps2, 3.0, shim=sh), eq)
    k1, k2, F12 = PC.triangulation_case(rng, 1000, 1000, 100)
    members["SearchForTriangulation 1000 x 1000, 100 nodes"] = pair(lambda sh: R.search_for_triangulation(k1, k2, F12, False, shim=sh), eq)
    s1, s2, s12, R12, t12, m_in = PC.sim3_pair_case(rng, 1000, 1000)
    members["SearchBySim3 1000 x 1000"] = pair(lambda sh: R.search_by_sim3(s1, s2, s12, R12, t12, 7.5, m_in, shim=sh), eq)
    f1, f2, prev = PC.initialization_case(rng, 2000, 2000)
    members["SearchForInitialization 2000 x 2000, window 100"] = pair(lambda sh: R.search_for_initialization(f1, f2, prev, 100, shim=sh), eq)
    out["other_members"] = members
    out["cpu_baseline"] = {"kind": "reference", "cores": 1, "unit": "ms per call",
                           "sample": "median of 15 (other_members: 9) calls of the reference's compiled bodies on the same mock objects"}
    out["exact_checked"] = True
    return out


def config5_leg(args, local_rank, nframes=512, nfeat=4000, w=1920, h=1080, warmup=3, reps=10, check=True, standalone=False, pipeline=True):
    """BASELINE config 5 as SURVEY 8(d) row 5 words it: 512 distinct device-resident 1920x1080 frames S(seed = 20000 + i)
    (1.06 GB: more than the 256 MB Infinity Cache), 4000 features, 8 levels, processed in ONE batched call; 3 warm-up + 10
    timed repetitions.  Stage times are the library's HIP events on the launch stream; the roofline object is the dominant
    kernel's, with counter fields only from a committed profile of this very shape."""
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
    stream = torch.cuda.current_stream().cuda_stream
    ext = ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nframes, device=local_rank)
    cap = ext.capacity()
    base = torch.from_numpy(base_frames("S", min(nframes, 64), w, h, 20000)).cuda()
    frames = expand_frames(base, nframes)
    kps = torch.zeros((nframes, cap, 7), dtype=torch.int32, device="cuda")
    desc = torch.zeros((nframes, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(nframes, dtype=torch.int32, device="cuda")

    def one():
        ext.extract_batch_device(frames.data_ptr(), nframes, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), stream)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    ext.set_profiling(True)
    t = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    stage = ext.stage_ms()
    ext.set_profiling(False)
    assert ext.overflow() == 0, "device-side capacity overflow in config 5"
    n_host = n.cpu().numpy()
    roof, stages, ncand = stage_report(ext, stage, float(n_host.mean()), w, h, nfeat, nframes, "S", local_rank)
    out = {"frames_per_s": round(nframes * reps / dt, 1), "ms_per_call": round(dt / reps * 1e3, 4), "frames_per_call": nframes,
           "width": w, "height": h, "nfeatures": nfeat, "cap": cap, "mean_keypoints_per_frame": round(float(n_host.mean()), 1),
           "fast_candidates_frame0": ncand, "resident_input_bytes": int(frames.numel()), "warmup": warmup, "reps": reps,
           "roofline": roof, "stages": stages}
    # the same 512 frames through ONE call of the sequence pipeline (extract only): 8 pipes x sub-batches of 64 frames
    # (not under the profiler: its 64-frame launches would be averaged into the per-launch counters of the 512-frame call)
    from orb_slam2_ssd_semantic_amd import FramePipeline
    if not pipeline:
        del ext, frames, kps, desc, n
        torch.cuda.empty_cache()
        return ({"metric": "ORB extract frames/sec on 1920x1080 (BASELINE config 5, HBM-roofline stress)", "value": out["frames_per_s"],
                 "unit": "frames/s", "n_gpus": 1, "steps": reps, "warmup": warmup, "ms_per_step": out["ms_per_call"],
                 "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                 "config": {"workload": "BASELINE config 5: 512 HBM-resident 1920x1080 frames S(seed), 4000 features, 8 levels, one "
                                        "batched call per step, extract only (profiling run: no pipeline leg)", "frames_per_launch": nframes,
                            "width": w, "height": h, "nfeatures": nfeat, "workload_name": "S"},
                 "roofline": roof, "stages": stages, "config5": out} if standalone else out)
    k1, d1, n1 = kps.clone(), desc.clone(), n.clone()
    pl = FramePipeline(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=max(1, nframes // 8), npipes=8, device=local_rank)

    def onep():
        pl.extract_match_device(frames.data_ptr(), nframes, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), None, None,
                                flags=pl.NO_JOIN, stream=stream)
    for _ in range(warmup):
        onep()
    pl.synchronize()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        onep()
    pl.synchronize()
    torch.cuda.synchronize()
    out["frames_per_s_pipeline"] = round(nframes * reps / (time.perf_counter() - t), 1)
    out["pipeline"] = "one orbfe_pipeline_extract_match_device call (extract only), 8 pipes x sub-batches of %d frames" % max(1, nframes // 8)
    out["pipeline_equals_one_call"] = bool(torch.equal(kps, k1) and torch.equal(desc, d1) and torch.equal(n, n1))
    assert pl.overflow() == 0
    pl.close()
    del pl, k1, d1, n1
    if check:   # two frames of the timed call's output against the oracle (count, keypoint bit patterns, descriptors, order)
        from oracle import oracle_ffi as O
        oe = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        for f in (0, int(np.random.default_rng(5).integers(1, nframes))):
            ok, od = oe(frames[f].cpu().numpy())
            nd = int(n_host[f])
            gk = kps[f, :nd].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
            if not (nd == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(desc[f, :nd].cpu().numpy(), od)):
                raise SystemExit(f"bench.py config 5: frame {f} differs from the oracle")
        out["exact_checked"] = True
    del ext, frames, kps, desc, n
    torch.cuda.empty_cache()
    if standalone:
        return {"metric": "ORB extract frames/sec on 1920x1080 (BASELINE config 5, HBM-roofline stress)", "value": out["frames_per_s"],
                "unit": "frames/s", "n_gpus": 1, "steps": reps, "warmup": warmup, "ms_per_step": out["ms_per_call"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "BASELINE config 5: 512 HBM-resident 1920x1080 frames S(seed), 4000 features, 8 levels, one "
                                       "batched call per step, extract only", "frames_per_launch": nframes, "width": w, "height": h,
                           "nfeatures": nfeat, "workload_name": "S"},
                "roofline": roof, "stages": stages, "config5": out}
    return out


def config4_group(args, rank, local_rank, world, uid, allf, lo, hi, global_batch, nfeat, steps, ncand=8):
    """Config 4 through the C-ABI's own multi-device layer (include/orbfe.h orbfe_group_*, one rank per process): extract the
    shard into its slice of the padded blocks, ONE in-place ncclAllGather (RCCL, loaded by liborbfe.so itself) and the
    consumer of the gather (src/LoopClosing.cc:312-342): every own frame is brute-force matched against `ncand`
    candidate frames spread over the whole gathered batch, i.e. over the other ranks' shards.  Runs in a worker thread when
    world > 1 (config4_group_guarded): no torch collective in here, ranks meet on the CPU-side group CTL["leg"]."""
    from orb_slam2_ssd_semantic_amd.distributed import KeyframeGroup
    w, h = args.width, args.height
    torch.cuda.set_device(local_rank)   # the current device is per thread

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=CTL["leg"])

    with c_stdout_to_stderr():
        g = KeyframeGroup(nfeat, 1.2, 8, 20, 7, w, h, global_batch, rank_of_world=(rank, world, uid), device=local_rank)
    shard = allf[lo:hi].contiguous()
    nloc = hi - lo
    qf = np.repeat(np.arange(lo, hi), ncand)
    tf = (qf + 1 + (np.arange(len(qf)) % ncand) * (global_batch // ncand)) % global_batch   # candidates in every shard
    g.nframes = global_batch
    qb = torch.tensor([g.block_index(int(f)) for f in qf], dtype=torch.int32, device="cuda")
    tb = torch.tensor([g.block_index(int(f)) for f in tf], dtype=torch.int32, device="cuda")
    d_match = torch.zeros((len(qf), g.cap), dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(len(qf), dtype=torch.int32, device="cuda")

    def one():
        g.extract_shard_device(0, shard.data_ptr(), global_batch, w, h, w, w * h)
        g.allgather()
        g.match_device(0, qb.data_ptr(), tb.data_ptr(), len(qf), d_match.data_ptr(), d_nm.data_ptr(), 0.9, 100, True)
    for _ in range(2):
        one()
    g.synchronize()
    fence()
    t = time.perf_counter()
    for _ in range(steps):
        one()
    g.synchronize()
    fence()
    dt = time.perf_counter() - t
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=CTL["leg"])
        dt = float(tt[0])
    nm = d_nm.cpu().numpy()
    res = {"frames_per_s": round(global_batch * steps / dt, 1), "pairs_per_rank": int(len(qf)), "candidates_per_frame": ncand,
           "mean_matches_per_pair": round(float(nm.mean()), 1), "frames_per_gpu": nloc,
           "what": "orbfe_group_extract_shard_device + orbfe_group_allgather (in-place ncclAllGather of counts / keypoints / "
                   "descriptors) + orbfe_group_match_device (own frames x candidate frames of the gathered batch), strong scaling"}
    if rank == 0 and not args.no_cpu_baseline:   # two pairs against the oracle, one of them across the batch
        from oracle import oracle_ffi as O
        oe = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
        for p in (0, len(qf) - 1):
            (qk, qd), (tk, td) = oe(allf[int(qf[p])].cpu().numpy()), oe(allf[int(tf[p])].cpu().numpy())
            om, _, _, on = O.match_bf(qd, td, qk["angle"], tk["angle"], 0.9, 100, True)
            if not (int(nm[p]) == on and np.array_equal(d_match[p, :len(qk)].cpu().numpy(), om)):
                raise SystemExit(f"bench.py config 4: gathered-set match of pair {p} differs from the oracle")
        res["exact_checked"] = True
    g.close()
    return res


def config4_group_guarded(args, rank, local_rank, world, allf, lo, hi, global_batch, nfeat, steps, timeout_s=None):
    """config4_group behind a watchdog.  With world > 1 the library's own RCCL communicator runs here for the first time on a
    given node; a rank that fails or stalls inside it must cost this leg, not the line: the leg runs in a worker thread, every
    rank waits `timeout_s` for its own, and the ranks then agree on the CPU-side group CTL["main"] whether all finished.  If one
    did not, the leg is reported as such and every rank leaves through os._exit after rank 0 has printed the line (main)."""
    import threading
    from orb_slam2_ssd_semantic_amd.distributed import KeyframeGroup
    if timeout_s is None:
        timeout_s = float(os.environ.get("ORBFE_BENCH_GROUP_TIMEOUT", "240"))
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        with c_stdout_to_stderr():
            uid.copy_(torch.frombuffer(bytearray(KeyframeGroup.unique_id()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(uid, 0, group=CTL["main"])
    uid = bytes(uid.numpy().tobytes())
    threaded = world > 1 or os.environ.get("ORBFE_BENCH_GUARD_THREAD") == "1"   # the env: exercise the worker-thread path on one GPU
    if not threaded:
        return config4_group(args, rank, local_rank, world, uid, allf, lo, hi, global_batch, nfeat, steps)
    box = {}

    def work():
        try:
            box["res"] = config4_group(args, rank, local_rank, world, uid, allf, lo, hi, global_batch, nfeat, steps)
        except BaseException as e:   # noqa: BLE001 -- reported, see below
            box["err"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    stuck = th.is_alive()
    flags = torch.tensor([0 if (stuck or "err" in box) else 1, 0 if stuck else 1], dtype=torch.int32)
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=CTL["main"])
    if int(flags[1]) == 0:
        CTL["abandon"] = True
    if int(flags[0]) == 1:
        return box["res"]
    return {"error": box.get("err") or ("this rank did not finish within %.0f s" % timeout_s if stuck else "another rank failed or stalled"),
            "abandoned": bool(CTL.get("abandon"))}


def config4_leg(args, rank, local_rank, world, fence, global_batch=1024, nfeat=2000, steps=10):
    """BASELINE config 4: a 1024-frame keyframe batch S(10000+i), 2000 features, contiguous shards, one all-gather of
    counts + padded keypoints + descriptors.  Strong figure: the fixed global batch; weak: 1024 frames on every rank."""
    from orb_slam2_ssd_semantic_amd import ORBextractor
    w, h = args.width, args.height
    lo, hi = shard_range(global_batch, rank, world)
    S = -(-global_batch // world)
    ext = ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=global_batch, device=local_rank)
    cap = ext.capacity()
    stream = torch.cuda.current_stream().cuda_stream
    # S(10000 + i), i < 1024 -- every frame its own generator seed; every rank builds the same global batch
    # and takes its shard
    allf = torch.from_numpy(base_frames("S", global_batch, w, h, 10000)).cuda()
    same = bool(getattr(args, "same", False))
    red_dev = "cpu" if same else "cuda"   # the default group is CPU-only (gloo) when the ranks share one device
    out = {}

    def run(frames, nfr, tag):
        kps = torch.zeros((max(nfr, S), cap, 7), dtype=torch.int32, device="cuda")
        desc = torch.zeros((max(nfr, S), cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(max(nfr, S), dtype=torch.int32, device="cuda")

        def one():
            ext.extract_batch_device(frames.data_ptr(), nfr, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap,
                                     n.data_ptr(), stream)
            return all_gather_keyframes(n[:S] if tag == "strong" else n, kps[:S] if tag == "strong" else kps,
                                        desc[:S] if tag == "strong" else desc, host_staged=same)
        for _ in range(3):
            one()
        fence()
        t = time.perf_counter()
        for _ in range(steps):
            g = one()
        fence()
        dt = time.perf_counter() - t
        # the exchange step alone
        fence()
        t2 = time.perf_counter()
        for _ in range(steps):
            all_gather_keyframes(n[:S] if tag == "strong" else n, kps[:S] if tag == "strong" else kps,
                                 desc[:S] if tag == "strong" else desc, host_staged=same)
        fence()
        dtg = (time.perf_counter() - t2) / steps
        if world > 1:
            tt = torch.tensor([dt, dtg], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, dtg = float(tt[0]), float(tt[1])
        rows = S if tag == "strong" else nfr
        per_rank = rows * (4 + cap * 28 + cap * 32)
        frames_total = global_batch if tag == "strong" else nfr * world
        out[tag] = {"frames_per_s": round(frames_total * steps / dt, 1), "frames_per_gpu": nfr,
                    "allgather_ms": round(dtg * 1e3, 4), "allgather_bytes_per_rank": int(per_rank),
                    "allgather_bus_GBps": round(per_rank * (world - 1) / dtg / 1e9, 2) if world > 1 else None,
                    "gathered_frames": int(g[0].shape[0])}
        if tag == "strong" and rank == 0 and not args.no_cpu_baseline:
            # frames of the LAST rank's shard (and one of the own), read from the gathered blocks on rank 0, against the oracle
            from oracle import oracle_ffi as O
            from orb_slam2_ssd_semantic_amd import KP_DTYPE
            oe = O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
            lo_l, hi_l = shard_range(global_batch, world - 1, world)
            torch.cuda.synchronize()
            for owner, f in ((world - 1, lo_l), (world - 1, hi_l - 1), (0, hi - 1)):
                lo_o, _ = shard_range(global_batch, owner, world)
                row = owner * S + (f - lo_o)
                ok, od = oe(allf[f].cpu().numpy())
                nd = int(g[0][row].item())
                gk = g[1][row, :nd].cpu().numpy().copy().view(KP_DTYPE).reshape(-1)
                if not (nd == len(ok) and np.array_equal(gk.view(np.uint8), ok.view(np.uint8)) and np.array_equal(g[2][row, :nd].cpu().numpy(), od)):
                    raise SystemExit(f"bench.py config 4: frame {f} (rank {owner}'s shard) read from the gathered blocks differs from the oracle")
            out["gathered_exact_checked"] = True
        del kps, desc, n

    run(allf[lo:hi].contiguous(), hi - lo, "strong")
    run(allf, global_batch, "weak")
    if rank == 0:
        # stage table and roofline object of this shape (the library's HIP events on the launch stream; counter fields only from a
        # committed same-round profile of exactly this shape: `bench.py --nfeatures 2000 --no-match --no-extras`, profiles/r06_cfg4_*)
        nfr = hi - lo
        shard = allf[lo:hi].contiguous()
        kps = torch.zeros((nfr, cap, 7), dtype=torch.int32, device="cuda")
        desc = torch.zeros((nfr, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        ext.set_option("overlap", 0)
        for i in range(4):
            if i == 1:
                torch.cuda.synchronize()
                ext.set_profiling(True)
            ext.extract_batch_device(shard.data_ptr(), nfr, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), stream)
        torch.cuda.synchronize()
        st4 = ext.stage_ms()
        ext.set_profiling(False)
        ext.set_option("overlap", -1)
        roof4, stages4, ncand4 = stage_report(ext, st4, float(n.float().mean().item()), w, h, nfeat, nfr, "S", local_rank)
        out["roofline"], out["stages"], out["fast_candidates_frame0"] = roof4, stages4, ncand4
        del kps, desc, n, shard
    if world == 1:   # the shard through ONE call of the sequence pipeline (extract only): 4 pipes x sub-batches of a quarter shard
        from orb_slam2_ssd_semantic_amd import FramePipeline
        nfr = hi - lo
        pl = FramePipeline(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, sub_batch=max(1, -(-nfr // 4)), npipes=4, device=local_rank)
        kps = torch.zeros((nfr, cap, 7), dtype=torch.int32, device="cuda")
        desc = torch.zeros((nfr, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        shard = allf[lo:hi].contiguous()

        def onep():
            pl.extract_match_device(shard.data_ptr(), nfr, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), None, None,
                                    flags=pl.NO_JOIN, stream=stream)
        for _ in range(3):
            onep()
        pl.synchronize()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            onep()
        pl.synchronize()
        torch.cuda.synchronize()
        out["pipeline_frames_per_s"] = round(nfr * steps / (time.perf_counter() - t), 1)
        out["pipeline"] = "one orbfe_pipeline_extract_match_device call per batch (extract only), 4 pipes"
        assert pl.overflow() == 0
        pl.close()
        del pl, kps, desc, n
    try:
        out["cabi_group"] = config4_group_guarded(args, rank, local_rank, world, allf, lo, hi, global_batch, nfeat, steps)
    except Exception as e:   # a second communicator next to torch's: report, never lose the line over it
        out["cabi_group"] = {"error": f"{type(e).__name__}: {e}"}
    out["nfeatures"], out["global_batch"], out["cap"], out["n_gpus"] = nfeat, global_batch, cap, world
    out["note"] = (f"strong: the {global_batch}-frame batch S(10000 + i) sharded in contiguous blocks (SURVEY 8(d) row 4); weak: "
                   f"{global_batch} frames on every rank.  The all-gather is synchronous here (its cost is visible); the main timed "
                   "region overlaps it." + ("  SAME-DEVICE TEST MODE: host-staged gloo exchange, not a bandwidth figure." if same else ""))
    assert ext.overflow() == 0
    return out


if __name__ == "__main__":
    main()
