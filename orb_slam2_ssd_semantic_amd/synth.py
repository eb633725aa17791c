"""Deterministic synthetic frames S(seed): "random textured tiles".

No TUM data exists in the container or on the GPU box, so every config of BASELINE.json falls back
to these seeded frames (numpy PCG64, bit-reproducible across machines).
"""
import numpy as np


def synth_frame(seed, h=480, w=640, sparse=False):
    rng = np.random.default_rng(seed)
    acc = np.zeros((h, w), np.float64)
    for s, wt in ((4, 0.4), (8, 0.3), (16, 0.2), (32, 0.1)):
        g = rng.integers(0, 256, (-(-h // s), -(-w // s)), dtype=np.uint8)
        up = np.repeat(np.repeat(g, s, axis=0), s, axis=1)[:h, :w]
        acc += wt * up
    acc += rng.normal(0.0, 2.0, (h, w))
    if sparse:  # smooth vignette: outer cells fall back to minThFAST or stay empty
        yy, xx = np.mgrid[0:h, 0:w]
        r2 = ((yy - h / 2) / (h / 2)) ** 2 + ((xx - w / 2) / (w / 2)) ** 2
        acc = 128 + (acc - 128) * np.clip(1.2 - r2, 0.0, 1.0) ** 2
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def synth_batch(seed0, n, h=480, w=640, sparse=False):
    return np.stack([synth_frame(seed0 + i, h, w, sparse) for i in range(n)])


# rational rotations (exact cos / sin): no libm in the generator, so frames are bit-reproducible across machines
_ROT = ((1.0, 0.0), (0.0, 1.0), (0.6, 0.8), (0.8, 0.6), (5 / 13, 12 / 13), (12 / 13, 5 / 13), (8 / 17, 15 / 17),
        (-0.6, 0.8), (-0.8, 0.6), (-5 / 13, 12 / 13))


def _value_noise(rng, h, w, s):
    """Gaussian lattice noise of period s, bilinearly interpolated to h x w (only mul/add: reproducible)."""
    gh, gw = h // s + 2, w // s + 2
    g = rng.normal(0.0, 1.0, (gh, gw))
    y = np.arange(h) / s
    x = np.arange(w) / s
    y0 = y.astype(np.int64)
    x0 = x.astype(np.int64)
    fy = (y - y0)[:, None]
    fx = (x - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _binomial5(img):
    """[1 4 6 4 1] / 16 separable smoothing (sigma 1.0; dyadic weights, edge-replicated): lens + demosaic blur."""
    p = np.pad(img, 2, mode="edge")
    r = (p[:, :-4] + 4 * p[:, 1:-3] + 6 * p[:, 2:-2] + 4 * p[:, 3:-1] + p[:, 4:]) / 16.0
    return (r[:-4] + 4 * r[1:-3] + 6 * r[2:-2] + 4 * r[3:-1] + r[4:]) / 16.0


def synth_tum_like(seed, h=480, w=640):
    """S_tum(seed): frames with the corner statistics of indoor camera images (TUM RGB-D is not in the container).

    Multi-octave (1/f-like) texture at natural contrast, 10-20 piecewise-smooth "objects" (rotated boxes / ellipses
    with their own albedo, mostly flat), a few fine-print patches, optics blur and sensor noise (sigma 1.6).  The
    reference's FAST stage finds a few thousand NMS candidates per frame on these (3-10 k on TUM)
    instead of the ~37 k of the corner-saturated S(seed); about 2.5 % of the pixels are FAST corners at minThFAST.
    """
    rng = np.random.default_rng(1_000_000 + seed)
    tex = np.zeros((h, w), np.float64)
    for s, amp in ((2, 0.7), (4, 1.0), (8, 1.3), (16, 1.7), (32, 2.4), (64, 3.4), (128, 4.6)):
        tex += amp * _value_noise(rng, h, w, s)
    tex /= 7.0
    img = 118.0 + 34.0 * tex
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for _ in range(int(rng.integers(10, 22))):
        cy, cx = float(rng.integers(0, h)), float(rng.integers(0, w))
        a, b = float(rng.integers(12, 140)), float(rng.integers(12, 140))
        co, si = _ROT[int(rng.integers(0, len(_ROT)))]
        u = (xx - cx) * co + (yy - cy) * si
        v = (yy - cy) * co - (xx - cx) * si
        if rng.random() < 0.7:
            m = (np.abs(u) < a) & (np.abs(v) < b)
        else:
            m = u * u * (b * b) + v * v * (a * a) < a * a * b * b
        alb = float(rng.integers(30, 221))
        k = float(rng.integers(1, 11)) / 20.0
        img = np.where(m, alb + k * (img - 118.0), img)
    for _ in range(int(rng.integers(2, 6))):   # fine print / keyboards
        y0, x0 = int(rng.integers(0, h - 60)), int(rng.integers(0, w - 90))
        ph, pw, s = int(rng.integers(30, 60)), int(rng.integers(40, 90)), int(rng.integers(3, 7))
        g = rng.integers(0, 2, (-(-ph // s), -(-pw // s))) * float(rng.integers(40, 121))
        patch = np.repeat(np.repeat(g, s, 0), s, 1)[:ph, :pw]
        img[y0:y0 + ph, x0:x0 + pw] = img[y0:y0 + ph, x0:x0 + pw] * 0.25 + patch + 40.0
    img = _binomial5(img) + rng.normal(0.0, 1.6, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def regular_vocabulary(k=10, L=6, seed=0, zero_frac=0.0):
    """Random DBoW2-shaped vocabulary tree (regular k-ary, depth L, breadth-first ids: children after parents) as the
    arrays orbfe_vocabulary_create takes.  ORBvoc has k = 10, L = 6 (1 111 111 nodes, 10^6 words); the real file is
    not in the container, so node descriptors are random and weights uniform -- the work per descriptor (L x k
    Hamming distances) and the FeatureVector shape (k^(L - levelsup) nodes) are the real ones."""
    rng = np.random.default_rng(seed)
    level_start = np.concatenate([[0], np.cumsum([k ** l for l in range(L + 1)])])
    nodes = int(level_start[-1])
    inner = int(level_start[L])                      # nodes that have children
    child_off = np.concatenate([np.arange(inner + 1, dtype=np.int64) * k, np.full(nodes - inner, inner * k, np.int64)])
    child_idx = np.arange(1, nodes, dtype=np.uint32)  # children of node i are 1 + i*k .. 1 + i*k + k - 1
    node_desc = rng.integers(0, 256, (nodes, 32), dtype=np.uint8)
    word_id = np.zeros(nodes, np.uint32)
    word_id[inner:] = np.arange(nodes - inner, dtype=np.uint32)
    weight = rng.uniform(0.1, 9.0, nodes)
    if zero_frac > 0:
        weight[rng.random(nodes) < zero_frac] = 0.0
    return dict(child_off=child_off.astype(np.uint32), child_idx=child_idx, node_desc=node_desc, word_id=word_id,
                weight=weight, L=L)


def _gen_chunk(job):
    """(generator name, first seed, count, h, w) -> uint8 [count, h, w]"""
    gen, seed0, count, h, w = job
    make = synth_frame if gen == "S" else synth_tum_like
    return np.stack([make(seed0 + i, h, w) for i in range(count)])


def synth_frames_parallel(gen, n, h, w, seed0, max_procs=64):
    """n frames gen(seed0 .. seed0 + n - 1), generated by worker PROCESSES started as `python -m ...synth` (plain interpreters
    that import numpy and this module only -- a multiprocessing pool would re-import the caller's main module, i.e. torch,
    in every worker; and nothing is forked from a process that has a GPU runtime open).  Each worker writes its chunk as a
    raw file into a temporary directory.  The frames are exactly synth_frame / synth_tum_like of those seeds."""
    import os
    import subprocess
    import sys
    import tempfile
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))   # ranks of one node share the host's cores
    per_min = 16 if gen == "S" else 2   # S takes 15 ms a frame, S_tum 0.36 s: worth a process for two frames
    procs = max(1, min(max_procs, (os.cpu_count() or 1) // world, n // per_min))
    if procs == 1:
        return _gen_chunk((gen, seed0, n, h, w))
    per = -(-n // procs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = np.empty((n, h, w), np.uint8)
    with tempfile.TemporaryDirectory(prefix="orbfe_synth_") as td:
        jobs = []
        for k, a in enumerate(range(0, n, per)):
            cnt = min(per, n - a)
            path = os.path.join(td, f"{k}.raw")
            p = subprocess.Popen([sys.executable, "-m", "orb_slam2_ssd_semantic_amd.synth", gen, str(seed0 + a), str(cnt), str(h), str(w), path],
                                 cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", "")))
            jobs.append((a, cnt, path, p))
        for a, cnt, path, p in jobs:
            if p.wait() != 0:
                raise RuntimeError("frame generator worker failed")
            out[a:a + cnt] = np.fromfile(path, np.uint8).reshape(cnt, h, w)
    return out


if __name__ == "__main__":   # worker: gen seed0 count h w outfile
    import sys
    _g, _s, _c, _h, _w, _o = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    _gen_chunk((_g, _s, _c, _h, _w)).tofile(_o)
