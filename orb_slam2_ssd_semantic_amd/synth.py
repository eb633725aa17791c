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
