"""The real-photograph frame set (tests/golden/real/): the stand-in for "identical TUM RGB-D frames" on boxes without the dataset.

TUM fr3/walking_xyz is on no box this project runs on; the build container does hold real photographs (scikit-image's data
directory, scipy.misc's face / ascent).  tests/golden/make_real_images.py turned them into the frames the extractor would be
handed -- 640 x 480 colour frames in cv::imread's B,G,R memory order (so the caller's gray conversion, src/Tracking.cc:339-353, is
in the path with both Camera.RGB settings), single-channel frames, native odd sizes, JPEG re-encodes -- and
tests/golden/make_real_golden.py recorded what the COMPILED REFERENCE (oracle/_ref) extracts from them.  This module only reads
the image files; it is input plumbing like tum.py (bench.py's `value_real_photo` leg, tests/test_real_photos.py,
tests/test_gpu_real_photos.py, tools/fast_pass_stats.py).
"""
import hashlib
import os

import numpy as np

from .tum import gray_from_interleaved

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real")


def available():
    return os.path.isdir(os.path.join(ROOT, "vga"))


def _read(path):
    """PNG / JPEG -> H x W uint8 (single channel) or H x W x 3 uint8 in cv::imread's memory order (B, G, R)"""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "L":
        return np.ascontiguousarray(np.asarray(im))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def _names(sub, ext):
    d = os.path.join(ROOT, sub)
    return sorted(f[:-len(ext)] for f in os.listdir(d) if f.endswith(ext))


def vga_images():
    """[(name, array)]: 640 x 480, colour ones H x W x 3 (B, G, R), the others H x W"""
    return [(n, _read(os.path.join(ROOT, "vga", n + ".png"))) for n in _names("vga", ".png")]


def jpeg_images():
    return [(n, _read(os.path.join(ROOT, "jpeg", n + ".jpg"))) for n in _names("jpeg", ".jpg")]


def native_images():
    """[(name, gray array)] at the sources' own sizes"""
    return [(n, _read(os.path.join(ROOT, "native", n + ".png"))) for n in _names("native", ".png")]


def vga_gray_frames(both_flags=True, jpeg=True):
    """The gray frames ORBextractor sees, [(tag, H x W uint8)] in a fixed order: every colour image through the caller's
    conversion with Camera.RGB = 1 (tag `<name>@rgb1`: CV_RGB2GRAY applied to B,G,R memory, what TUM3.yaml selects) and, with
    both_flags, Camera.RGB = 0 (`@rgb0`); single-channel images as they are (`<name>`); the JPEG re-encodes with Camera.RGB = 1."""
    out = []
    for name, a in vga_images():
        if a.ndim == 3:
            out.append((name + "@rgb1", gray_from_interleaved(a, True)))
            if both_flags:
                out.append((name + "@rgb0", gray_from_interleaved(a, False)))
        else:
            out.append((name, a))
    if jpeg:
        for name, a in jpeg_images():
            out.append((name + "@rgb1", gray_from_interleaved(a, True)))
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
