"""Input side of the hot path for the TUM RGB-D sequences (BASELINE configs 1-3): the association list and the colour ->
gray conversion the caller applies before ORBextractor::operator() sees the image.

  * tool/associate.txt lines: `t_rgb rgb/<t>.png t_depth depth/<t>.png` (827 for fr3/walking_xyz); the reference's
    Examples read them in order (perfect/Examples/RGB-D/rgbd_tum.cc LoadImages).
  * Tracking::GrabImageRGBD (src/Tracking.cc:339-353): cv::imread delivers BGR memory; with `Camera.RGB: 1`
    (TUM3.yaml:28) the frame goes through cvtColor(CV_RGB2GRAY), i.e. OpenCV's 14-bit fixed point luma with the R weight
    applied to the FIRST byte in memory (blue):      gray = (4899*c0 + 9617*c1 + 1868*c2 + 8192) >> 14
    With Camera.RGB: 0 it is CV_BGR2GRAY:             gray = (1868*c0 + 9617*c1 + 4899*c2 + 8192) >> 14
The dataset is not in the container: point $TUM_FR3_WALKING_XYZ at an extracted sequence to use it (bench.py --tum,
tests/test_gpu_fuzz.py::test_tum_sequence); everything else falls back to the synthetic generators of synth.py.
"""
import os

import numpy as np

R2Y, G2Y, B2Y, SHIFT = 4899, 9617, 1868, 14  # OpenCV color.cpp (CV_DESCALE, yuv_shift = 14)


def gray_from_interleaved(img3, rgb_flag=True):
    """img3: H x W x 3 uint8 in MEMORY order (c0, c1, c2) as cv::imread returns it (B, G, R).
    rgb_flag = the reference's mbRGB (Camera.RGB): True -> CV_RGB2GRAY on that memory, False -> CV_BGR2GRAY."""
    a = np.asarray(img3)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
    w0, w2 = (R2Y, B2Y) if rgb_flag else (B2Y, R2Y)
    acc = a[..., 0].astype(np.int32) * w0 + a[..., 1].astype(np.int32) * G2Y + a[..., 2].astype(np.int32) * w2
    return ((acc + (1 << (SHIFT - 1))) >> SHIFT).astype(np.uint8)


def load_associations(path):
    """[(t_rgb, rgb_file, t_depth, depth_file)] in file order; blank lines and '#' comments skipped."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 4 or p[0].startswith("#"):
                continue
            out.append((float(p[0]), p[1], float(p[2]), p[3]))
    return out


def sequence_dir():
    d = os.environ.get("TUM_FR3_WALKING_XYZ", "")
    return d if d and os.path.isdir(d) else None


def default_association_file(seq_dir):
    """associate.txt / associations.txt next to the images, else $TUM_ASSOCIATE (e.g. the reference's tool/associate.txt)"""
    for cand in (os.path.join(seq_dir, "associate.txt"), os.path.join(seq_dir, "associations.txt"),
                 os.environ.get("TUM_ASSOCIATE", "")):
        if cand and os.path.exists(cand):
            return cand
    return None


def rgb_list(seq_dir):
    """fallback when no association file exists: every image of the dataset's own rgb.txt (`timestamp filename`)"""
    out = []
    with open(os.path.join(seq_dir, "rgb.txt")) as f:
        for line in f:
            p = line.split()
            if len(p) >= 2 and not p[0].startswith("#"):
                out.append((float(p[0]), p[1], float(p[0]), ""))
    return out


def read_bgr(path):
    """PNG/JPEG -> H x W x 3 uint8 in cv::imread's memory order (B, G, R)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def load_gray_frames(seq_dir=None, assoc=None, limit=None, rgb_flag=True):
    """The gray frames ORBextractor sees on the sequence, in association order: uint8 [N, H, W]."""
    seq_dir = seq_dir or sequence_dir()
    if not seq_dir:
        raise FileNotFoundError("TUM sequence not available: set $TUM_FR3_WALKING_XYZ")
    assoc = assoc or default_association_file(seq_dir)
    rows = load_associations(assoc) if assoc else rgb_list(seq_dir)
    if limit:
        rows = rows[:limit]
    return np.stack([gray_from_interleaved(read_bgr(os.path.join(seq_dir, r[1])), rgb_flag) for r in rows])
