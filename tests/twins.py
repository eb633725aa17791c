"""Definition-level twins (numpy / pure Python) used to PIN the C oracle.

The reference ships no tests or golden vectors and OpenCV is absent, so each oracle stage is checked
against an independently written restatement of the published definition:
brute-force FAST arcs, closed-form bilinear resize and Gaussian blur, a literal std::list quadtree,
mask-based IC moments, vectorised rBRIEF, popcount Hamming and a plain-Python SearchByBoW.
Written for clarity, not speed: use on small inputs.
"""
import math

import numpy as np

F32 = np.float32

# FAST-9/16 Bresenham circle, radius 3
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_strength(img):
    """A(x,y) = max over the 16 nine-arcs and both polarities of min |difference| (signed), interior only."""
    img = img.astype(np.int32)
    h, w = img.shape
    out = np.full((h, w), -999, np.int32)
    if h < 7 or w < 7:
        return out
    c = img[3:h - 3, 3:w - 3]
    d = np.stack([c - img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])  # v - p_k
    best = np.full(c.shape, -999, np.int32)
    for k in range(16):
        idx = [(k + i) % 16 for i in range(9)]
        best = np.maximum(best, d[idx].min(0))      # dark arc: all p < v - t
        best = np.maximum(best, (-d[idx]).min(0))   # bright arc: all p > v + t
    out[3:h - 3, 3:w - 3] = best
    return out


def fast9(img, threshold, nonmax=True):
    """cv::FAST(img, kps, threshold, nonmax): list of (x, y, score) in raster order."""
    a = fast_strength(img)
    h, w = img.shape
    corner = a > threshold
    score = np.where(corner, a - 1, 0)
    res = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            s = score[y, x]
            if nonmax:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if not (s > nb.max()):
                    continue
            res.append((x, y, int(s) if nonmax else 0))
    return res


def cv_round(v):
    """cvRound: half-to-even."""
    return int(np.rint(v))


def resize_linear(src, dw, dh):
    """cv::resize INTER_LINEAR 8UC1, OpenCV 3.2 generic path, per-pixel closed form."""
    sh, sw = src.shape
    src = src.astype(np.int64)

    def axis(ssize, dsize, is_x):
        scale = 1.0 / (float(dsize) / float(ssize))
        res = []
        for d in range(dsize):
            f = F32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = F32(f - F32(s))
            if is_x:
                if s < 0:
                    f, s = F32(0), 0
                if s >= ssize - 1:
                    f, s = F32(0), ssize - 1
            c0 = cv_round(F32(F32(1.0) - f) * F32(2048))
            c1 = cv_round(f * F32(2048))
            res.append((s, c0, c1))
        return res

    xs, ys = axis(sw, dw, True), axis(sh, dh, False)
    dst = np.zeros((dh, dw), np.uint8)
    for dy, (sy, b0, b1) in enumerate(ys):
        y0 = min(max(sy, 0), sh - 1)
        y1 = min(max(sy + 1, 0), sh - 1)
        for dx, (sx, a0, a1) in enumerate(xs):
            x1 = min(sx + 1, sw - 1)
            h0 = int(src[y0, sx]) * a0 + int(src[y0, x1]) * a1
            h1 = int(src[y1, sx]) * a0 + int(src[y1, x1]) * a1
            dst[dy, dx] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    return dst


def gaussian_blur7(img, sse2=False):
    """GaussianBlur 7x7 sigma 2 REFLECT_101 with the 8-bit kernel."""
    x = np.arange(7) - 3.0
    k = np.exp(-0.5 / 4.0 * x * x).astype(np.float32)
    k = (k * (1.0 / k.astype(np.float64).sum())).astype(np.float32)
    ki = np.rint(k * F32(256)).astype(np.int64)
    assert ki.tolist() == [18, 34, 49, 55, 49, 34, 18]
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    h, w = img.shape
    rows = sum(ki[i] * p[:, i:i + w] for i in range(7))
    acc = sum(ki[j] * rows[j:j + h, :] for j in range(7))
    v = (acc + 32768) >> 16
    ties = (acc & 0xFFFF) == 0x8000
    if sse2:
        vec = np.zeros_like(ties)
        vec[:, :w & ~3] = True
        v = np.where(ties & vec & ((v & 1) == 1), v - 1, v)
    return np.minimum(v, 255).astype(np.uint8), int(ties.sum())


def copy_make_border101(img, b):
    return np.pad(img, b, mode="reflect")


def umax_table():
    """src/ORBextractor.cc:449-465."""
    hp = 15
    umax = [0] * 16
    vmax = int(math.floor(hp * math.sqrt(2.0) / 2 + 1))
    vmin = int(math.ceil(hp * math.sqrt(2.0) / 2))
    for v in range(vmax + 1):
        umax[v] = cv_round(math.sqrt(hp * hp - v * v))
    v0 = 0
    for v in range(hp, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    return umax


def ic_moments(img, x, y):
    um = umax_table()
    m10 = m01 = 0
    for v in range(-15, 16):
        d = um[abs(v)]
        for u in range(-d, d + 1):
            val = int(img[y + v, x + u])
            m10 += u * val
            m01 += v * val
    return m10, m01


def fast_atan2(y, x):
    """cv::fastAtan2 (OpenCV 3.2), float32 with separately rounded operations."""
    y, x = F32(y), F32(x)
    s = F32(180.0 / math.pi)
    p1, p3 = F32(0.9997878412794807) * s, F32(-0.3258083974640975) * s
    p5, p7 = F32(0.1555786518463281) * s, F32(-0.04432655554792128) * s
    eps = F32(2.220446049250313e-16)
    ax, ay = abs(x), abs(y)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = F32(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = F32(180.0) - a
    if y < 0:
        a = F32(360.0) - a
    return F32(a)


def sincos_correctly_rounded(angle_deg):
    """(float)cos / (float)sin of the fp32 radian angle, through libm double precision."""
    ang = F32(F32(angle_deg) * F32(math.pi / 180.0))
    return F32(math.cos(float(ang))), F32(math.sin(float(ang)))


def descriptor(blurred, x, y, angle_deg, pattern, a=None, b=None):
    """computeOrbDescriptor (src/ORBextractor.cc:92-131) with given or libm cos/sin."""
    if a is None:
        a, b = sincos_correctly_rounded(angle_deg)
    a, b = F32(a), F32(b)
    pat = np.asarray(pattern, np.float32).reshape(256, 4)
    out = np.zeros(32, np.uint8)

    def val(px, py):
        r = cv_round(F32(F32(px * b) + F32(py * a)))
        c = cv_round(F32(F32(px * a) - F32(py * b)))
        return int(blurred[y + r, x + c])

    for i in range(256):
        x0, y0, x1, y1 = pat[i]
        if val(x0, y0) < val(x1, y1):
            out[i // 8] |= 1 << (i % 8)
    return out


# ---- literal std::list quadtree (src/ORBextractor.cc:478-765) -----------------------------------------
class _Node:
    __slots__ = ("ulx", "uly", "urx", "bry", "keys", "seq")

    def __init__(self, ulx, uly, urx, bry):
        self.ulx, self.uly, self.urx, self.bry = ulx, uly, urx, bry
        self.keys = []
        self.seq = 0


def _divide(n):
    hx = int(math.ceil(F32(n.urx - n.ulx) / 2))
    hy = int(math.ceil(F32(n.bry - n.uly) / 2))
    mx, my = n.ulx + hx, n.uly + hy
    ch = [_Node(n.ulx, n.uly, mx, my), _Node(mx, n.uly, n.urx, my), _Node(n.ulx, my, mx, n.bry),
          _Node(mx, my, n.urx, n.bry)]
    for k in n.keys:
        if k[0] < mx:
            (ch[0] if k[1] < my else ch[2]).keys.append(k)
        else:
            (ch[1] if k[1] < my else ch[3]).keys.append(k)
    return ch


def distribute_octtree(cands, minx, maxx, miny, maxy, N):
    """cands: list of (x, y, response) in candidate order.  Pointer tie-break replaced by creation order."""
    # std::round(float): half away from zero (positive argument)
    n_ini = int(math.floor(float(F32(maxx - minx) / F32(maxy - miny)) + 0.5))
    hx = F32(maxx - minx) / F32(n_ini)
    nodes = []
    roots = []
    for i in range(n_ini):
        r = _Node(int(hx * F32(i)), 0, int(hx * F32(i + 1)), maxy - miny)
        roots.append(r)
        nodes.append(r)
    for k in cands:
        roots[int(F32(k[0]) / hx)].keys.append(tuple(k))
    nodes = [n for n in nodes if n.keys]
    seq = [0]

    def split_into(lst, node, collect):
        added = 0
        for c in _divide(node):
            if c.keys:
                lst.insert(0, c)
                if len(c.keys) > 1:
                    seq[0] += 1
                    c.seq = seq[0]
                    collect.append(c)
                    added += 1
        return added

    finish = False
    while not finish:
        prev = len(nodes)
        work = list(nodes)  # iteration visits the nodes present at the start, in list order
        to_expand = 0
        collect = []
        for n in work:
            if len(n.keys) == 1:
                continue
            to_expand += split_into(nodes, n, collect)
            nodes.remove(n)
        if len(nodes) >= N or len(nodes) == prev:
            finish = True
        elif len(nodes) + to_expand * 3 > N:
            while not finish:
                prev = len(nodes)
                prevc = sorted(collect, key=lambda c: (len(c.keys), c.seq))
                collect = []
                for n in reversed(prevc):
                    split_into(nodes, n, collect)
                    nodes.remove(n)
                    if len(nodes) >= N:
                        break
                if len(nodes) >= N or len(nodes) == prev:
                    finish = True
    out = []
    for n in nodes:
        best = n.keys[0]
        for k in n.keys[1:]:
            if k[2] > best[2]:
                best = k
        out.append(best)
    return out


# ---- matcher twins -------------------------------------------------------------------------------------
def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())


def hamming_matrix(q, t):
    q = np.asarray(q, np.uint8).reshape(-1, 32)
    t = np.asarray(t, np.uint8).reshape(-1, 32)
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    return lut[q[:, None, :] ^ t[None, :, :]].sum(-1)


def rot_bin(a1, a2):
    rot = F32(a1) - F32(a2)
    if rot < 0:
        rot = F32(rot + F32(360.0))
    v = float(F32(rot * F32(1.0 / 30)))
    b = int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))  # C round(): half away from zero
    return 0 if b == 30 else b


def three_maxima(counts):
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(counts):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if F32(max2) < F32(0.1) * F32(max1):
        i2 = i3 = -1
    elif F32(max3) < F32(0.1) * F32(max1):
        i3 = -1
    return i1, i2, i3


def _prune(matches, bins):
    """bins: list of (key, bin).  Returns number removed."""
    counts = [0] * 30
    for _, b in bins:
        counts[b] += 1
    keep = set(three_maxima(counts))
    removed = 0
    for k, b in bins:
        if b not in keep:
            matches[k] = -1
            removed += 1
    return removed


def match_bf(q, t, qa, ta, nnratio, th, check_ori):
    D = hamming_matrix(q, t)
    nq = D.shape[0]
    m = np.full(nq, -1, np.int32)
    best = np.full(nq, 256, np.int32)
    second = np.full(nq, 256, np.int32)
    bins = []
    nm = 0
    for i in range(nq):
        b1, b2, bi = 256, 256, -1
        for j in range(D.shape[1]):
            d = int(D[i, j])
            if d < b1:
                b2, b1, bi = b1, d, j
            elif d < b2:
                b2 = d
        best[i], second[i] = b1, b2
        if bi >= 0 and b1 <= th and F32(b1) < F32(nnratio) * F32(b2):
            m[i] = bi
            nm += 1
            if check_ori:
                bins.append((i, rot_bin(qa[i], ta[bi])))
    if check_ori:
        nm -= _prune(m, bins)
    return m, best, second, nm


def search_by_bow(descKF, validKF, angKF, fvKF, descF, validF, angF, fvF, nnratio, th_low, strict_lt, check_ori):
    """fvKF / fvF: dict node -> list of feature indices.  Literal walk of src/ORBmatcher.cc:242-335."""
    nF = len(descF)
    m = np.full(nF, -1, np.int32)
    bins = []
    nm = 0
    for node in sorted(set(fvKF) & set(fvF)):
        for rk in fvKF[node]:
            if validKF is not None and not validKF[rk]:
                continue
            b1, b2, bi = 256, 256, -1
            for rf in fvF[node]:
                if m[rf] >= 0:
                    continue
                if validF is not None and not validF[rf]:
                    continue
                d = hamming(descKF[rk], descF[rf])
                if d < b1:
                    b2, b1, bi = b1, d, rf
                elif d < b2:
                    b2 = d
            ok = (b1 < th_low) if strict_lt else (b1 <= th_low)
            if ok and bi >= 0 and F32(b1) < F32(nnratio) * F32(b2):
                m[bi] = rk
                nm += 1
                if check_ori:
                    bins.append((bi, rot_bin(angKF[rk], angF[bi])))
    if check_ori:
        nm -= _prune(m, bins)
    return m, nm


# ---- OpenCV 3.2's SIMD formulations, as executed on an x86-64 build (CV_SSE2) ---------------------------------------
# Independent of the definition-level twins above AND of the oracle: these emulate, intrinsic by intrinsic, what
# features2d/src/fast.cpp FAST_t<16> + fast_score.cpp cornerScore<16> and imgproc/src/imgwarp.cpp
# VResizeLinearVec_32s8u do with their registers (saturating unsigned byte arithmetic, the 0x80 xor that turns unsigned
# order into signed-compare order, counting by mask subtraction, 16-bit min/max ladders, _mm_mulhi_epi16), so that
# "the restatement reads the published scalar code right" and "the SIMD build computes the same thing" are two checks.
def _u8(a):
    return np.asarray(a).astype(np.uint8)


def _adds_epu8(a, b):
    return np.minimum(a.astype(np.int32) + b.astype(np.int32), 255).astype(np.uint8)


def _subs_epu8(a, b):
    return np.maximum(a.astype(np.int32) - b.astype(np.int32), 0).astype(np.uint8)


def _as_epi8(a):
    return a.astype(np.uint8).view(np.int8)


def _cmpgt_epi8(a, b):
    """signed byte compare -> 0xFF / 0x00 mask"""
    return np.where(_as_epi8(a) > _as_epi8(b), 0xFF, 0).astype(np.uint8)


def _fast_offsets():
    """makeOffsets: pixel[k] for k < 25 = CIRCLE[k % 16]"""
    return [CIRCLE[k % 16] for k in range(25)]


def corner_score16_sse2(img, xs, ys):
    """cornerScore<16>, SSE2 branch (fast_score.cpp): d[k] = v - ptr[pixel[k]] as short, k < 25; for k in {0, 8}: eight
    lanes j hold the arc starting at k + j: a = min(d[k+j+1 .. k+j+8]), b = max(same); q0 = max(q0, min(a, d[k+j]),
    min(a, d[k+j+9])), q1 = min(q1, max(b, d[k+j]), max(b, d[k+j+9])); q0 = max(q0, 0 - q1); horizontal max; result - 1.
    Note: unlike the scalar branch the SSE2 branch does not start from `threshold` (q0 = -1000)."""
    img = img.astype(np.int16)
    off = _fast_offsets()
    v = img[ys, xs]
    d = np.stack([v - img[ys + dy, xs + dx] for dx, dy in off], 1).astype(np.int16)      # (n, 25)
    n = len(xs)
    q0 = np.full((n, 8), -1000, np.int16)
    q1 = np.full((n, 8), 1000, np.int16)
    lanes = np.arange(8)
    for k in (0, 8):
        v0, v1 = d[:, k + 1 + lanes], d[:, k + 2 + lanes]
        a, b = np.minimum(v0, v1), np.maximum(v0, v1)
        for t in range(3, 9):
            v0 = d[:, k + t + lanes]
            a, b = np.minimum(a, v0), np.maximum(b, v0)
        v0 = d[:, k + lanes]
        q0 = np.maximum(q0, np.minimum(a, v0))
        q1 = np.minimum(q1, np.maximum(b, v0))
        v0 = d[:, k + 9 + lanes]
        q0 = np.maximum(q0, np.minimum(a, v0))
        q1 = np.minimum(q1, np.maximum(b, v0))
    q0 = np.maximum(q0, (np.int16(0) - q1).astype(np.int16))
    return (q0.max(1).astype(np.int32) - 1)


def fast9_sse2(img, threshold, nonmax=True, stats=None):
    """cv::FAST(img, kps, threshold, nonmax) as FAST_t<16> runs it with CV_SSE2: per row the 16-pixel SSE2 blocks (with the
    `mask == 0 -> skip 16`, `(mask & 255) == 0 -> step 8` early-outs of the 4-point pre-test), then the scalar
    threshold_tab tail for the last < 19 columns; scores by cornerScore<16> (SSE2 branch); NMS over the three row buffers.
    Returns [(x, y, score)] in OpenCV's output order."""
    img = _u8(img)
    h, w = img.shape
    if h < 7 or w < 7:
        return []
    threshold = min(max(int(threshold), 0), 255)
    t = np.uint8(threshold)
    off = _fast_offsets()
    K = 8
    ys, xs = np.mgrid[3:h - 3, 3:w - 3]
    v = img[3:h - 3, 3:w - 3]
    delta = np.uint8(0x80)
    v1 = _subs_epu8(v, np.full_like(v, t)) ^ delta          # v1 = (v -sat t) ^ 0x80
    v0 = _adds_epu8(v, np.full_like(v, t)) ^ delta          # v0 = (v +sat t) ^ 0x80
    ring = [img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in off]
    # 4-point pre-test on pixel[0], [4], [8], [12]: x_i = pix - 0x80 (== pix ^ 0x80 as bytes)
    x = [(ring[q] - delta).astype(np.uint8) for q in (0, 4, 8, 12)]
    m0 = np.zeros_like(v)
    m1 = np.zeros_like(v)
    for i in range(4):
        a, b = x[i], x[(i + 1) % 4]
        m0 |= _cmpgt_epi8(a, v0) & _cmpgt_epi8(b, v0)
        m1 |= _cmpgt_epi8(v1, a) & _cmpgt_epi8(v1, b)
    pre = (m0 | m1) != 0
    # full test: counters by mask subtraction, N = 25 positions
    c0 = np.zeros_like(v)
    c1 = np.zeros_like(v)
    max0 = np.zeros_like(v)
    max1 = np.zeros_like(v)
    for k in range(25):
        xk = ring[k] ^ delta
        mm0 = _cmpgt_epi8(xk, v0)
        mm1 = _cmpgt_epi8(v1, xk)
        c0 = ((c0.astype(np.int32) - mm0.astype(np.int32)) & 0xFF).astype(np.uint8) & mm0   # _mm_sub_epi8 wraps: c - 0xFF = c + 1
        c1 = ((c1.astype(np.int32) - mm1.astype(np.int32)) & 0xFF).astype(np.uint8) & mm1
        max0 = np.maximum(max0, c0)
        max1 = np.maximum(max1, c1)
    sse_corner = _as_epi8(np.maximum(max0, max1)) > np.int8(K)
    # scalar tail: threshold_tab + run counting
    d = [ring[k].astype(np.int32) - v.astype(np.int32) for k in range(25)]        # x - v
    tab = lambda q: np.where(d[q] < -threshold, 1, np.where(d[q] > threshold, 2, 0))
    dd = tab(0) | tab(8)
    for q in (2, 4, 6):
        dd = dd & (tab(q) | tab(q + 8))
    for q in (1, 3, 5, 7):
        dd = dd & (tab(q) | tab(q + 8))
    sc_corner = np.zeros(v.shape, bool)
    for bit, cond in ((1, lambda q: d[q] < -threshold), (2, lambda q: d[q] > threshold)):
        cnt = np.zeros(v.shape, np.int32)
        hit = np.zeros(v.shape, bool)
        for k in range(25):
            c = cond(k)
            cnt = np.where(c, cnt + 1, 0)
            hit |= cnt > K
        sc_corner |= hit & ((dd & bit) != 0)
    # row walk: which pixels each path classifies
    corner = np.zeros((h, w), bool)
    nblocks = nskip16 = nskip8 = 0
    for yy in range(h - 6):
        j = 3
        while j < w - 16 - 3:
            blk = pre[yy, j - 3:j - 3 + 16]
            if not blk.any():
                nskip16 += 1
                j += 16
                continue
            if not blk[:8].any():
                nskip8 += 1
                j += 8
                continue
            corner[yy + 3, j:j + 16] = sse_corner[yy, j - 3:j - 3 + 16]
            nblocks += 1
            j += 16
        corner[yy + 3, j:w - 3] = sc_corner[yy, j - 3:w - 6]
    if stats is not None:
        stats.update(blocks=nblocks, skip16=nskip16, skip8=nskip8,
                     missed=int((sse_corner & ~corner[3:h - 3, 3:w - 3]).sum()))   # corners the early-outs would have hidden
    cy, cx = np.nonzero(corner)
    if not nonmax:
        return [(int(a), int(b), 0) for a, b in zip(cx, cy)]
    score = np.zeros((h, w), np.int32)
    if len(cx):
        score[cy, cx] = corner_score16_sse2(img, cx, cy).astype(np.uint8)        # curr[j] = (uchar)cornerScore
    res = []
    p = np.pad(score, 1)
    for a, b in zip(cx, cy):
        s = score[b, a]
        nb = p[b:b + 3, a:a + 3].copy()
        nb[1, 1] = -1
        if s > nb.max():
            res.append((int(a), int(b), int(s)))
    return res


def _packs_epi32(a):
    return np.clip(a, -32768, 32767).astype(np.int16)


def _mulhi_epi16(a, b):
    return ((a.astype(np.int32) * b.astype(np.int32)) >> 16).astype(np.int16)


def _adds_epi16(a, b):
    return np.clip(a.astype(np.int32) + b.astype(np.int32), -32768, 32767).astype(np.int16)


def resize_linear_sse2(src, dw, dh, stats=None):
    """cv::resize INTER_LINEAR 8UC1 as an x86-64 OpenCV 3.2 build runs it: HResizeLinear<uchar,int,short,2048,HResizeNoVec>
    (scalar) into int rows, then VResizeLinearVec_32s8u on the first dw - (dw % 4 or 4) .. columns -- `srai 4`,
    `packs_epi32`, `mulhi_epi16(x, b0) +sat mulhi_epi16(y, b1)`, `+sat 2`, `srai 2`, `packus` -- and the scalar
    FixedPtCast tail for the rest.  Both forms are evaluated for every pixel and asserted equal where each applies."""
    src = _u8(src)
    sh, sw = src.shape

    def axis(ssize, dsize, is_x):
        scale = 1.0 / (float(dsize) / float(ssize))
        ofs, c0, c1 = [], [], []
        for dd in range(dsize):
            f = F32((dd + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = F32(f - F32(s))
            if is_x:
                if s < 0:
                    f, s = F32(0), 0
                if s >= ssize - 1:
                    f, s = F32(0), ssize - 1
            ofs.append(s)
            c0.append(cv_round(F32(F32(1.0) - f) * F32(2048)))
            c1.append(cv_round(f * F32(2048)))
        return np.array(ofs), np.array(c0, np.int16), np.array(c1, np.int16)

    xo, a0, a1 = axis(sw, dw, True)
    yo, b0, b1 = axis(sh, dh, False)
    s32 = src.astype(np.int32)
    x1 = np.minimum(xo + 1, sw - 1)
    H = s32[:, xo] * a0.astype(np.int32) + s32[:, x1] * a1.astype(np.int32)       # int rows (sh, dw)
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    S0, S1 = H[y0], H[y1]                                                            # (dh, dw) int32
    # SSE2 kernel
    X = _packs_epi32(S0 >> 4)
    Y = _packs_epi32(S1 >> 4)
    if stats is not None:
        stats["packs_saturated"] = int(((S0 >> 4) > 32767).sum() + ((S1 >> 4) > 32767).sum())
    r = _adds_epi16(_mulhi_epi16(X, b0[:, None]), _mulhi_epi16(Y, b1[:, None]))
    r = (_adds_epi16(r, np.int16(2)) >> 2)
    vec = np.clip(r, 0, 255).astype(np.uint8)                                        # packus_epi16
    # scalar tail: uchar(( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
    sc = ((((b0.astype(np.int32)[:, None] * (S0 >> 4)) >> 16) + ((b1.astype(np.int32)[:, None] * (S1 >> 4)) >> 16) + 2) >> 2)
    sc = (sc & 0xFF).astype(np.uint8)
    # the vector loops cover x <= width - 16 by 16 and then x < width - 4 by 4; the rest is scalar
    xv = 0
    while xv <= dw - 16:
        xv += 16
    while xv < dw - 4:
        xv += 4
    out = sc.copy()
    out[:, :xv] = vec[:, :xv]
    if stats is not None:
        stats["vector_columns"] = xv
        stats["forms_differ"] = int((vec != sc).sum())
    return out
