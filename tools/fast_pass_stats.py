#!/usr/bin/env python3
"""FAST work statistics of frames, on the CPU (numpy + the oracle's level / score taps): what decides between the dense and the
lane-compacting FAST kernel, and what an iniThFAST-first pass order (the reference's own, src/ORBextractor.cc:818-825) could save.

Per frame, summed over the 8 pyramid levels (detection window only):
  pass7 / pass20   share of PIXEL PAIRS (the kernel's unit: columns x, x + 1 of a 4-pixel lane) of which at least one pixel passes
                   the exact necessary test of k_fast_map_c (fast_compass_from: one of each opposite compass pair brighter /
                   darker than the centre by more than the threshold) at minThFAST = 7 / iniThFAST = 20
  corner7 / corner20  share of pixels that ARE FAST corners at the threshold (score map >= threshold)
  cand             candidates the reference's cell loop hands to DistributeOctTree (oracle tap; after per-cell NMS and fallback)
  cells, cells_fallback, cells_empty   FAST cells; cells whose cv::FAST(iniThFAST) came back empty and were re-run at minThFAST;
                   of those, cells still empty
  area_fallback    tile area (cell + 6) of the fallback cells / the window area: the share of the pixels a second pass touches

Usage: python tools/fast_pass_stats.py [--out profiles/r06_fast_pass_stats.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O                                     # noqa: E402
from orb_slam2_ssd_semantic_amd import photos                           # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like  # noqa: E402

EDGE = 16


def compass_pass(img, t):
    """H x W bool: the necessary test at threshold t (valid where the radius-3 circle is inside the image)"""
    a = img.astype(np.int16)
    v = a[3:-3, 3:-3]
    c0, c8 = a[6:, 3:-3], a[:-6, 3:-3]          # below / above (rows y + 3, y - 3)
    c4, c12 = a[3:-3, 6:], a[3:-3, :-6]         # right / left
    mb = np.minimum(np.maximum(c0, c8), np.maximum(c4, c12))
    md = np.maximum(np.minimum(c0, c8), np.minimum(c4, c12))
    out = np.zeros(img.shape, bool)
    out[3:-3, 3:-3] = (mb - v > t) | (v - md > t)
    return out


def frame_stats(img, oe, ini=20, mn=7):
    oe(img)
    tot = dict(px=0, pairs=0, p7=0, p20=0, c7=0, c20=0, cand=0, cells=0, fb=0, empty=0, area_fb=0)
    for l in range(oe.nlevels):
        lv = oe.level(l)
        h, w = lv.shape
        sc = O.fast_score_map(lv)
        x0, y0, x1, y1 = EDGE - 3, EDGE - 3, w - EDGE + 3, h - EDGE + 3      # minBorder .. maxBorder (:780-783)
        W, H = x1 - x0, y1 - y0
        ok, ncols, nrows, wcell, hcell = O.cell_grid(w, h)
        assert ok
        win = (slice(EDGE, h - EDGE), slice(EDGE, w - EDGE))                  # pixels that can be corners: tile interiors
        for t, key in ((mn, "p7"), (ini, "p20")):
            p = compass_pass(lv, t)[win]
            pw = p.shape[1] // 2 * 2
            tot[key] += int((p[:, 0:pw:2] | p[:, 1:pw:2]).sum())
        tot["pairs"] += (win[0].stop - win[0].start) * ((win[1].stop - win[1].start) // 2)
        tot["px"] += (win[0].stop - win[0].start) * (win[1].stop - win[1].start)
        tot["c7"] += int((sc[win] >= mn).sum())
        tot["c20"] += int((sc[win] >= ini).sum())
        tot["cand"] += len(oe.candidates(l))
        for i in range(nrows):
            iy = y0 + i * hcell
            my = min(iy + hcell + 6, y1)
            if iy >= y1 - 3:
                continue
            for j in range(ncols):
                ix = x0 + j * wcell
                mx = min(ix + wcell + 6, x1)
                if ix >= x1 - 6:
                    continue
                tot["cells"] += 1
                inner = sc[iy + 3:my - 3, ix + 3:mx - 3]
                if not (inner >= ini).any():
                    tot["fb"] += 1
                    tot["area_fb"] += (my - iy) * (mx - ix)
                    if not (inner >= mn).any():
                        tot["empty"] += 1
    return dict(pass7=tot["p7"] / tot["pairs"], pass20=tot["p20"] / tot["pairs"], corner7=tot["c7"] / tot["px"], corner20=tot["c20"] / tot["px"],
                cand=tot["cand"], cells=tot["cells"], cells_fallback=tot["fb"], cells_empty=tot["empty"], area_fallback=tot["area_fb"] / tot["px"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_fast_pass_stats.json"))
    ap.add_argument("--synth", type=int, default=8)
    a = ap.parse_args()
    oe = O.OracleExtractor(1000, 1.2, 8, 20, 7)
    rows = {}
    for s in range(a.synth):
        rows[f"S({s})"] = frame_stats(synth_frame(s), oe)
    for s in range(a.synth):
        rows[f"S_tum({s})"] = frame_stats(synth_tum_like(s), oe)
    for tag, g in photos.vga_gray_frames(both_flags=False):
        rows["photo:" + tag] = frame_stats(g, oe)

    def mean(prefix):
        sel = [v for k, v in rows.items() if k.startswith(prefix)]
        return {k: float(np.mean([r[k] for r in sel])) for k in sel[0]}
    summary = {"S": mean("S("), "S_tum": mean("S_tum("), "photos": mean("photo:")}
    json.dump(dict(what=__doc__.split("\n\n")[0], summary=summary, frames=rows), open(a.out, "w"), indent=1)
    print(f"{'frame':34s} pass7  pass20  corner7 corner20   cand  cells  fallback empty  area_fb")
    for k, r in list(rows.items()) + [("MEAN " + k, v) for k, v in summary.items()]:
        print(f"{k:34s} {r['pass7']:.3f}  {r['pass20']:.3f}   {r['corner7']:.4f}  {r['corner20']:.4f}  {r['cand']:6.0f}  {r['cells']:5.0f}  {r['cells_fallback']:6.1f}  {r['cells_empty']:5.1f}  {r['area_fallback']:.3f}")


if __name__ == "__main__":
    main()
