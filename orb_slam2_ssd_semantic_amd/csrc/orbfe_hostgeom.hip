// orbfe_hostgeom.hip -- HOST-side parts of the projection-gated matchers, behind the C-ABI so that a host
// shim only flattens its objects, calls in, and replays decisions:
//   orbfe_project_points        pose projection + depth / image / distance / viewing-angle gates of the SearchByProjection
//                               family, Fuse x2 and SearchBySim3 (src/ORBmatcher.cc:401-433, :1060-1101, :1224-1255,
//                               :1389-1424, :1620-1642, :1778-1803), one call per list of map points
//   orbfe_proj_queries_local_map  the (trivial) gating of SearchByProjection(Frame&, vector<MapPoint*>&, th) (:63-93)
//   orbfe_rotation_consistency  the rotation histogram all matchers end with (:308-316 binning, :1912-1957 three maxima)
//   orbfe_initialization_resolve  SearchForInitialization's in-order acceptance rule on the device's window lists (:547-617)
// Plain float / double arithmetic in a FIXED, documented operation order (this file is compiled with -ffp-contract=off, so
// nothing is fused): the order is the one cv::Mat expressions of the reference evaluate to -- 3x3 * 3x1 products accumulate
// left to right in float (OpenCV's small-matrix gemm path and the test stub alike), norms and dot products in double.
// No device code here: the extension .hip only puts the file into the same build.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "orbfe_common.h"

namespace
{
// r = M (row-major 3x3) * x + t, each product and each sum rounded to float, summed left to right
inline void affine3(const float *M, const float *t, const float *x, float *r)
{
    for (int k = 0; k < 3; ++k) {
        float acc = M[3 * k] * x[0];
        acc += M[3 * k + 1] * x[1];
        acc += M[3 * k + 2] * x[2];
        r[k] = acc + t[k];
    }
}
inline double dot3d(const float *a, const float *b) { return (double)a[0] * (double)b[0] + ((double)a[1] * (double)b[1]) + ((double)a[2] * (double)b[2]); }
}  // namespace

extern "C" orbfe_status orbfe_project_points(const float *R, const float *t, const float *R2, const float *t2, const float *Ow, float fx,
                                             float fy, float cx, float cy, float bf, float minx, float maxx, float miny, float maxy,
                                             int32_t flags, int32_t n, const float *world_pos, const float *normal,
                                             const float *min_dist, const float *max_dist, float *u, float *v, float *invz, float *dist,
                                             float *ur, uint8_t *ok)
{
    if (n < 0 || !R || !t || (n > 0 && (!world_pos || !u || !v || !ok)) || ((min_dist == nullptr) != (max_dist == nullptr)) ||
        (R2 == nullptr) != (t2 == nullptr)) {
        orbfe_set_error("bad argument to orbfe_project_points");
        return ORBFE_ERR_ARG;
    }
    for (int i = 0; i < n; ++i) {
        const float *xw = world_pos + 3 * (size_t)i;
        ok[i] = 0;
        float pc[3];
        affine3(R, t, xw, pc);
        if (R2) {  // SearchBySim3: camera 1 -> camera 2 under the similarity
            float p2[3];
            affine3(R2, t2, pc, p2);
            pc[0] = p2[0]; pc[1] = p2[1]; pc[2] = p2[2];
        }
        if ((flags & ORBFE_PJ_SKIP_NEG_DEPTH) && pc[2] < 0.0f) continue;
        const float iz = 1.0f / pc[2];   // `1.0 / z` in double and `1 / z` in float round to the same float (53 >= 2 * 24 + 2)
        if ((flags & ORBFE_PJ_SKIP_NEG_INVZ) && iz < 0.0f) continue;
        float pu, pv;
        if (flags & ORBFE_PJ_UV_CHAINED) {   // fx * xc * invzc + cx  (:1626, :1789)
            pu = fx * pc[0] * iz + cx;
            pv = fy * pc[1] * iz + cy;
        } else {                             // x = X * invz; u = fx * x + cx  (:413-417, :1072-1076)
            const float nx = pc[0] * iz, ny = pc[1] * iz;
            pu = fx * nx + cx;
            pv = fy * ny + cy;
        }
        if (flags & ORBFE_PJ_BOUNDS_CLOSED) {   // u < min || u > max  ->  out (:1629-1632)
            if (pu < minx || pu > maxx || pv < miny || pv > maxy) continue;
        } else {                                // KeyFrame::IsInImage: min <= u < max
            if (!(pu >= minx && pu < maxx && pv >= miny && pv < maxy)) continue;
        }
        float d3 = 0.f;
        if (min_dist) {
            // |PO| with PO = Pw - Ow in float, or |Pc| when there is no centre (:1417); the square sum and the root in double
            float po[3] = {pc[0], pc[1], pc[2]};
            if (Ow) { po[0] = xw[0] - Ow[0]; po[1] = xw[1] - Ow[1]; po[2] = xw[2] - Ow[2]; }
            d3 = (float)sqrt(dot3d(po, po));
            if (d3 < min_dist[i] || d3 > max_dist[i]) continue;
            if (normal && Ow && dot3d(po, normal + 3 * (size_t)i) < 0.5 * (double)d3) continue;   // viewing angle > 60 deg (:428, :1096)
        }
        u[i] = pu;
        v[i] = pv;
        if (invz) invz[i] = iz;
        if (dist) dist[i] = d3;
        if (ur) ur[i] = pu - bf * iz;
        ok[i] = 1;
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_proj_queries_local_map(const float *scale_factors, int32_t n, const uint8_t *in_view, const uint8_t *bad,
                                                     const int32_t *level, const float *view_cos, const float *proj_uvr,
                                                     const uint8_t *obs_gt0, float th, orbfe_proj_query *q, int32_t *src, int32_t *nq)
{
    if (n < 0 || !nq || (n > 0 && (!scale_factors || !in_view || !bad || !level || !view_cos || !proj_uvr || !obs_gt0 || !q || !src))) {
        orbfe_set_error("bad argument to orbfe_proj_queries_local_map");
        return ORBFE_ERR_ARG;
    }
    const bool widen = th != 1.0f;   // :67
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (!in_view[i] || bad[i]) continue;
        float rad = (double)view_cos[i] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos (:159-165), the comparison in double
        if (widen) rad *= th;
        orbfe_proj_query e;
        e.u = proj_uvr[3 * (size_t)i];
        e.v = proj_uvr[3 * (size_t)i + 1];
        e.r = rad * scale_factors[level[i]];
        e.min_level = level[i] - 1;
        e.max_level = level[i];
        e.ur = proj_uvr[3 * (size_t)i + 2];
        e.flags = ORBFE_PROJ_RIGHT_GATE | (obs_gt0[i] ? ORBFE_PROJ_CLAIMS : 0);
        e.pad = 0;
        q[k] = e;
        src[k++] = i;
    }
    *nq = k;
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_rotation_consistency(const float *angle_a, const float *angle_b, int32_t n, int32_t histo_len, uint8_t *drop)
{
    if (n < 0 || histo_len < 1 || histo_len > 360 || (n > 0 && (!angle_a || !angle_b || !drop))) {
        orbfe_set_error("bad argument to orbfe_rotation_consistency");
        return ORBFE_ERR_ARG;
    }
    std::vector<int> count((size_t)histo_len, 0), bin_of((size_t)n, 0);
    const float inv = 1.0f / (float)histo_len;   // the reference multiplies by 1 / HISTO_LENGTH: bins 0 .. 12 in practice (sic)
    for (int i = 0; i < n; ++i) {
        float d = angle_a[i] - angle_b[i];
        if (d < 0.0f) d += 360.0f;
        const float fb = roundf(d * inv);
        // the reference asserts bin >= 0 && bin < HISTO_LENGTH (src/ORBmatcher.cc:314); here an angle pair whose bin falls outside
        // the histogram (NaN, angles outside [0, 360), or a histo_len below 19 with ordinary angles) is an argument error
        if (!(fb >= 0.0f && fb <= (float)histo_len)) {
            orbfe_set_error("orbfe_rotation_consistency: match %d (angles %g, %g) falls into bin %g of %d", i, (double)angle_a[i],
                            (double)angle_b[i], (double)fb, histo_len);
            return ORBFE_ERR_ARG;
        }
        int b = (int)fb;
        if (b == histo_len) b = 0;
        bin_of[(size_t)i] = b;
        count[(size_t)b]++;
    }
    // the three fullest bins, an earlier bin wins a tie; second / third dropped under a tenth of the first (:1912-1957)
    int top[3] = {0, 0, 0}, idx[3] = {-1, -1, -1};
    for (int b = 0; b < histo_len; ++b) {
        const int c = count[(size_t)b];
        int r = 3;
        while (r > 0 && c > top[r - 1]) --r;
        if (r == 3) continue;
        for (int k = 2; k > r; --k) { top[k] = top[k - 1]; idx[k] = idx[k - 1]; }
        top[r] = c;
        idx[r] = b;
    }
    if ((float)top[1] < 0.1f * (float)top[0]) idx[1] = idx[2] = -1;
    else if ((float)top[2] < 0.1f * (float)top[0]) idx[2] = -1;
    for (int i = 0; i < n; ++i) {
        const int b = bin_of[(size_t)i];
        drop[i] = !(b == idx[0] || b == idx[1] || b == idx[2]);
    }
    return ORBFE_OK;
}

extern "C" orbfe_status orbfe_initialization_resolve(const uint32_t *off, const uint32_t *ent, int32_t nq, int32_t n2, int32_t th,
                                                     float nnratio, int32_t *accepted, int32_t *holder)
{
    if (nq < 0 || n2 < 0 || !off || (nq > 0 && !accepted) || (n2 > 0 && !holder) || (nq > 0 && off[nq] > 0 && !ent)) {
        orbfe_set_error("bad argument to orbfe_initialization_resolve");
        return ORBFE_ERR_ARG;
    }
    std::vector<int> held((size_t)n2, INT32_MAX);   // distance at which a second-frame feature is currently held
    for (int j = 0; j < n2; ++j) holder[j] = -1;
    for (int qi = 0; qi < nq; ++qi) {
        accepted[qi] = -1;
        int d1 = INT32_MAX, d2 = INT32_MAX, arg = -1;
        for (uint32_t k = off[qi]; k < off[qi + 1]; ++k) {
            const int j = (int)(ent[k] & 0xFFFFu), d = (int)(ent[k] >> 16);
            if (held[(size_t)j] <= d) continue;          // an earlier query holds it at least as close (:573)
            if (d < d1) { d2 = d1; d1 = d; arg = j; }
            else if (d < d2) d2 = d;
        }
        if (d1 > th || !((float)d1 < (float)d2 * nnratio)) continue;   // :583-585
        holder[arg] = qi;            // the previous holder, if any, loses the feature (:587-592)
        held[(size_t)arg] = d1;
        accepted[qi] = arg;
    }
    return ORBFE_OK;
}
