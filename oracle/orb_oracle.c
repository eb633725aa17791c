/*
 * orb_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See orb_oracle.h.
 *
 * Compile with:  gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math  (unfused IEEE fp32).
 * Every function cites the reference lines (relative to /root/reference) it restates.
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_pattern.inc"

#define ORC_PATCH_SIZE 31      /* ORBextractor.cc:52 */
#define ORC_HALF_PATCH 15      /* ORBextractor.cc:53 */
#define ORC_EDGE 19            /* ORBextractor.cc:54 */
#define ORC_MAX_LEVELS 32

/* cvRound: round-half-to-even (cvtss2si / lrint), SURVEY 9.6 */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* ------------------------------------------------------------------------------------------------
 * E0  constructor tables
 * ---------------------------------------------------------------------------------------------- */
struct orc_extractor {
    int nfeatures, nlevels, ini_th, min_th;
    float scale_factor;
    float scale[ORC_MAX_LEVELS], inv_scale[ORC_MAX_LEVELS], sigma2[ORC_MAX_LEVELS], inv_sigma2[ORC_MAX_LEVELS];
    int feat_per_level[ORC_MAX_LEVELS];
    int umax[ORC_HALF_PATCH + 1];
    int blur_mode;
    /* taps of the last extract */
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    uint8_t *level[ORC_MAX_LEVELS];
    uint8_t *blurred[ORC_MAX_LEVELS];
    orc_cand *cand[ORC_MAX_LEVELS];
    int ncand[ORC_MAX_LEVELS];
    orc_cand *sel[ORC_MAX_LEVELS];
    int nsel[ORC_MAX_LEVELS];
    long blur_ties;
    int tie_breaks;
};

/* ORBextractor.cc:449-465 */
void orc_get_umax(int out[16])
{
    int v, v0;
    int vmax = cv_floor_f(ORC_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceilf(ORC_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORC_HALF_PATCH * ORC_HALF_PATCH;
    for (v = 0; v <= ORC_HALF_PATCH; ++v) out[v] = 0;
    for (v = 0; v <= vmax; ++v) out[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = ORC_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (out[v0] == out[v0 + 1]) ++v0;
        out[v] = v0;
        ++v0;
    }
}

const signed char *orc_get_pattern(void) { return orc_pattern31; }

/* ORBextractor.cc:399-439 */
orc_extractor *orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > ORC_MAX_LEVELS || nfeatures < 0) return NULL;
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->nfeatures = nfeatures;
    e->scale_factor = scale_factor;
    e->nlevels = nlevels;
    e->ini_th = ini_th;
    e->min_th = min_th;
    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->scale[i] = e->scale[i - 1] * scale_factor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    float factor = 1.0f / scale_factor;
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->feat_per_level[l] = cv_round_f(desired);
        sum += e->feat_per_level[l];
        desired *= factor;
    }
    e->feat_per_level[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    orc_get_umax(e->umax);
    return e;
}

static void free_taps(orc_extractor *e)
{
    for (int l = 0; l < ORC_MAX_LEVELS; l++) {
        free(e->level[l]); e->level[l] = NULL;
        free(e->blurred[l]); e->blurred[l] = NULL;
        free(e->cand[l]); e->cand[l] = NULL;
        free(e->sel[l]); e->sel[l] = NULL;
        e->ncand[l] = e->nsel[l] = 0;
    }
}

void orc_destroy(orc_extractor *e)
{
    if (!e) return;
    free_taps(e);
    free(e);
}

int orc_nlevels(const orc_extractor *e) { return e->nlevels; }

void orc_get_scales(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2)
{
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->scale[i];
        if (inv_scale) inv_scale[i] = e->inv_scale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    }
}

void orc_get_features_per_level(const orc_extractor *e, int *out)
{
    for (int i = 0; i < e->nlevels; i++) out[i] = e->feat_per_level[i];
}

/* ORBextractor.cc:1121-1122 */
void orc_level_sizes(const orc_extractor *e, int w, int h, int *lw, int *lh)
{
    for (int l = 0; l < e->nlevels; l++) {
        float s = e->inv_scale[l];
        lw[l] = cv_round_f((float)w * s);
        lh[l] = cv_round_f((float)h * s);
    }
}

/* ORBextractor.cc:780-796.  W = 30 px cells over the [16, dim-16) detection window. */
int orc_cell_grid(int lw, int lh, int *ncols, int *nrows, int *wcell, int *hcell)
{
    const int minb = ORC_EDGE - 3;
    const int maxbx = lw - ORC_EDGE + 3, maxby = lh - ORC_EDGE + 3;
    const float width = (float)(maxbx - minb), height = (float)(maxby - minb);
    const float W = 30;
    if (width < W || height < W) return 0; /* reference: nCols==0 -> division by zero (undefined) */
    const int nc = (int)(width / W), nr = (int)(height / W);
    *ncols = nc;
    *nrows = nr;
    *wcell = (int)ceilf(width / nc);
    *hcell = (int)ceilf(height / nr);
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * E2  cv::resize(INTER_LINEAR) for CV_8UC1, OpenCV 3.2 generic path
 * ---------------------------------------------------------------------------------------------- */
static inline int16_t sat_short(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_resize_tables(int ssize, int dsize, int is_x, int *ofs, int16_t *coef)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short(cv_round_f((1.f - f) * 2048));
        coef[2 * d + 1] = sat_short(cv_round_f(f * 2048));
    }
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride)
{
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    int *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    int16_t *alpha = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)dw);
    int16_t *beta = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)dh);
    int *row0 = (int *)malloc(sizeof(int) * (size_t)dw);
    int *row1 = (int *)malloc(sizeof(int) * (size_t)dw);
    orc_resize_tables(sw, dw, 1, xofs, alpha);
    orc_resize_tables(sh, dh, 0, yofs, beta);
    int prev0 = INT_MIN, prev1 = INT_MIN;
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        if (sy0 < 0) sy0 = 0;
        if (sy0 > sh - 1) sy0 = sh - 1;
        if (sy1 < 0) sy1 = 0;
        if (sy1 > sh - 1) sy1 = sh - 1;
        /* horizontal pass for the two source rows (recomputed unless reusable) */
        if (sy0 == prev1) { int *t = row0; row0 = row1; row1 = t; prev0 = prev1; prev1 = INT_MIN; }
        if (sy0 != prev0) {
            const uint8_t *S = src + (size_t)sy0 * sstride;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                row0[dx] = S[sx] * alpha[2 * dx] + S[sx1] * alpha[2 * dx + 1];
            }
            prev0 = sy0;
        }
        if (sy1 != prev1) {
            const uint8_t *S = src + (size_t)sy1 * sstride;
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
                row1[dx] = S[sx] * alpha[2 * dx] + S[sx1] * alpha[2 * dx + 1];
            }
            prev1 = sy1;
        }
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
}

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ORBextractor.cc:1136-1142 (copyMakeBorder, BORDER_REFLECT_101); dst is (w+2b) x (h+2b) */
void orc_copy_make_border101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                             int border)
{
    for (int y = -border; y < h + border; y++) {
        const uint8_t *S = src + (size_t)reflect101(y, h) * sstride;
        uint8_t *D = dst + (size_t)(y + border) * dstride;
        for (int x = -border; x < w + border; x++) D[x + border] = S[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------------------------------------
 * E3a  cv::FAST (FAST-9 on the radius-3 16-pixel circle), SURVEY 9.3
 * ---------------------------------------------------------------------------------------------- */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* A = max(A_dark, A_bright): A_dark = max over the 16 nine-arcs of min(v - p_k), A_bright likewise
 * with p_k - v.  The pixel is a corner at threshold t  <=>  A > t ; cv score = A - 1. */
static int fast_arc_strength(const uint8_t *p, const int *off)
{
    int d[16], m2[16], m4[16], m8[16];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[off[k]];
    int best = INT_MIN;
    /* dark arcs: sliding-window minimum of d over 9 circular taps */
    for (int k = 0; k < 16; k++) { int a = d[k], b = d[(k + 1) & 15]; m2[k] = a < b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m2[k], b = m2[(k + 2) & 15]; m4[k] = a < b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m4[k], b = m4[(k + 4) & 15]; m8[k] = a < b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m8[k], b = d[(k + 8) & 15]; int m = a < b ? a : b; if (m > best) best = m; }
    /* bright arcs: sliding-window minimum of -d  == -(sliding max of d) */
    for (int k = 0; k < 16; k++) { int a = d[k], b = d[(k + 1) & 15]; m2[k] = a > b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m2[k], b = m2[(k + 2) & 15]; m4[k] = a > b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m4[k], b = m4[(k + 4) & 15]; m8[k] = a > b ? a : b; }
    for (int k = 0; k < 16; k++) { int a = m8[k], b = d[(k + 8) & 15]; int m = a > b ? a : b; if (-m > best) best = -m; }
    return best;
}

void orc_fast_score_map(const uint8_t *img, int w, int h, int stride, uint8_t *score, int score_stride)
{
    int off[16];
    for (int k = 0; k < 16; k++) off[k] = fast_dy[k] * stride + fast_dx[k];
    for (int y = 0; y < h; y++) memset(score + (size_t)y * score_stride, 0, (size_t)w);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int a = fast_arc_strength(img + (size_t)y * stride + x, off) - 1;
            score[(size_t)y * score_stride + x] = (uint8_t)(a < 0 ? 0 : a > 255 ? 255 : a);
        }
}

int orc_fast9(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax, orc_cand *out, int cap)
{
    if (w < 7 || h < 7) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    int off[16];
    for (int k = 0; k < 16; k++) off[k] = fast_dy[k] * stride + fast_dx[k];
    /* score buffer: 0 = not a corner at this threshold (cv keeps three rolling rows; a full map is equivalent) */
    int16_t *sc = (int16_t *)calloc((size_t)w * h, sizeof(int16_t));
    const int t = threshold;
    for (int y = 3; y < h - 3; y++) {
        const uint8_t *row = img + (size_t)y * stride;
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = row + x;
            const int v = p[0];
            /* any 9-arc contains >= 2 of the 4 compass taps: cheap necessary test */
            const int c0 = p[off[0]], c4 = p[off[4]], c8 = p[off[8]], c12 = p[off[12]];
            int nb = (c0 > v + t) + (c4 > v + t) + (c8 > v + t) + (c12 > v + t);
            int nd = (c0 < v - t) + (c4 < v - t) + (c8 < v - t) + (c12 < v - t);
            if (nb < 2 && nd < 2) continue;
            int a = fast_arc_strength(p, off);
            if (a > t) sc[(size_t)y * w + x] = (int16_t)(a - 1 > 0 ? a - 1 : 0) + 1; /* store score+1 so 0 == none */
        }
    }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s1 = sc[(size_t)y * w + x];
            if (!s1) continue;
            int s = s1 - 1;
            if (nonmax) {
                /* strict maximum over the 8 neighbours; non-corners and off-interior count as 0 */
                int ok = 1;
                for (int dy = -1; dy <= 1 && ok; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        int q1 = sc[(size_t)(y + dy) * w + (x + dx)];
                        int q = q1 ? q1 - 1 : 0;
                        if (!(s > q)) { ok = 0; break; }
                    }
                if (!ok) continue;
            }
            if (n >= cap) { free(sc); return -1; }
            out[n].x = (float)x;
            out[n].y = (float)y;
            out[n].response = nonmax ? (float)s : 0.f;
            n++;
        }
    free(sc);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * E4  DistributeOctTree (ORBextractor.cc:540-765) + ExtractorNode::DivideNode (:478-534)
 * std::list<ExtractorNode> is modelled by an index-linked list over a node pool.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int ulx, uly, urx, bry; /* UL.x, UL.y, UR.x, BR.y (BL/BR derive from these) */
    int *keys;              /* indices into the candidate array, original order preserved */
    int nkeys;
    int prev, next;         /* list links; -1 = none */
    int alive;
} qnode;

typedef struct {
    qnode *n;
    int count, cap;
    int head, tail, size;
} qlist;

static int ql_new(qlist *L)
{
    if (L->count == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 256;
        L->n = (qnode *)realloc(L->n, sizeof(qnode) * (size_t)L->cap);
    }
    qnode *q = &L->n[L->count];
    memset(q, 0, sizeof(*q));
    q->prev = q->next = -1;
    return L->count++;
}
static void ql_push_back(qlist *L, int id)
{
    qnode *q = &L->n[id];
    q->prev = L->tail; q->next = -1; q->alive = 1;
    if (L->tail >= 0) L->n[L->tail].next = id; else L->head = id;
    L->tail = id; L->size++;
}
static void ql_push_front(qlist *L, int id)
{
    qnode *q = &L->n[id];
    q->next = L->head; q->prev = -1; q->alive = 1;
    if (L->head >= 0) L->n[L->head].prev = id; else L->tail = id;
    L->head = id; L->size++;
}
static int ql_erase(qlist *L, int id) /* returns next */
{
    qnode *q = &L->n[id];
    int nx = q->next;
    if (q->prev >= 0) L->n[q->prev].next = q->next; else L->head = q->next;
    if (q->next >= 0) L->n[q->next].prev = q->prev; else L->tail = q->prev;
    q->alive = 0; L->size--;
    free(q->keys); q->keys = NULL;
    return nx;
}

/* DivideNode: creates up to 4 children, push_front those with keys in order n1..n4 (:623-662);
 * appends (size, id) of children with >1 keys to (vs, vid). Returns count appended. */
typedef struct { int size, id; } szid;

static int divide_and_push(qlist *L, int id, const orc_cand *c, szid *vs, int *nvs)
{
    /* copy what we need first: ql_new may realloc the pool */
    const int ulx = L->n[id].ulx, uly = L->n[id].uly, urx = L->n[id].urx, bry = L->n[id].bry;
    const int nk = L->n[id].nkeys;
    const int halfx = (int)ceilf((float)(urx - ulx) / 2);
    const int halfy = (int)ceilf((float)(bry - uly) / 2);
    const int midx = ulx + halfx, midy = uly + halfy;
    int *kb[4];
    int kn[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++) kb[i] = (int *)malloc(sizeof(int) * (size_t)(nk > 0 ? nk : 1));
    const int *keys = L->n[id].keys;
    for (int i = 0; i < nk; i++) {
        const orc_cand *kp = &c[keys[i]];
        int q;
        if (kp->x < (float)midx) q = (kp->y < (float)midy) ? 0 : 2;
        else q = (kp->y < (float)midy) ? 1 : 3;
        kb[q][kn[q]++] = keys[i];
    }
    const int bx[4][4] = {/* ulx, uly, urx, bry */
                          {ulx, uly, midx, midy},
                          {midx, uly, urx, midy},
                          {ulx, midy, midx, bry},
                          {midx, midy, urx, bry}};
    int added = 0;
    for (int q = 0; q < 4; q++) {
        if (kn[q] > 0) {
            int cid = ql_new(L);
            qnode *n = &L->n[cid];
            n->ulx = bx[q][0]; n->uly = bx[q][1]; n->urx = bx[q][2]; n->bry = bx[q][3];
            n->keys = kb[q]; n->nkeys = kn[q];
            ql_push_front(L, cid);
            if (kn[q] > 1) { vs[*nvs].size = kn[q]; vs[*nvs].id = cid; (*nvs)++; added++; }
        } else {
            free(kb[q]);
        }
    }
    return added;
}

/* stable ascending sort by size: the documented replacement for the pointer tie-break at :686 */
static void stable_sort_szid(szid *a, int n)
{
    if (n < 2) return;
    szid *tmp = (szid *)malloc(sizeof(szid) * (size_t)n);
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (a[j].size < a[i].size) ? a[j++] : a[i++];
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, sizeof(szid) * (size_t)n);
    }
    free(tmp);
}

int orc_distribute_octtree(const orc_cand *in, int n, int minx, int maxx, int miny, int maxy, int N,
                           orc_cand *out, int cap, orc_octree_stats *st)
{
    orc_octree_stats s = {0, 0, 0};
    if (st) *st = s;
    const int nIni = (int)roundf((float)(maxx - minx) / (float)(maxy - miny)); /* :545 */
    if (nIni < 1) return n == 0 ? 0 : -1; /* reference: division by zero / out-of-range index */
    const float hX = (float)(maxx - minx) / nIni;                              /* :547 */

    qlist L = {NULL, 0, 0, -1, -1, 0};
    int *ini = (int *)malloc(sizeof(int) * (size_t)nIni);
    for (int i = 0; i < nIni; i++) { /* :553-564 */
        int id = ql_new(&L);
        qnode *q = &L.n[id];
        q->ulx = (int)(hX * (float)i);
        q->urx = (int)(hX * (float)(i + 1));
        q->uly = 0;
        q->bry = maxy - miny;
        q->keys = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        q->nkeys = 0;
        ql_push_back(&L, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; i++) { /* :567-571 */
        int r = (int)(in[i].x / hX);
        if (r < 0) r = 0;
        if (r >= nIni) r = nIni - 1; /* unreachable for in-window points; guards UB */
        qnode *q = &L.n[ini[r]];
        q->keys[q->nkeys++] = i;
    }
    free(ini);
    for (int id = L.head; id >= 0;) { /* :573-587 */
        if (L.n[id].nkeys == 0) id = ql_erase(&L, id);
        else id = L.n[id].next;
    }

    int finish = 0;
    szid *vs = (szid *)malloc(sizeof(szid) * (size_t)(4 * (n + nIni) + 16));
    szid *vprev = (szid *)malloc(sizeof(szid) * (size_t)(4 * (n + nIni) + 16));
    int nvs = 0;
    while (!finish) { /* :596-741 */
        s.iterations++;
        int prev_size = L.size;
        int n_to_expand = 0;
        nvs = 0;
        for (int id = L.head; id >= 0;) {
            if (L.n[id].nkeys == 1) { id = L.n[id].next; continue; } /* bNoMore */
            n_to_expand += divide_and_push(&L, id, in, vs, &nvs);
            id = ql_erase(&L, id);
        }
        if (L.size >= N || L.size == prev_size) {
            finish = 1;
        } else if (L.size + n_to_expand * 3 > N) {
            while (!finish) {
                s.phaseb_passes++;
                prev_size = L.size;
                int nprev = nvs;
                memcpy(vprev, vs, sizeof(szid) * (size_t)nprev);
                nvs = 0;
                stable_sort_szid(vprev, nprev);
                for (int j = nprev - 1; j >= 0; j--) {
                    divide_and_push(&L, vprev[j].id, in, vs, &nvs);
                    ql_erase(&L, vprev[j].id);
                    if (L.size >= N) {
                        if (j > 0 && vprev[j - 1].size == vprev[j].size) s.tie_breaks++;
                        break;
                    }
                }
                if (L.size >= N || L.size == prev_size) finish = 1;
            }
        }
    }
    free(vs);
    free(vprev);

    /* :743-762 keep the strongest key per node, list order */
    int nout = 0, rc = 0;
    for (int id = L.head; id >= 0; id = L.n[id].next) {
        const qnode *q = &L.n[id];
        int bi = q->keys[0];
        float best = in[bi].response;
        for (int k = 1; k < q->nkeys; k++)
            if (in[q->keys[k]].response > best) { bi = q->keys[k]; best = in[bi].response; }
        if (nout >= cap) { rc = -2; break; }
        out[nout++] = in[bi];
    }
    for (int i = 0; i < L.count; i++) if (L.n[i].alive) free(L.n[i].keys);
    free(L.n);
    if (st) *st = s;
    return rc ? rc : nout;
}

/* ------------------------------------------------------------------------------------------------
 * E6  IC_Angle + cv::fastAtan2 (OpenCV 3.2 atan_f32), SURVEY 9.5
 * ---------------------------------------------------------------------------------------------- */
float orc_fast_atan2(float y, float x)
{
    const float s = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s;
    const float p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s;
    const float p7 = -0.04432655554792128f * s;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ORBextractor.cc:59-85 */
void orc_ic_moments(const uint8_t *img, int stride, int x, int y, int *m10, int *m01)
{
    int umax[16];
    orc_get_umax(umax);
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -ORC_HALF_PATCH; u <= ORC_HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= ORC_HALF_PATCH; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    *m10 = m_10;
    *m01 = m_01;
}

float orc_ic_angle(const uint8_t *img, int stride, int x, int y)
{
    int m10, m01;
    orc_ic_moments(img, stride, x, y, &m10, &m01);
    return orc_fast_atan2((float)m01, (float)m10); /* :87 */
}

/* ------------------------------------------------------------------------------------------------
 * E7  GaussianBlur(7x7, sigma 2, REFLECT_101), OpenCV <= 3.3 8-bit path, SURVEY 9.4
 * ---------------------------------------------------------------------------------------------- */
static void gauss7_kernel_q8(int k[7])
{
    /* getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256) */
    float cf[7];
    double sum = 0;
    const double scale2x = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        cf[i] = (float)exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cv_round_f(cf[i] * 256.f);
    }
}

void orc_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride, int mode,
                        long *ties)
{
    int k[7];
    gauss7_kernel_q8(k); /* {18,34,49,55,49,34,18}, sum 257 */
    int *R = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *r = R + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = -3; i <= 3; i++) acc += k[i + 3] * S[reflect101(x + i, w)];
            r[x] = acc;
        }
    }
    long nties = 0;
    const int vec_w = w & ~3;
    for (int y = 0; y < h; y++) {
        const int *rr[7];
        for (int j = -3; j <= 3; j++) rr[j + 3] = R + (size_t)reflect101(y + j, h) * w;
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += k[j] * rr[j][x];
            int v = (acc + 32768) >> 16;
            if ((acc & 0xFFFF) == 0x8000) {
                nties++;
                if (mode == 1 && x < vec_w && (v & 1)) v -= 1; /* cvtps2dq: half-to-even */
            }
            D[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    if (ties) *ties = nties;
    free(R);
}

/* ------------------------------------------------------------------------------------------------
 * E8  steered BRIEF (ORBextractor.cc:92-131)
 * ---------------------------------------------------------------------------------------------- */
/* Canonical cos/sin: the reference line :97 resolves to glibc cosf/sinf (faithfully but not always
 * correctly rounded, CPU-ifunc dependent).  Contract: evaluate in fp64 with the FIXED operation
 * sequence below (Cody-Waite reduction by pi/2, fdlibm kernel polynomials, separate mul/add, no
 * FMA) and round once to fp32.  The HIP kernel executes the identical sequence. */
void orc_sincos_rad(float angle, float *cos_a, float *sin_b);
void orc_sincos(float angle_deg, float *cos_a, float *sin_b)
{
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f); /* :91 */
    const float angle = angle_deg * factor_pi;                                   /* :96 */
    orc_sincos_rad(angle, cos_a, sin_b);
}

/* the canonical (float)cos / (float)sin of a float argument in radians (what :97 computes) */
void orc_sincos_rad(float angle, float *cos_a, float *sin_b)
{
    const double x = (double)angle;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
    const double kf = floor(x * two_over_pi + 0.5);
    const int k = (int)kf;
    const double r = (x - kf * pio2_hi) - kf * pio2_lo;
    const double z = r * r;
    /* kernel sin */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S5 + z * S6;
    ps = S4 + z * ps;
    ps = S3 + z * ps;
    ps = S2 + z * ps;
    ps = S1 + z * ps;
    const double sn = r + (z * r) * ps;
    double pc = C5 + z * C6;
    pc = C4 + z * pc;
    pc = C3 + z * pc;
    pc = C2 + z * pc;
    pc = C1 + z * pc;
    const double cs = 1.0 - (0.5 * z - (z * z) * pc);
    double s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *cos_a = (float)c;
    *sin_b = (float)s;
}

void orc_descriptor(const uint8_t *img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    float a, b;
    orc_sincos(angle_deg, &a, &b);
    const uint8_t *center = img + (size_t)y * stride + x;
    const signed char *p = orc_pattern31;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++, p += 4) {
            const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            /* GET_VALUE(idx): center[cvRound(x*b + y*a)*step + cvRound(x*a - y*b)] (:102-104) */
            int t0 = center[cv_round_f(x0 * b + y0 * a) * stride + cv_round_f(x0 * a - y0 * b)];
            int t1 = center[cv_round_f(x1 * b + y1 * a) * stride + cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------------
 * E1  operator()  (ORBextractor.cc:1052-1114), E2 ComputePyramid (:1117-1145),
 *     E3 ComputeKeyPointsOctTree (:771-862)
 * ---------------------------------------------------------------------------------------------- */
void orc_set_blur_mode(orc_extractor *e, int mode) { e->blur_mode = mode; }

int orc_extract(orc_extractor *e, const uint8_t *gray, int w, int h, int stride, orc_keypoint *kps,
                uint8_t *desc, int cap, int *n_out)
{
    if (!e) return -1;
    if (!gray || w == 0 || h == 0) return 0; /* :1055 empty image -> silent return */
    if (w < 0 || h < 0 || stride < w) return -1;
    const int nl = e->nlevels;
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    orc_level_sizes(e, w, h, lw, lh);
    for (int l = 0; l < nl; l++) {
        int a, b, c, d;
        if (!orc_cell_grid(lw[l], lh[l], &a, &b, &c, &d)) return -1;
        if ((int)roundf((float)(lw[l] - 32) / (float)(lh[l] - 32)) < 1) return -1;
    }
    free_taps(e);
    e->blur_ties = 0;
    e->tie_breaks = 0;

    /* ---- ComputePyramid: level 0 = input, level l = resize(level l-1) chained (:1134) ---- */
    for (int l = 0; l < nl; l++) {
        e->lw[l] = lw[l];
        e->lh[l] = lh[l];
        e->level[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
        if (l == 0)
            for (int y = 0; y < h; y++) memcpy(e->level[0] + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
        else
            orc_resize_linear_u8(e->level[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], e->level[l], lw[l], lh[l], lw[l]);
    }

    /* ---- ComputeKeyPointsOctTree ---- */
    int total = 0;
    for (int l = 0; l < nl; l++) {
        const int W = lw[l], H = lh[l];
        const int minbx = ORC_EDGE - 3, minby = minbx;
        const int maxbx = W - ORC_EDGE + 3, maxby = H - ORC_EDGE + 3;
        int ncols, nrows, wcell, hcell;
        orc_cell_grid(W, H, &ncols, &nrows, &wcell, &hcell);
        int ccap = 1024, nc = 0;
        orc_cand *cd = (orc_cand *)malloc(sizeof(orc_cand) * (size_t)ccap);
        orc_cand *cell = (orc_cand *)malloc(sizeof(orc_cand) * 72 * 72); /* tile <= 66x66 (ncols==1) */
        for (int i = 0; i < nrows; i++) {
            const float iniY = (float)(minby + i * hcell);
            float maxY = iniY + hcell + 6;
            if (iniY >= maxby - 3) continue;
            if (maxY > maxby) maxY = (float)maxby;
            for (int j = 0; j < ncols; j++) {
                const float iniX = (float)(minbx + j * wcell);
                float maxX = iniX + wcell + 6;
                if (iniX >= maxbx - 6) continue;
                if (maxX > maxbx) maxX = (float)maxbx;
                const int x0 = (int)iniX, y0 = (int)iniY, tw = (int)maxX - x0, th = (int)maxY - y0;
                const uint8_t *tile = e->level[l] + (size_t)y0 * W + x0;
                int nk;
                if (tw > 72 || th > 72) {
                    free(cd);
                    free(cell);
                    return -1;
                }
                nk = orc_fast9(tile, tw, th, W, e->ini_th, 1, cell, 72 * 72);
                if (nk == 0) nk = orc_fast9(tile, tw, th, W, e->min_th, 1, cell, 72 * 72); /* :821-825 */
                for (int k = 0; k < nk; k++) {
                    if (nc == ccap) { ccap *= 2; cd = (orc_cand *)realloc(cd, sizeof(orc_cand) * (size_t)ccap); }
                    cd[nc].x = cell[k].x + (float)(j * wcell); /* :831-832 */
                    cd[nc].y = cell[k].y + (float)(i * hcell);
                    cd[nc].response = cell[k].response;
                    nc++;
                }
            }
        }
        free(cell);
        e->cand[l] = cd;
        e->ncand[l] = nc;
        const int N = e->feat_per_level[l];
        int scap = nc > 0 ? nc : 1;
        orc_cand *sel = (orc_cand *)malloc(sizeof(orc_cand) * (size_t)scap);
        orc_octree_stats st;
        int ns = orc_distribute_octtree(cd, nc, minbx, maxbx, minby, maxby, N, sel, scap, &st);
        if (ns < 0) { free(sel); return -1; }
        e->tie_breaks += st.tie_breaks;
        for (int k = 0; k < ns; k++) { /* :851-857 (only the border shift; octave/size set on output) */
            sel[k].x += (float)minbx;
            sel[k].y += (float)minby;
        }
        e->sel[l] = sel;
        e->nsel[l] = ns;
        total += ns;
    }
    if (n_out) *n_out = total;
    if (total > cap) return -2;

    /* ---- orientation on the UNBLURRED levels (:860-861), then blur + descriptors per level ---- */
    int ofs = 0;
    for (int l = 0; l < nl; l++) {
        const int ns = e->nsel[l];
        if (ns == 0) continue; /* :1090 */
        const int W = lw[l], H = lh[l];
        e->blurred[l] = (uint8_t *)malloc((size_t)W * H);
        long ties = 0;
        orc_gaussian_blur7(e->level[l], W, H, W, e->blurred[l], W, e->blur_mode, &ties); /* :1094-1095 */
        e->blur_ties += ties;
        const int scaled_patch = (int)(ORC_PATCH_SIZE * e->scale[l]); /* :846 */
        for (int k = 0; k < ns; k++) {
            const orc_cand *c = &e->sel[l][k];
            orc_keypoint *kp = &kps[ofs + k];
            const int xi = cv_round_f(c->x), yi = cv_round_f(c->y);
            kp->angle = orc_ic_angle(e->level[l], W, xi, yi);
            kp->response = c->response;
            kp->octave = l;
            kp->class_id = -1;
            kp->size = (float)scaled_patch;
            orc_descriptor(e->blurred[l], W, xi, yi, kp->angle, desc + (size_t)(ofs + k) * 32);
            kp->x = c->x;
            kp->y = c->y;
            if (l != 0) { /* :1104-1110 */
                kp->x = c->x * e->scale[l];
                kp->y = c->y * e->scale[l];
            }
        }
        ofs += ns;
    }
    return 0;
}

const uint8_t *orc_tap_level(const orc_extractor *e, int level, int *w, int *h, int *stride)
{
    if (w) *w = e->lw[level];
    if (h) *h = e->lh[level];
    if (stride) *stride = e->lw[level];
    return e->level[level];
}
const uint8_t *orc_tap_blurred(const orc_extractor *e, int level, int *w, int *h, int *stride)
{
    if (w) *w = e->lw[level];
    if (h) *h = e->lh[level];
    if (stride) *stride = e->lw[level];
    return e->blurred[level];
}
const orc_cand *orc_tap_candidates(const orc_extractor *e, int level, int *n)
{
    *n = e->ncand[level];
    return e->cand[level];
}
const orc_cand *orc_tap_selected(const orc_extractor *e, int level, int *n)
{
    *n = e->nsel[level];
    return e->sel[level];
}
long orc_tap_blur_ties(const orc_extractor *e) { return e->blur_ties; }
int orc_tap_octree_tie_breaks(const orc_extractor *e) { return e->tie_breaks; }

/* ------------------------------------------------------------------------------------------------
 * Matcher core
 * ---------------------------------------------------------------------------------------------- */
/* ORBmatcher.cc:1968-1984 */
int orc_hamming(const uint8_t a[32], const uint8_t b[32])
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24);
    }
    return dist;
}

/* ORBmatcher.cc:1912-1957 */
void orc_three_maxima(const int *counts, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    int i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = counts[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            i3 = i2; i2 = i1; i1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            i3 = i2; i2 = i;
        } else if (s > max3) {
            max3 = s; i3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
    *ind1 = i1; *ind2 = i2; *ind3 = i3;
}

/* ORBmatcher.cc:308-313 (factor = 1.0f/HISTO_LENGTH, sic) */
int orc_rot_bin(float angle1, float angle2)
{
    const float factor = 1.0f / 30;
    float rot = angle1 - angle2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == 30) bin = 0;
    return bin;
}

/* histogram prune shared by M1/M2/M3 (:338-360): matches[key] = -1 for keys outside the 3 maxima */
static int rot_prune(const int *bin_of_key, const int *keys, int nkeys, int32_t *matches)
{
    int counts[30];
    memset(counts, 0, sizeof(counts));
    for (int i = 0; i < nkeys; i++) counts[bin_of_key[i]]++;
    int i1, i2, i3;
    orc_three_maxima(counts, 30, &i1, &i2, &i3);
    int removed = 0;
    for (int i = 0; i < nkeys; i++) {
        int b = bin_of_key[i];
        if (b == i1 || b == i2 || b == i3) continue;
        matches[keys[i]] = -1;
        removed++;
    }
    return removed;
}

/* M3: per query best/2nd-best over ALL train rows (update idiom of :280-289, initial
 * 256), accept best <= th && (float)best < nnratio*(float)second, then M5 with the query index. */
int orc_match_bf(const uint8_t *q, int nq, const uint8_t *t, int nt, const float *q_angle, const float *t_angle,
                 float nnratio, int th, int check_ori, int32_t *match_q2t, int32_t *best, int32_t *second,
                 int *nmatches)
{
    if (nq < 0 || nt < 0) return -1;
    int *bins = (int *)malloc(sizeof(int) * (size_t)(nq > 0 ? nq : 1));
    int *keys = (int *)malloc(sizeof(int) * (size_t)(nq > 0 ? nq : 1));
    int nk = 0, nm = 0;
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_hamming(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) { b2 = d; }
        }
        if (best) best[i] = b1;
        if (second) second[i] = b2;
        match_q2t[i] = -1;
        if (bi >= 0 && b1 <= th && (float)b1 < nnratio * (float)b2) {
            match_q2t[i] = bi;
            nm++;
            if (check_ori && q_angle && t_angle) {
                bins[nk] = orc_rot_bin(q_angle[i], t_angle[bi]);
                keys[nk] = i;
                nk++;
            }
        }
    }
    if (check_ori && q_angle && t_angle) nm -= rot_prune(bins, keys, nk, match_q2t);
    free(bins);
    free(keys);
    if (nmatches) *nmatches = nm;
    return 0;
}

/* M1 (strict_lt = 0, validF = NULL) and M2 (strict_lt = 1, validF = MapPoint-good flags).
 * matchF2KF[iF] = index of the KF feature whose MapPoint was assigned to F feature iF, -1 = none.
 * For M2 the reference output is indexed by KF1 feature (vpMatches12[idx1] = MP2[bestIdx2]); the
 * same routine covers it with (KF:=KF1, F:=KF2) and the caller inverting the map -- the greedy
 * exclusion (vbMatched2 / vpMapPointMatches) is on the F side in both. */
int orc_search_by_bow(const uint8_t *descKF, int nKF, const uint8_t *validKF, const float *angKF,
                      const uint32_t *nodeKF, const uint32_t *offKF, const uint32_t *idxKF, int nnodesKF,
                      const uint8_t *descF, int nF, const uint8_t *validF, const float *angF,
                      const uint32_t *nodeF, const uint32_t *offF, const uint32_t *idxF, int nnodesF,
                      float nnratio, int th_low, int strict_lt, int check_ori, int32_t *matchF2KF,
                      int *nmatches)
{
    if (nKF < 0 || nF < 0) return -1;
    for (int i = 0; i < nF; i++) matchF2KF[i] = -1;
    int *bins = (int *)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
    int *keys = (int *)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
    int nk = 0, nm = 0;
    int a = 0, b = 0;
    while (a < nnodesKF && b < nnodesF) {
        if (nodeKF[a] == nodeF[b]) {
            for (uint32_t ik = offKF[a]; ik < offKF[a + 1]; ik++) {
                const uint32_t rk = idxKF[ik];
                if ((int)rk >= nKF) { free(bins); free(keys); return -1; }
                if (validKF && !validKF[rk]) continue; /* !pMP || pMP->isBad() (:256-259) */
                int b1 = 256, b2 = 256, bi = -1;
                for (uint32_t jf = offF[b]; jf < offF[b + 1]; jf++) {
                    const uint32_t rf = idxF[jf];
                    if ((int)rf >= nF) { free(bins); free(keys); return -1; }
                    if (matchF2KF[rf] >= 0) continue;        /* :273 / vbMatched2 :725 */
                    if (validF && !validF[rf]) continue;     /* :725-728 */
                    int d = orc_hamming(descKF + (size_t)rk * 32, descF + (size_t)rf * 32);
                    if (d < b1) { b2 = b1; b1 = d; bi = (int)rf; }
                    else if (d < b2) { b2 = d; }
                }
                const int pass = strict_lt ? (b1 < th_low) : (b1 <= th_low); /* :745 vs :292 */
                if (pass && bi >= 0 && (float)b1 < nnratio * (float)b2) {
                    matchF2KF[bi] = (int32_t)rk;
                    if (check_ori) {
                        bins[nk] = orc_rot_bin(angKF[rk], angF[bi]);
                        keys[nk] = bi;
                        nk++;
                    }
                    nm++;
                }
            }
            a++;
            b++;
        } else if (nodeKF[a] < nodeF[b]) {
            while (a < nnodesKF && nodeKF[a] < nodeF[b]) a++; /* lower_bound (:329) */
        } else {
            while (b < nnodesF && nodeF[b] < nodeKF[a]) b++; /* :333 */
        }
    }
    if (check_ori) nm -= rot_prune(bins, keys, nk, matchF2KF);
    free(bins);
    free(keys);
    if (nmatches) *nmatches = nm;
    return 0;
}

/* 8(f).1: per query i, candidates cand[off[i]..off[i+1]) -> best/2nd-best with the :280-289 idiom; second_idx (optional)
 * = the candidate that last set the runner-up, i.e. what ORBmatcher.cc:128-140 tracks as bestLevel2's owner */
int orc_hamming_csr2(const uint8_t *q, int nq, const uint8_t *t, int nt, const uint32_t *off, const uint32_t *cand,
                     int32_t *best_idx, int32_t *best, int32_t *second, int32_t *second_idx)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1, si = -1;
        for (uint32_t j = off[i]; j < off[i + 1]; j++) {
            if ((int)cand[j] >= nt) return -1;
            int d = orc_hamming(q + (size_t)i * 32, t + (size_t)cand[j] * 32);
            if (d < b1) { b2 = b1; si = bi; b1 = d; bi = (int)cand[j]; }
            else if (d < b2) { b2 = d; si = (int)cand[j]; }
        }
        best_idx[i] = bi;
        best[i] = b1;
        second[i] = b2;
        if (second_idx) second_idx[i] = si;
    }
    return 0;
}
int orc_hamming_csr(const uint8_t *q, int nq, const uint8_t *t, int nt, const uint32_t *off,
                    const uint32_t *cand, int32_t *best_idx, int32_t *best, int32_t *second)
{
    return orc_hamming_csr2(q, nq, t, nt, off, cand, best_idx, best, second, 0);
}

/* ------------------------------------------------------------------------------------------------
 * 8(f).2  Frame grid index (src/Frame.cc:319-334, 465-531)
 * ---------------------------------------------------------------------------------------------- */
static int pos_in_grid(float x, float y, float minx, float miny, float gw_inv, float gh_inv, int *px, int *py)
{
    *px = (int)roundf((x - minx) * gw_inv); /* :525 std::round(float) */
    *py = (int)roundf((y - miny) * gh_inv);
    return !(*px < 0 || *px >= ORC_GRID_COLS || *py < 0 || *py >= ORC_GRID_ROWS);
}

int orc_assign_grid(const float *xy, int n, float minx, float miny, float gw_inv, float gh_inv, uint32_t *cell_off,
                    uint32_t *cell_idx)
{
    const int nc = ORC_GRID_COLS * ORC_GRID_ROWS;
    uint32_t *cnt = (uint32_t *)calloc((size_t)nc + 1, sizeof(uint32_t));
    int placed = 0;
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(xy[2 * i], xy[2 * i + 1], minx, miny, gw_inv, gh_inv, &px, &py)) cnt[px * ORC_GRID_ROWS + py]++;
    }
    cell_off[0] = 0;
    for (int c = 0; c < nc; c++) cell_off[c + 1] = cell_off[c] + cnt[c];
    memset(cnt, 0, sizeof(uint32_t) * (size_t)nc);
    for (int i = 0; i < n; i++) { /* push_back in keypoint order (:326-333) */
        int px, py;
        if (pos_in_grid(xy[2 * i], xy[2 * i + 1], minx, miny, gw_inv, gh_inv, &px, &py)) {
            const int c = px * ORC_GRID_ROWS + py;
            cell_idx[cell_off[c] + cnt[c]++] = (uint32_t)i;
            placed++;
        }
    }
    free(cnt);
    return placed;
}

int orc_features_in_area(const float *xy, const int32_t *octave, const uint32_t *cell_off, const uint32_t *cell_idx,
                         float minx, float miny, float gw_inv, float gh_inv, float x, float y, float r, int min_level,
                         int max_level, uint32_t *out, int cap)
{
    int nmin_x = (int)floorf((x - minx - r) * gw_inv); /* :470 */
    if (nmin_x < 0) nmin_x = 0;
    if (nmin_x >= ORC_GRID_COLS) return 0;
    int nmax_x = (int)ceilf((x - minx + r) * gw_inv);
    if (nmax_x > ORC_GRID_COLS - 1) nmax_x = ORC_GRID_COLS - 1;
    if (nmax_x < 0) return 0;
    int nmin_y = (int)floorf((y - miny - r) * gh_inv);
    if (nmin_y < 0) nmin_y = 0;
    if (nmin_y >= ORC_GRID_ROWS) return 0;
    int nmax_y = (int)ceilf((y - miny + r) * gh_inv);
    if (nmax_y > ORC_GRID_ROWS - 1) nmax_y = ORC_GRID_ROWS - 1;
    if (nmax_y < 0) return 0;
    const int check = (min_level > 0) || (max_level >= 0); /* :486 */
    int n = 0;
    for (int ix = nmin_x; ix <= nmax_x; ix++)
        for (int iy = nmin_y; iy <= nmax_y; iy++) {
            const int c = ix * ORC_GRID_ROWS + iy;
            for (uint32_t j = cell_off[c]; j < cell_off[c + 1]; j++) {
                const uint32_t k = cell_idx[j];
                if (check) {
                    if (octave[k] < min_level) continue;
                    if (max_level >= 0 && octave[k] > max_level) continue;
                }
                const float dx = xy[2 * k] - x, dy = xy[2 * k + 1] - y;
                if (fabsf(dx) < r && fabsf(dy) < r) {
                    if (n >= cap) return -1;
                    out[n++] = k;
                }
            }
        }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * M4  projection-gated searches (src/ORBmatcher.cc:63-157, :1578-1724)
 * ---------------------------------------------------------------------------------------------- */
int orc_search_by_projection(const uint8_t *descF, const float *xyF, const int32_t *octF, int nF, const uint32_t *cell_off,
                             const uint32_t *cell_idx, float minx, float miny, float gw_inv, float gh_inv,
                             const float *uRight, const uint8_t *blocked, const orc_proj_query *q, const uint8_t *qdesc,
                             int nq, int th, float nnratio, int ratio_rule, int32_t *match, int32_t *best, int32_t *second)
{
    return orc_search_by_projection_chi2(descF, xyF, octF, nF, cell_off, cell_idx, minx, miny, gw_inv, gh_inv, uRight, blocked, 0, 0, q,
                                         qdesc, nq, th, nnratio, ratio_rule, match, best, second);
}

/* the same loop with Fuse's candidate gate (src/ORBmatcher.cc:1112-1139) on the queries that carry flag 4: a candidate is
 * skipped when its reprojection error e2 * mvInvLevelSigma2[octave] exceeds 7.8 (mvuRight[idx] >= 0, three terms) or 5.99
 * (monocular keypoint, two terms); float arithmetic term by term, the bound compared in double as `float > 7.8` does */
int orc_search_by_projection_chi2(const uint8_t *descF, const float *xyF, const int32_t *octF, int nF, const uint32_t *cell_off,
                                  const uint32_t *cell_idx, float minx, float miny, float gw_inv, float gh_inv,
                                  const float *uRight, const uint8_t *blocked, const float *inv_sigma2, int nlevels,
                                  const orc_proj_query *q, const uint8_t *qdesc, int nq, int th, float nnratio, int ratio_rule,
                                  int32_t *match, int32_t *best, int32_t *second)
{
    if (nF < 0 || nq < 0 || th > 255) return -1; /* th = TH_HIGH / ORBdist (<= 100 in the reference): 256 is "no candidate" */
    uint8_t *taken = (uint8_t *)calloc((size_t)(nF > 0 ? nF : 1), 1); /* slot holds a MapPoint with Observations() > 0 */
    uint32_t *cand = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nF > 0 ? nF : 1));
    if (blocked)
        for (int i = 0; i < nF; i++) taken[i] = blocked[i] ? 1 : 0;
    for (int i = 0; i < nq; i++) {
        const orc_proj_query *Q = &q[i];
        match[i] = -1;
        if (best) best[i] = 256;
        if (second) second[i] = 256;
        const int nc = orc_features_in_area(xyF, octF, cell_off, cell_idx, minx, miny, gw_inv, gh_inv, Q->u, Q->v, Q->r,
                                            Q->min_level, Q->max_level, cand, nF);
        if (nc <= 0) continue; /* :94 / :1635 */
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = (int)cand[k];
            if (taken[idx]) continue; /* :108-110 / :1647-1649 */
            if ((Q->flags & 2) && uRight && uRight[idx] > 0) { /* :114-119 / :1654-1660 */
                const float er = fabsf(Q->ur - uRight[idx]);
                if (er > Q->r) continue;
            }
            if ((Q->flags & 4) && inv_sigma2) {
                const float ex = Q->u - xyF[2 * idx], ey = Q->v - xyF[2 * idx + 1];
                const int lv = octF[idx] < 0 ? 0 : (octF[idx] >= nlevels ? nlevels - 1 : octF[idx]);
                if (uRight && uRight[idx] >= 0) {
                    const float er = Q->ur - uRight[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_sigma2[lv] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_sigma2[lv] > 5.99) continue;
                }
            }
            const int dist = orc_hamming(qdesc + (size_t)i * 32, descF + (size_t)idx * 32);
            if (dist < bestDist) { /* :128-140 */
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = octF[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = octF[idx];
                bestDist2 = dist;
            }
        }
        if (best) best[i] = bestDist;
        if (second) second[i] = bestDist2;
        if (bestDist <= th) { /* :143 / :1673 */
            if (ratio_rule && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match[i] = bestIdx;                   /* F.mvpMapPoints[bestIdx] = pMP */
            taken[bestIdx] = (Q->flags & 1) ? 1 : 0; /* later queries skip the slot iff this point has observations */
        }
    }
    free(taken);
    free(cand);
    return 0;
}

/* r = A (3x3, row-major, row stride sa) * b (3 vector), the stub's float accumulation or OpenCV's double accumulation */
static void mat3_mul_vec(const float *A, int sa, const float b[3], int gemm_double, float r[3])
{
    for (int y = 0; y < 3; y++) {
        if (gemm_double) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)A[y * sa + k] * (double)b[k];
            r[y] = (float)s;
        } else {
            float s = 0;
            for (int k = 0; k < 3; k++) s += A[y * sa + k] * b[k];
            r[y] = s;
        }
    }
}

int orc_proj_queries_last_frame(const float Tcw_cur[16], const float Tcw_last[16], float fx, float fy, float cx, float cy,
                                float mbf, float mb, float minx, float maxx, float miny, float maxy,
                                const float *scale_factors, int nlast, const uint8_t *has_mp, const uint8_t *outlier,
                                const float *world_pos, const int32_t *octave_last, const uint8_t *mp_obs_gt0, float th,
                                int mono, int gemm_double, orc_proj_query *q, uint8_t *valid)
{
    /* :1593-1607: Rcw, tcw, twc = -Rcw.t() * tcw, Rlw, tlw, tlc = Rlw * twc + tlw */
    float Rt_neg[9], tcw[3], twc[3], tlw[3], tlc[3];
    for (int y = 0; y < 3; y++) {
        tcw[y] = Tcw_cur[y * 4 + 3];
        tlw[y] = Tcw_last[y * 4 + 3];
        for (int x = 0; x < 3; x++) Rt_neg[y * 3 + x] = (float)((double)Tcw_cur[x * 4 + y] * -1.0); /* -(Rcw.t()) */
    }
    mat3_mul_vec(Rt_neg, 3, tcw, gemm_double, twc);
    mat3_mul_vec(Tcw_last, 4, twc, gemm_double, tlc);
    for (int y = 0; y < 3; y++) tlc[y] = tlc[y] + tlw[y];
    const int forward = tlc[2] > mb && !mono;   /* :1604 */
    const int backward = -tlc[2] > mb && !mono; /* :1607 */
    for (int i = 0; i < nlast; i++) {
        valid[i] = 0;
        memset(&q[i], 0, sizeof(q[i]));
        if (!has_mp[i] || outlier[i]) continue; /* :1613-1617 */
        float x3Dc[3];
        mat3_mul_vec(Tcw_cur, 4, world_pos + 3 * (size_t)i, gemm_double, x3Dc); /* :1621 Rcw * x3Dw + tcw */
        for (int y = 0; y < 3; y++) x3Dc[y] = x3Dc[y] + tcw[y];
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / (double)x3Dc[2]); /* :1624 */
        if (invzc < 0) continue;
        const float u = fx * xc * invzc + cx;
        const float v = fy * yc * invzc + cy;
        if (u < minx || u > maxx) continue; /* :1629-1632 */
        if (v < miny || v > maxy) continue;
        const int nLastOctave = octave_last[i];
        const float radius = th * scale_factors[nLastOctave]; /* :1642 */
        q[i].u = u;
        q[i].v = v;
        q[i].r = radius;
        if (forward) { q[i].min_level = nLastOctave; q[i].max_level = -1; }          /* :1645 */
        else if (backward) { q[i].min_level = 0; q[i].max_level = nLastOctave; }     /* :1647 */
        else { q[i].min_level = nLastOctave - 1; q[i].max_level = nLastOctave + 1; } /* :1649 */
        q[i].ur = u - mbf * invzc; /* :1656 */
        q[i].flags = (mp_obs_gt0[i] ? 1 : 0) | 2;
        valid[i] = 1;
    }
    return 0;
}

int orc_proj_queries_local_map(const float *scale_factors, int nmp, const uint8_t *in_view, const uint8_t *bad,
                               const int32_t *scale_level, const float *view_cos, const float *proj_xyr,
                               const uint8_t *mp_obs_gt0, float th, orc_proj_query *q, uint8_t *valid)
{
    const int bFactor = th != 1.0; /* :67 */
    for (int i = 0; i < nmp; i++) {
        valid[i] = 0;
        memset(&q[i], 0, sizeof(q[i]));
        if (!in_view[i] || bad[i]) continue; /* :73-77 */
        const int lvl = scale_level[i];
        float r = (double)view_cos[i] > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos :159-165 */
        if (bFactor) r *= th;
        q[i].u = proj_xyr[3 * i];
        q[i].v = proj_xyr[3 * i + 1];
        q[i].r = r * scale_factors[lvl]; /* :91 */
        q[i].min_level = lvl - 1;
        q[i].max_level = lvl;
        q[i].ur = proj_xyr[3 * i + 2];
        q[i].flags = (mp_obs_gt0[i] ? 1 : 0) | 2;
        valid[i] = 1;
    }
    return 0;
}

/* ---- M4: SearchForTriangulation (src/ORBmatcher.cc:827-1012) ---- */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float F[9], float sigma2)
{
    /* :175-196, F12.at<float>(r, c) = F[3r + c] */
    const float a = x1 * F[0] + y1 * F[3] + F[6];
    const float b = x1 * F[1] + y1 * F[4] + F[7];
    const float c = x1 * F[2] + y1 * F[5] + F[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2;
}

int orc_search_for_triangulation(const uint8_t *desc1, const float *xy1, const uint8_t *elig1, const uint8_t *stereo1, int n1,
                                 const uint32_t *node1, const uint32_t *off1, const uint32_t *idx1, int nn1,
                                 const uint8_t *desc2, const float *xy2, const int32_t *oct2, const uint8_t *elig2,
                                 const uint8_t *stereo2, int n2, const uint32_t *node2, const uint32_t *off2,
                                 const uint32_t *idx2, int nn2, const float F12[9], float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int th_low, int32_t *match12)
{
    (void)n2;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) { /* :849-964: merge walk over the two std::maps */
        if (node1[a] == node2[b]) {
            for (uint32_t i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const uint32_t f1 = idx1[i1];
                if (!elig1[f1]) continue; /* :860-868 */
                const int bStereo1 = stereo1[f1];
                int bestDist = th_low, bestIdx2 = -1;
                for (uint32_t i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const uint32_t f2 = idx2[i2];
                    if (!elig2[f2]) continue; /* :881-889 */
                    const int dist = orc_hamming(desc1 + (size_t)f1 * 32, desc2 + (size_t)f2 * 32);
                    if (dist > th_low || dist > bestDist) continue; /* :895 */
                    if (!bStereo1 && !stereo2[f2]) { /* :900-907 */
                        const float distex = ex - xy2[2 * f2], distey = ey - xy2[2 * f2 + 1];
                        if (distex * distex + distey * distey < 100 * scale_factors2[oct2[f2]]) continue;
                    }
                    if (check_dist_epipolar_line(xy1[2 * f1], xy1[2 * f1 + 1], xy2[2 * f2], xy2[2 * f2 + 1], F12, level_sigma2_2[oct2[f2]])) {
                        bestIdx2 = (int)f2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) match12[f1] = bestIdx2;
            }
            a++;
            b++;
        } else if (node1[a] < node2[b]) {
            while (a < nn1 && node1[a] < node2[b]) a++; /* lower_bound */
        } else {
            while (b < nn2 && node2[b] < node1[a]) b++;
        }
    }
    return 0;
}

/* ---- 8(f).4: MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:284-345) ---- */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

int orc_distinctive(const uint8_t *pool, int npool, const uint32_t *off, const uint32_t *idx, int npoints,
                    int32_t *best_idx, int32_t *median)
{
    for (int p = 0; p < npoints; ++p) {
        const int n = (int)(off[p + 1] - off[p]);
        best_idx[p] = -1;
        median[p] = -1;
        if (n <= 0) continue; /* :308-309 */
        const uint32_t *ob = idx + off[p];
        for (int i = 0; i < n; ++i)
            if ((int)ob[i] >= npool) return -1;
        int *dist = (int *)malloc(sizeof(int) * (size_t)n * (size_t)n);
        int *row = (int *)malloc(sizeof(int) * (size_t)n);
        if (!dist || !row) { free(dist); free(row); return -2; }
        for (int i = 0; i < n; ++i) { /* :314-323 */
            dist[i * n + i] = 0;
            for (int j = i + 1; j < n; ++j) {
                const int d = orc_hamming(pool + (size_t)ob[i] * 32, pool + (size_t)ob[j] * 32);
                dist[i * n + j] = d;
                dist[j * n + i] = d;
            }
        }
        int bestm = 0x7fffffff, besti = 0; /* :328-341 */
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) row[j] = dist[i * n + j];
            qsort(row, (size_t)n, sizeof(int), cmp_int);
            const int med = row[(int)(0.5 * (n - 1))];
            if (med < bestm) {
                bestm = med;
                besti = i;
            }
        }
        best_idx[p] = besti;
        median[p] = bestm;
        free(dist);
        free(row);
    }
    return 0;
}

/* ---- 8(f).2b: Frame::ComputeStereoMatches (src/Frame.cc:642-846) ---- */
typedef struct { int d, i; } orc_di;
static int cmp_di(const void *a, const void *b)
{
    const orc_di *x = (const orc_di *)a, *y = (const orc_di *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

int orc_stereo_matches(const orc_extractor *eL, const orc_extractor *eR, const orc_keypoint *kpsL, const uint8_t *descL,
                       int nL, const orc_keypoint *kpsR, const uint8_t *descR, int nR, float mbf, float mb,
                       float *uRight, float *depth, int32_t *sad)
{
    const int thOrbDist = (100 + 50) / 2; /* (TH_HIGH + TH_LOW) / 2, :647 */
    const float minZ = mb, minD = 0.f;
    const float maxD = mbf / minZ;
    orc_di *vd = (orc_di *)malloc(sizeof(orc_di) * (size_t)(nL > 0 ? nL : 1));
    int nvd = 0;
    if (!vd) return -2;
    for (int i = 0; i < nL; ++i) {
        uRight[i] = -1.0f;
        depth[i] = -1.0f;
        if (sad) sad[i] = -1;
    }
    for (int iL = 0; iL < nL; ++iL) {
        const orc_keypoint *kL = &kpsL[iL];
        const int levelL = kL->octave;
        const float vL = kL->y, uL = kL->x;
        const int rowL = (int)vL; /* vRowIndices[vL]: float -> size_t */
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100; /* TH_HIGH */
        int bestIdxR = -1;
        for (int iR = 0; iR < nR; ++iR) { /* candidates of row rowL in push_back order = ascending iR (:665-679) */
            const orc_keypoint *kR = &kpsR[iR];
            const float r = 2.0f * eR->scale[kR->octave];
            const int maxr = (int)ceilf(kR->y + r), minr = (int)floorf(kR->y - r);
            if (rowL < minr || rowL > maxr) continue;
            if (kR->octave < levelL - 1 || kR->octave > levelL + 1) continue;
            const float uR = kR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming(descL + (size_t)iL * 32, descR + (size_t)iR * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestIdxR < 0 || !(bestDist < thOrbDist)) continue;
        /* sub-pixel refinement by SAD over 11 x 11 windows, 11 shifts (:752-827) */
        const float uR0 = kpsR[bestIdxR].x;
        const float sf = eL->inv_scale[levelL];
        const float scaleduL = roundf(kL->x * sf), scaledvL = roundf(kL->y * sf), scaleduR0 = roundf(uR0 * sf);
        const int w = 5, L = 5;
        const uint8_t *imL = eL->level[levelL], *imR = eR->level[levelL];
        const int wl = eL->lw[levelL], wr = eR->lw[levelL];
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)wr) continue;
        const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
        int bestSad = 0x7fffffff, bestinc = 0;
        float vDists[11];
        const int cl = imL[cv * wl + cu];
        for (int inc = -L; inc <= L; ++inc) {
            const int crc = imR[cv * wr + cr + inc];
            int acc = 0; /* exact: |(l - cl) - (r - crc)| summed over 121 pixels */
            for (int dy = -w; dy <= w; ++dy)
                for (int dx = -w; dx <= w; ++dx) {
                    const int a = imL[(cv + dy) * wl + cu + dx] - cl;
                    const int b = imR[(cv + dy) * wr + cr + inc + dx] - crc;
                    acc += a > b ? a - b : b - a;
                }
            const float dist = (float)acc;
            if (dist < (float)bestSad) {
                bestSad = (int)dist;
                bestinc = inc;
            }
            vDists[L + inc] = dist;
        }
        if (bestinc == -L || bestinc == L) continue;
        const float d1 = vDists[L + bestinc - 1], d2 = vDists[L + bestinc], d3 = vDists[L + bestinc + 1];
        const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
        if (deltaR < -1 || deltaR > 1) continue;
        float bestuR = eL->scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
        float disparity = uL - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) {
                disparity = 0.01;
                bestuR = uL - 0.01; /* double arithmetic, then float */
            }
            depth[iL] = mbf / disparity;
            uRight[iL] = bestuR;
            if (sad) sad[iL] = bestSad;
            vd[nvd].d = bestSad;
            vd[nvd].i = iL;
            ++nvd;
        }
    }
    if (nvd > 0) { /* :831-845; with no match the reference indexes an empty vector (undefined): nothing to do */
        qsort(vd, (size_t)nvd, sizeof(orc_di), cmp_di);
        const float median = (float)vd[nvd / 2].d;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nvd - 1; i >= 0; --i) {
            if ((float)vd[i].d < thDist) break;
            uRight[vd[i].i] = -1;
            depth[vd[i].i] = -1;
        }
    }
    free(vd);
    return 0;
}

/* ---- 8(f).3: DBoW2 TemplatedVocabulary::transform ---- */
typedef struct { uint32_t key; int i; } orc_ki;
static int cmp_ki(const void *a, const void *b)
{
    const orc_ki *x = (const orc_ki *)a, *y = (const orc_ki *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

int orc_bow_transform(int nnodes, const uint32_t *child_off, const uint32_t *child_idx, const uint8_t *node_desc,
                      const uint32_t *word_id, const double *weight, int L, int levelsup, const uint8_t *desc, int n,
                      int32_t *f_word, int32_t *f_node, double *f_weight, uint32_t *bow_id, double *bow_val, int *nbow,
                      uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx, int *nfv)
{
    *nbow = 0;
    *nfv = 0;
    fv_off[0] = 0;
    if (nnodes <= 0 || child_off[1] == child_off[0]) { /* empty vocabulary */
        for (int i = 0; i < n; ++i) { f_word[i] = -1; f_node[i] = -1; f_weight[i] = 0.0; }
        return 0;
    }
    const int nid_level = L - levelsup;
    orc_ki *kw = (orc_ki *)malloc(sizeof(orc_ki) * (size_t)(n > 0 ? n : 1));
    orc_ki *kn = (orc_ki *)malloc(sizeof(orc_ki) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    if (!kw || !kn) { free(kw); free(kn); return -2; }
    for (int i = 0; i < n; ++i) {
        uint32_t nid = 0; /* nid_level <= 0 -> root */
        uint32_t fin = 0;
        int level = 0;
        do {
            ++level;
            const uint32_t c0 = child_off[fin], c1 = child_off[fin + 1];
            fin = child_idx[c0];
            int best = orc_hamming(desc + (size_t)i * 32, node_desc + (size_t)fin * 32);
            for (uint32_t c = c0 + 1; c < c1; ++c) {
                const uint32_t id = child_idx[c];
                const int d = orc_hamming(desc + (size_t)i * 32, node_desc + (size_t)id * 32);
                if (d < best) { best = d; fin = id; }
            }
            if (level == nid_level) nid = fin;
        } while (child_off[fin + 1] != child_off[fin]);
        const double w = weight[fin];
        if (w > 0) {
            f_word[i] = (int32_t)word_id[fin];
            f_node[i] = (int32_t)nid;
            f_weight[i] = w;
            kw[m].key = word_id[fin]; kw[m].i = i;
            kn[m].key = nid; kn[m].i = i;
            ++m;
        } else {
            f_word[i] = -1;
            f_node[i] = -1;
            f_weight[i] = 0.0;
        }
    }
    qsort(kw, (size_t)m, sizeof(orc_ki), cmp_ki);
    qsort(kn, (size_t)m, sizeof(orc_ki), cmp_ki);
    int nb = 0;
    for (int j = 0; j < m;) { /* std::map<WordId, double>: += in feature order */
        int e = j;
        double v = 0.0;
        while (e < m && kw[e].key == kw[j].key) { v += f_weight[kw[e].i]; ++e; }
        bow_id[nb] = kw[j].key;
        bow_val[nb] = v;
        ++nb;
        j = e;
    }
    double norm = 0.0; /* BowVector::normalize(L1) */
    for (int j = 0; j < nb; ++j) norm += fabs(bow_val[j]);
    if (norm > 0.0)
        for (int j = 0; j < nb; ++j) bow_val[j] /= norm;
    int nf = 0;
    for (int j = 0; j < m;) {
        int e = j;
        while (e < m && kn[e].key == kn[j].key) { fv_idx[e] = (uint32_t)kn[e].i; ++e; }
        fv_node[nf] = kn[j].key;
        fv_off[nf] = (uint32_t)j;
        ++nf;
        j = e;
    }
    fv_off[nf] = (uint32_t)m;
    *nbow = nb;
    *nfv = nf;
    free(kw);
    free(kn);
    return 0;
}
