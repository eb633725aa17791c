/*
 * ref_extractor_api.cpp -- C entry points around the UNMODIFIED reference class ORB_SLAM2::ORBextractor
 * (/root/reference/src/ORBextractor.cc, compiled as its own translation unit next to this file).
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: lives in oracle/_ref/libref_orb.so, used by tests/ to pin oracle/orb_oracle.c.
 *
 * Nothing here restates the reference: this file only constructs the class, calls its members (the protected
 * ones through a derived class) and copies what they produce into flat buffers.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "ORBextractor.h" /* the reference's own header */

extern "C" {
void ref_set_alloc_mode(int bump);
void ref_arena_reset(void);
}

namespace
{
/* protected members of the reference class, made reachable without touching its source */
struct RefTap : public ORB_SLAM2::ORBextractor {
    RefTap(int n, float s, int l, int ini, int mn) : ORB_SLAM2::ORBextractor(n, s, l, ini, mn) {}
    using ORB_SLAM2::ORBextractor::ComputeKeyPointsOctTree;
    using ORB_SLAM2::ORBextractor::ComputePyramid;
    using ORB_SLAM2::ORBextractor::DistributeOctTree;
    using ORB_SLAM2::ORBextractor::mnFeaturesPerLevel;
    using ORB_SLAM2::ORBextractor::pattern;
    using ORB_SLAM2::ORBextractor::umax;
};

struct Cand {
    float x, y, response;
};

struct RefExt {
    RefTap *ext;
    int nlevels;
    std::vector<std::vector<Cand>> cands; /* per level, vToDistributeKeys (:829-835) rebuilt from the FAST tap */
    std::vector<cvstub::BlurCall> blur;   /* GaussianBlur results in call order */
    long fast_calls, blur_ties;
};

int g_bump = 1;

/* Reference code runs between enter() and leave(): with g_bump every operator new inside comes from a fresh bump
 * arena (addresses grow with creation order).  Anything that must outlive the region is copied after leave(). */
void enter()
{
    if (g_bump) {
        ref_arena_reset();
        ref_set_alloc_mode(1);
    }
}
void leave() { ref_set_alloc_mode(0); }

void drop_tap()
{
    cvstub::Tap &t = cvstub::tap();
    std::vector<cvstub::FastCall>().swap(t.fast); /* storage may sit in the arena: forget it, do not reuse it */
    std::vector<cvstub::BlurCall>().swap(t.blur);
    t.clear();
}

/* vToDistributeKeys of a level = concatenation, in call order, of the non-empty cv::FAST results of that level's
 * tiles, shifted by the tile origin relative to the detection window (ORBextractor.cc:829-835) */
void rebuild_candidates(RefExt *r)
{
    cvstub::Tap &t = cvstub::tap();
    r->cands.assign((size_t)r->nlevels, std::vector<Cand>());
    for (size_t c = 0; c < t.fast.size(); c++) {
        const cvstub::FastCall &fc = t.fast[c];
        for (int l = 0; l < r->nlevels; l++) {
            const cv::Mat &m = r->ext->mvImagePyramid[l];
            const unsigned char *b = m.data, *e = m.data + (size_t)m.step * m.rows;
            if (fc.tile < b || fc.tile >= e) continue;
            const size_t off = (size_t)(fc.tile - b);
            const int y0 = (int)(off / (size_t)m.step), x0 = (int)(off % (size_t)m.step);
            for (size_t k = 0; k < fc.out.size(); k++) {
                Cand cd = {fc.out[k].pt.x + (float)(x0 - 16), fc.out[k].pt.y + (float)(y0 - 16), fc.out[k].response};
                r->cands[(size_t)l].push_back(cd);
            }
            break;
        }
    }
}
} // namespace

extern "C" {

struct ref_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(ref_keypoint) == sizeof(cv::KeyPoint), "layout");

void ref_config_bump(int bump) { g_bump = bump ? 1 : 0; }
/* for other glue files that run reference code which calls operator() itself (frame_stereo_api.cpp) */
void ref_region_enter() { enter(); }
void ref_region_leave() { leave(); }

/* the reference object behind a handle (for ref_slices.cpp: Frame::ComputeStereoMatches reads its mvImagePyramid) */
void *ref_ext_object(void *h) { return static_cast<ORB_SLAM2::ORBextractor *>(((RefExt *)h)->ext); }

void *ref_ext_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    RefExt *r = new RefExt();
    r->ext = new RefTap(nfeatures, scale_factor, nlevels, ini_th, min_th);
    r->nlevels = nlevels;
    r->fast_calls = r->blur_ties = 0;
    return r;
}
void ref_ext_destroy(void *h)
{
    RefExt *r = (RefExt *)h;
    if (!r) return;
    delete r->ext;
    delete r;
}

/* E0: what the reference constructor computed (ORBextractor.cc:399-466) */
void ref_ext_tables(void *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int *feat_per_level,
                    int *umax16, int *pattern1024)
{
    RefExt *r = (RefExt *)h;
    std::vector<float> a = r->ext->GetScaleFactors(), b = r->ext->GetInverseScaleFactors(),
                       c = r->ext->GetScaleSigmaSquares(), d = r->ext->GetInverseScaleSigmaSquares();
    for (int i = 0; i < r->nlevels; i++) {
        scale[i] = a[(size_t)i];
        inv_scale[i] = b[(size_t)i];
        sigma2[i] = c[(size_t)i];
        inv_sigma2[i] = d[(size_t)i];
        feat_per_level[i] = r->ext->mnFeaturesPerLevel[(size_t)i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = r->ext->umax[(size_t)i];
    for (int i = 0; i < 512; i++) {
        pattern1024[2 * i] = r->ext->pattern[(size_t)i].x;
        pattern1024[2 * i + 1] = r->ext->pattern[(size_t)i].y;
    }
}

/* E1: ORBextractor::operator() (:1052-1114).  Returns 0, or -2 when cap is too small (*n_out is still set). */
int ref_ext_extract(void *h, const uint8_t *gray, int w, int hh, int stride, ref_keypoint *kps, uint8_t *desc, int cap,
                    int *n_out)
{
    RefExt *r = (RefExt *)h;
    drop_tap();
    cvstub::tap().enabled = true;
    int n = 0, rc = 0;
    enter();
    {
        cv::Mat image = (w > 0 && hh > 0) ? cv::Mat(hh, w, CV_8UC1, (void *)gray, (size_t)stride) : cv::Mat();
        std::vector<cv::KeyPoint> keys;
        cv::Mat descriptors;
        (*r->ext)(image, cv::Mat(), keys, descriptors);
        n = (int)keys.size();
        if (n > cap) rc = -2;
        else {
            if (n) memcpy(kps, keys.data(), sizeof(cv::KeyPoint) * (size_t)n);
            for (int i = 0; i < n; i++) memcpy(desc + (size_t)i * 32, descriptors.ptr(i), 32);
        }
    }
    leave();
    cvstub::tap().enabled = false;
    if (n_out && !(w == 0 || hh == 0)) *n_out = n; /* empty image: outputs untouched (:1055) */
    rebuild_candidates(r);
    r->blur = cvstub::tap().blur; /* copied with malloc-backed storage */
    r->fast_calls = cvstub::tap().fast_calls;
    r->blur_ties = cvstub::tap().blur_ties;
    drop_tap();
    return rc;
}

/* E2 + E3 + E4 + E5/E6 only: ComputePyramid (:1117-1145) then ComputeKeyPointsOctTree (:771-862); keypoints per
 * level in level coordinates (before the :1103-1110 rescale), counts in n_level[nlevels] */
int ref_ext_keypoints_octtree(void *h, const uint8_t *gray, int w, int hh, int stride, ref_keypoint *kps, int cap,
                              int *n_level)
{
    RefExt *r = (RefExt *)h;
    drop_tap();
    int total = 0, rc = 0;
    enter();
    {
        cv::Mat image(hh, w, CV_8UC1, (void *)gray, (size_t)stride);
        std::vector<std::vector<cv::KeyPoint>> all;
        r->ext->ComputePyramid(image);
        r->ext->ComputeKeyPointsOctTree(all);
        for (int l = 0; l < r->nlevels; l++) {
            n_level[l] = (int)all[(size_t)l].size();
            if (total + n_level[l] > cap) {
                rc = -2;
                break;
            }
            if (n_level[l]) memcpy(kps + total, all[(size_t)l].data(), sizeof(cv::KeyPoint) * (size_t)n_level[l]);
            total += n_level[l];
        }
    }
    leave();
    drop_tap();
    return rc;
}

/* mvImagePyramid[level] of the last call (:1128); with_border = the whole (w+38) x (h+38) buffer */
int ref_ext_level(void *h, int level, int with_border, uint8_t *dst, int dst_cap, int *w, int *hh)
{
    RefExt *r = (RefExt *)h;
    cv::Mat m = r->ext->mvImagePyramid[(size_t)level];
    if (m.empty()) return -1;
    if (with_border) m.adjustROI(19, 19, 19, 19);
    *w = m.cols;
    *hh = m.rows;
    if (m.cols * m.rows > dst_cap) return -2;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

int ref_ext_num_candidates(void *h, int level) { return (int)((RefExt *)h)->cands[(size_t)level].size(); }
void ref_ext_candidates(void *h, int level, float *xyr)
{
    const std::vector<Cand> &c = ((RefExt *)h)->cands[(size_t)level];
    if (!c.empty()) memcpy(xyr, c.data(), sizeof(Cand) * c.size());
}
int ref_ext_num_blur(void *h) { return (int)((RefExt *)h)->blur.size(); }
int ref_ext_blur(void *h, int call, uint8_t *dst, int dst_cap, int *w, int *hh)
{
    const cvstub::BlurCall &b = ((RefExt *)h)->blur[(size_t)call];
    *w = b.w;
    *hh = b.h;
    if (b.w * b.h > dst_cap) return -2;
    memcpy(dst, b.out.data(), b.out.size());
    return 0;
}
long ref_ext_fast_calls(void *h) { return ((RefExt *)h)->fast_calls; }
long ref_ext_blur_ties(void *h) { return ((RefExt *)h)->blur_ties; }

/* E4 alone: ORBextractor::DistributeOctTree (:540-765) on an arbitrary candidate list */
int ref_distribute_octtree(void *h, const float *xyr, int n, int minx, int maxx, int miny, int maxy, int N, int level,
                           float *out_xyr, int cap)
{
    RefExt *r = (RefExt *)h;
    int ns = 0;
    enter();
    {
        std::vector<cv::KeyPoint> in;
        in.reserve((size_t)n);
        for (int i = 0; i < n; i++) in.push_back(cv::KeyPoint(xyr[3 * i], xyr[3 * i + 1], 7.f, -1, xyr[3 * i + 2]));
        std::vector<cv::KeyPoint> out = r->ext->DistributeOctTree(in, minx, maxx, miny, maxy, N, level);
        ns = (int)out.size();
        if (ns <= cap)
            for (int i = 0; i < ns; i++) {
                out_xyr[3 * i] = out[(size_t)i].pt.x;
                out_xyr[3 * i + 1] = out[(size_t)i].pt.y;
                out_xyr[3 * i + 2] = out[(size_t)i].response;
            }
    }
    leave();
    return ns <= cap ? ns : -2;
}
}
