/*
 * cvstub_impl.cpp -- the five OpenCV algorithms behind the cv stub (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * Each routes to oracle/orb_oracle.c's restatement of the OpenCV 3.2 generic path; see the
 * header comment of cvstub/opencv2/core/core.hpp.  Also here: the allocator and cosf/sinf switches that make the
 * two machine-dependent spots of the reference (ORBextractor.cc:686 pointer-ordered sort, :97 glibc cosf/sinf)
 * controllable from the tests.
 */
#include <dlfcn.h>
#include <new>
#include <sys/mman.h>

#include "opencv2/core/core.hpp"
#include "../orb_oracle.h"

namespace cvstub
{
Tap &tap()
{
    static thread_local Tap t = {false, {}, {}, 0, 0, 0, 0, 0, 0}; /* per thread: the stereo Frame constructor extracts on two */
    return t;
}
int blur_mode = 0;
} // namespace cvstub

namespace cv
{

/* imgproc/src/imgwarp.cpp cv::resize -> orc_resize_linear_u8 */
void resize(InputArray _src, OutputArray _dst, Size dsize, double fx, double fy, int interpolation)
{
    Mat src = _src.getMat();
    assert(src.type() == CV_8UC1 && interpolation == INTER_LINEAR && fx == 0 && fy == 0 && dsize.area() > 0);
    (void)fx;
    (void)fy;
    (void)interpolation;
    _dst.create(dsize, src.type()); /* keeps the caller's ROI when it already has this size (:1134) */
    Mat dst = _dst.getMat();
    orc_resize_linear_u8(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
    cvstub::tap().resize_calls++;
}

/* core/src/copy.cpp cv::copyMakeBorder + copyMakeBorder_8u, borderInterpolate(REFLECT_101) */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}
void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType)
{
    Mat src = _src.getMat();
    assert(src.type() == CV_8UC1 && top >= 0 && bottom >= 0 && left >= 0 && right >= 0);
    if (src.isSubmatrix() && (borderType & BORDER_ISOLATED) == 0) {
        /* a ROI without BORDER_ISOLATED borrows the real pixels around it */
        Size whole;
        Point ofs;
        src.locateROI(whole, ofs);
        int dtop = std::min(ofs.y, top), dbottom = std::min(whole.height - src.rows - ofs.y, bottom);
        int dleft = std::min(ofs.x, left), dright = std::min(whole.width - src.cols - ofs.x, right);
        src.adjustROI(dtop, dbottom, dleft, dright);
        top -= dtop;
        left -= dleft;
        bottom -= dbottom;
        right -= dright;
    }
    _dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
    Mat dst = _dst.getMat();
    borderType &= ~BORDER_ISOLATED;
    assert(borderType == BORDER_REFLECT_101);
    cvstub::tap().border_calls++;
    if (top == 0 && left == 0 && bottom == 0 && right == 0) {
        if (src.data != dst.data || (size_t)src.step != (size_t)dst.step) src.copyTo(dst);
        return;
    }
    const int w = src.cols, h = src.rows;
    /* inner rows first (in place when dst's interior IS src, as at :1136), then whole rows above / below */
    for (int y = 0; y < h; y++) {
        uchar *drow = dst.ptr(y + top);
        const uchar *srow = src.ptr(y);
        if (drow + left != srow) memmove(drow + left, srow, (size_t)w);
        for (int x = 0; x < left; x++) drow[x] = drow[left + reflect101(x - left, w)];
        for (int x = 0; x < right; x++) drow[left + w + x] = drow[left + reflect101(w + x, w)];
    }
    for (int y = 0; y < top; y++) memcpy(dst.ptr(y), dst.ptr(top + reflect101(y - top, h)), (size_t)dst.cols);
    for (int y = 0; y < bottom; y++)
        memcpy(dst.ptr(top + h + y), dst.ptr(top + reflect101(h + y, h)), (size_t)dst.cols);
}

/* features2d/src/fast.cpp cv::FAST (TYPE_9_16) -> orc_fast9 */
void FAST(InputArray _img, std::vector<KeyPoint> &keypoints, int threshold, bool nonmax)
{
    Mat img = _img.getMat();
    assert(img.type() == CV_8UC1);
    keypoints.clear();
    std::vector<orc_cand> buf((size_t)std::max(1, img.rows * img.cols));
    int n = orc_fast9(img.data, img.cols, img.rows, (int)img.step, threshold, nonmax ? 1 : 0, buf.data(), (int)buf.size());
    assert(n >= 0);
    keypoints.reserve((size_t)n);
    for (int i = 0; i < n; i++) keypoints.push_back(KeyPoint(buf[i].x, buf[i].y, 7.f, -1, buf[i].response));
    cvstub::Tap &t = cvstub::tap();
    t.fast_calls++;
    if (t.enabled) {
        cvstub::FastCall c;
        c.tile = img.data;
        c.w = img.cols;
        c.h = img.rows;
        c.threshold = threshold;
        c.out = keypoints;
        t.fast.push_back(c);
    }
}

/* imgproc/src/smooth.cpp cv::GaussianBlur -> orc_gaussian_blur7 */
void GaussianBlur(InputArray _src, OutputArray _dst, Size ksize, double sigmaX, double sigmaY, int borderType)
{
    Mat src = _src.getMat();
    assert(src.type() == CV_8UC1 && ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && sigmaY == 2 &&
           (borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    (void)ksize;
    (void)sigmaX;
    (void)sigmaY;
    (void)borderType;
    _dst.create(src.size(), src.type());
    Mat dst = _dst.getMat();
    long ties = 0;
    /* the oracle's blur finishes its row pass before it writes -> in-place use (:1095) is safe */
    orc_gaussian_blur7(src.data, src.cols, src.rows, (int)src.step, dst.data, (int)dst.step, cvstub::blur_mode, &ties);
    cvstub::Tap &t = cvstub::tap();
    t.blur_calls++;
    t.blur_ties += ties;
    if (t.enabled) {
        cvstub::BlurCall c;
        c.w = dst.cols;
        c.h = dst.rows;
        c.out.resize((size_t)dst.cols * dst.rows);
        for (int y = 0; y < dst.rows; y++) memcpy(&c.out[(size_t)y * dst.cols], dst.ptr(y), (size_t)dst.cols);
        t.blur.push_back(c);
    }
}

/* core/src/mathfuncs.cpp cv::fastAtan2 -> orc_fast_atan2 */
float fastAtan2(float y, float x)
{
    cvstub::tap().atan_calls++;
    return orc_fast_atan2(y, x);
}

void KeyPointsFilter::retainBest(std::vector<KeyPoint> &, int)
{
    fprintf(stderr, "cvstub: KeyPointsFilter::retainBest is only reachable from the dead ComputeKeyPointsOld\n");
    abort();
}

} // namespace cv

/* =====================================================================================================
 * Machine-dependent spot 1: ORBextractor.cc:686 sorts pair<int, ExtractorNode*>, i.e. equal-size nodes by heap
 * address.  ref_set_alloc_mode(1) serves every operator new of this library from a bump arena, so addresses grow
 * with creation order (the tie-break the oracle and the HIP kernel declare); mode 0 uses malloc, whose address
 * order depends on the heap's history.  The library is linked with a version script that keeps these symbols
 * local, so nothing outside oracle/_ref is affected.
 * =================================================================================================== */
namespace
{
const size_t ARENA_BYTES = (size_t)8 << 30; /* virtual; pages are touched on demand */
unsigned char *g_arena = 0;
size_t g_arena_used = 0;
int g_alloc_mode = 0;

unsigned char *arena_map()
{
    void *p = mmap(0, ARENA_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) {
        fprintf(stderr, "oracle/_ref: arena mmap failed\n");
        abort();
    }
    return (unsigned char *)p;
}

void *arena_alloc(size_t n)
{
    static unsigned char *mapped = arena_map(); /* thread-safe one-time mapping */
    g_arena = mapped;
    n = (n + 15) & ~(size_t)15;
    /* atomic: the stereo Frame constructor runs two extractors on two threads (src/Frame.cc:121-124); addresses still
     * grow with creation order inside each thread, which is all the :686 tie-break needs */
    const size_t at = __atomic_fetch_add(&g_arena_used, n, __ATOMIC_RELAXED);
    if (at + n > ARENA_BYTES) {
        fprintf(stderr, "oracle/_ref: bump arena exhausted\n");
        abort();
    }
    return g_arena + at;
}
inline bool in_arena(void *p) { return g_arena && (unsigned char *)p >= g_arena && (unsigned char *)p < g_arena + ARENA_BYTES; }
} // namespace

void *operator new(size_t n)
{
    if (g_alloc_mode == 1) return arena_alloc(n ? n : 1);
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new[](size_t n) { return operator new(n); }
void operator delete(void *p) noexcept
{
    if (!p || in_arena(p)) return;
    free(p);
}
void operator delete[](void *p) noexcept { operator delete(p); }
void operator delete(void *p, size_t) noexcept { operator delete(p); }
void operator delete[](void *p, size_t) noexcept { operator delete(p); }

/* =====================================================================================================
 * Machine-dependent spot 2: ORBextractor.cc:97 `cos(angle)` / `sin(angle)` on a float resolve (using namespace
 * std) to std::cos(float) = glibc cosf / sinf (g++ -O2 merges the pair into sincosf).  Mode 0 forwards to glibc;
 * mode 1 substitutes the canonical orc_sincos sequence the oracle and the HIP kernel use.
 * =================================================================================================== */
namespace
{
int g_trig_mode = 0;
typedef float (*f1_t)(float);
typedef void (*sc_t)(float, float *, float *);
f1_t real_cosf = 0, real_sinf = 0;
sc_t real_sincosf = 0;
void resolve_trig()
{
    if (real_cosf) return;
    real_cosf = (f1_t)dlsym(RTLD_NEXT, "cosf");
    real_sinf = (f1_t)dlsym(RTLD_NEXT, "sinf");
    real_sincosf = (sc_t)dlsym(RTLD_NEXT, "sincosf");
    if (!real_cosf || !real_sinf || !real_sincosf) {
        fprintf(stderr, "oracle/_ref: cannot resolve glibc cosf/sinf/sincosf\n");
        abort();
    }
}
/* orc_sincos takes DEGREES and multiplies by factorPI itself; the reference has already done that
 * multiplication when it calls cos/sin, so the canonical sequence is entered after that step (orc_sincos_rad). */
} // namespace

extern "C" {
void orc_sincos_rad(float angle_rad, float *cos_a, float *sin_b); /* oracle/orb_oracle.c */

float cosf(float x)
{
    if (g_trig_mode == 1) {
        float c, s;
        orc_sincos_rad(x, &c, &s);
        return c;
    }
    resolve_trig();
    return real_cosf(x);
}
float sinf(float x)
{
    if (g_trig_mode == 1) {
        float c, s;
        orc_sincos_rad(x, &c, &s);
        return s;
    }
    resolve_trig();
    return real_sinf(x);
}
void sincosf(float x, float *s, float *c)
{
    if (g_trig_mode == 1) {
        orc_sincos_rad(x, c, s);
        return;
    }
    resolve_trig();
    real_sincosf(x, s, c);
}

/* ---- switches (exported) ---- */
void ref_set_alloc_mode(int bump) { g_alloc_mode = bump ? 1 : 0; }
int ref_get_alloc_mode(void) { return g_alloc_mode; }
void ref_arena_reset(void) { g_arena_used = 0; }
size_t ref_arena_used(void) { return g_arena_used; }
void ref_set_trig_mode(int canonical) { g_trig_mode = canonical ? 1 : 0; }
void ref_set_blur_mode(int mode) { cvstub::blur_mode = mode; }
/* what glibc's own cosf / sinf return on this machine (for the "how often does glibc differ" count) */
void ref_glibc_sincosf(float x, float *s, float *c)
{
    resolve_trig();
    *s = real_sinf(x);
    *c = real_cosf(x);
}
}
