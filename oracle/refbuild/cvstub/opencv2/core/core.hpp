/*
 * cv stub for oracle/_ref  (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Purpose: let g++ compile the UNMODIFIED reference translation units
 *     /root/reference/src/ORBextractor.cc  (+ include/ORBextractor.h)
 *     /root/reference/src/ORBmatcher.cc    (+ include/ORBmatcher.h)
 * in a container that has no OpenCV.  The stub supplies the OpenCV *types and containers* those files use
 * (Mat with reference counting / ROI views / create() that keeps a matching buffer, Point_, Size_, Rect_,
 * KeyPoint, _InputArray / _OutputArray, cvRound / cvFloor / cvCeil) with OpenCV 2.4 / 3.2 semantics, and routes
 * the FIVE OpenCV *algorithms* the extractor calls to oracle/orb_oracle.c's restatements of them:
 *
 *     cv::resize (INTER_LINEAR, 8UC1)          -> orc_resize_linear_u8      
 *     cv::copyMakeBorder (REFLECT_101)         -> written here               (pure index reflection)
 *     cv::FAST (threshold, nonmax)             -> orc_fast9                  
 *     cv::GaussianBlur (7x7, sigma 2)          -> orc_gaussian_blur7         
 *     cv::fastAtan2                            -> orc_fast_atan2             
 *
 * Those five stay "restated from the published OpenCV 3.2 algorithm"; everything else that runs in oracle/_ref is
 * the reference's own code.  Every stub call can be recorded (cvstub::tap) so that the test-suite can compare the
 * oracle with the reference stage by stage.
 */
#ifndef ORBFE_CVSTUB_CORE_HPP
#define ORBFE_CVSTUB_CORE_HPP

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <stdexcept>
#include <vector>

#define CV_PI 3.1415926535897932384626433832795
#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)
#define CV_MAT_CN(flags) ((((flags) >> CV_CN_SHIFT) & 511) + 1)

/* cvRound: round half to even (cvtsd2si / lrint), OpenCV core/fast_math.hpp */
static inline int cvRound(double value) { return (int)lrint(value); }
static inline int cvRound(float value) { return (int)lrintf(value); }
static inline int cvRound(int value) { return value; }
static inline int cvFloor(double value)
{
    int i = (int)value;
    return i - (i > value);
}
static inline int cvCeil(double value)
{
    int i = (int)value;
    return i + (i < value);
}

/* OpenCV declares these at GLOBAL scope (core/hal/interface.h); perfect/src/Frame.cc:366 writes at<uchar> inside ORB_SLAM2 */
typedef unsigned char uchar;
typedef unsigned short ushort;

namespace cv
{
using ::uchar;
using ::ushort;

template <typename T> static inline T saturate_cast(float v) { return (T)v; }
template <> inline int saturate_cast<int>(float v) { return cvRound(v); }

/* ---------------------------------------------------------------- small value types (core/types.hpp) */
template <typename T> class Point_
{
  public:
    typedef T value_type;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    template <typename U> Point_(const Point_<U> &p) : x((T)p.x), y((T)p.y) {}
    T dot(const Point_ &p) const { return x * p.x + y * p.y; }
    T x, y;
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point2i Point;

template <typename T> static inline Point_<T> &operator+=(Point_<T> &a, const Point_<T> &b)
{
    a.x += b.x;
    a.y += b.y;
    return a;
}
template <typename T> static inline Point_<T> &operator-=(Point_<T> &a, const Point_<T> &b)
{
    a.x -= b.x;
    a.y -= b.y;
    return a;
}
/* core/types.hpp: a.x = saturate_cast<_Tp>(a.x * b); for float this is a plain fp32 product */
static inline Point2f &operator*=(Point2f &a, float b)
{
    a.x = a.x * b;
    a.y = a.y * b;
    return a;
}
static inline Point2f &operator*=(Point2f &a, double b)
{
    a.x = (float)(a.x * b);
    a.y = (float)(a.y * b);
    return a;
}
static inline Point2f &operator*=(Point2f &a, int b)
{
    a.x = a.x * (float)b;
    a.y = a.y * (float)b;
    return a;
}
template <typename T> static inline Point_<T> operator+(const Point_<T> &a, const Point_<T> &b)
{
    return Point_<T>(a.x + b.x, a.y + b.y);
}
template <typename T> static inline Point_<T> operator-(const Point_<T> &a, const Point_<T> &b)
{
    return Point_<T>(a.x - b.x, a.y - b.y);
}
template <typename T> static inline bool operator==(const Point_<T> &a, const Point_<T> &b)
{
    return a.x == b.x && a.y == b.y;
}

template <typename T> class Point3_
{
  public:
    Point3_() : x(0), y(0), z(0) {}
    Point3_(T _x, T _y, T _z) : x(_x), y(_y), z(_z) {}
    T x, y, z;
};
typedef Point3_<float> Point3f;

template <typename T> class Size_
{
  public:
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    T area() const { return width * height; }
    T width, height;
};
typedef Size_<int> Size2i;
typedef Size2i Size;
template <typename T> static inline bool operator==(const Size_<T> &a, const Size_<T> &b)
{
    return a.width == b.width && a.height == b.height;
}

template <typename T> class Rect_
{
  public:
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T _x, T _y, T _w, T _h) : x(_x), y(_y), width(_w), height(_h) {}
    T x, y, width, height;
};
typedef Rect_<int> Rect;

class Range
{
  public:
    Range() : start(0), end(0) {}
    Range(int s, int e) : start(s), end(e) {}
    int start, end;
};

/* features2d: cv::KeyPoint (28 bytes) */
class KeyPoint
{
  public:
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(Point2f _pt, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(_pt), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id)
    {
    }
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0,
             int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id)
    {
    }
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

/* ---------------------------------------------------------------- cv::Mat (core/mat.hpp), 2-D only */
struct MatStep {
    MatStep() : v(0) {}
    MatStep(size_t s) : v(s) {}
    operator size_t() const { return v; }
    MatStep &operator=(size_t s)
    {
        v = s;
        return *this;
    }
    size_t v;
};

class Mat;
class MatExpr;

/* allocation block shared by all views of one matrix (plays UMatData / refcount) */
struct MatBlock {
    int refcount;
    uchar *base; /* owned, malloc'ed */
    int rows, cols; /* whole size of the allocation, for locateROI */
    size_t step;
};

class Mat
{
  public:
    Mat() : flags(0), dims(0), rows(0), cols(0), data(0), step(), blk(0), ext_rows(0), ext_cols(0), ext_base(0) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size sz, int type) : Mat() { create(sz.height, sz.width, type); }
    /* user-allocated data (no copy, no ownership) */
    Mat(int r, int c, int type, void *d, size_t s = 0) : Mat()
    {
        flags = type;
        dims = 2;
        rows = r;
        cols = c;
        data = (uchar *)d;
        step = s ? s : (size_t)c * elemSize();
        ext_rows = r;
        ext_cols = c;
        ext_base = data;
    }
    Mat(const Mat &m)
        : flags(m.flags), dims(m.dims), rows(m.rows), cols(m.cols), data(m.data), step(m.step), blk(m.blk),
          ext_rows(m.ext_rows), ext_cols(m.ext_cols), ext_base(m.ext_base)
    {
        if (blk) blk->refcount++;
    }
    Mat(const Mat &m, const Rect &roi) : Mat(m)
    {
        assert(0 <= roi.x && 0 <= roi.width && roi.x + roi.width <= m.cols && 0 <= roi.y && 0 <= roi.height &&
               roi.y + roi.height <= m.rows);
        data += (size_t)roi.y * step.v + (size_t)roi.x * elemSize();
        rows = roi.height;
        cols = roi.width;
    }
    Mat(const MatExpr &e);
    ~Mat() { release(); }
    Mat &operator=(const Mat &m)
    {
        if (this != &m) {
            if (m.blk) m.blk->refcount++;
            release();
            flags = m.flags;
            dims = m.dims;
            rows = m.rows;
            cols = m.cols;
            data = m.data;
            step = m.step;
            blk = m.blk;
            ext_rows = m.ext_rows;
            ext_cols = m.ext_cols;
            ext_base = m.ext_base;
        }
        return *this;
    }
    /* Mat = MatExpr: the expression is evaluated INTO this matrix; create() keeps the buffer when size and type
     * already match (core/matop.cpp MatOp_Initializer::assign) -- computeDescriptors relies on that (:1046) */
    Mat &operator=(const MatExpr &e);

    void release()
    {
        if (blk && --blk->refcount == 0) {
            free(blk->base);
            delete_block(blk);
        }
        blk = 0;
        data = 0;
        rows = cols = 0;
        ext_base = 0;
    }
    /* Mat::create: no-op when the matrix already has this size and type (core/matrix.cpp) */
    void create(int r, int c, int type)
    {
        type &= 0xFFF;
        if (dims == 2 && rows == r && cols == c && this->type() == type && data) return;
        release();
        flags = type;
        dims = 2;
        rows = r;
        cols = c;
        step = (size_t)c * elemSize();
        if ((size_t)r * c > 0) {
            blk = new_block();
            blk->refcount = 1;
            blk->rows = r;
            blk->cols = c;
            blk->step = step.v;
            blk->base = (uchar *)malloc(step.v * (size_t)r + 64);
            data = blk->base;
        }
    }
    void create(Size sz, int type) { create(sz.height, sz.width, type); }

    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat &m) const
    {
        m.create(rows, cols, type());
        if (m.data == data && m.step.v == step.v) return;
        for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step.v, data + (size_t)y * step.v, (size_t)cols * elemSize());
    }
    Mat &setTo(double v)
    {
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols * channels(); x++) set_elem(y, x, v);
        return *this;
    }

    /* Mat::reshape(cn): same data, another channel count (continuous matrices only; Frame::UndistortKeyPoints :577-579) */
    Mat reshape(int cn, int new_rows = 0) const
    {
        assert(isContinuous() && (cols * channels()) % cn == 0);
        Mat m(*this);
        m.cols = cols * channels() / cn;
        m.flags = CV_MAKETYPE(depth(), cn);
        if (new_rows > 0 && new_rows != rows) { /* core/matrix.cpp Mat::reshape: total elements kept, rows changed */
            const size_t total = (size_t)rows * m.cols;
            assert(total % (size_t)new_rows == 0);
            m.rows = new_rows;
            m.cols = (int)(total / (size_t)new_rows);
            m.step = (size_t)m.cols * m.elemSize();
        }
        return m;
    }
    /* Mat::push_back(const Mat&) (core/matrix.cpp): an empty matrix becomes a copy of m, otherwise m's rows are appended
     * (same type and width); perfect/src/Frame.cc:372 collects the descriptor rows of the unmasked keypoints with it */
    void push_back(const Mat &m)
    {
        if (m.rows == 0) return;
        if (!data || rows == 0) {
            Mat c;
            m.copyTo(c);
            *this = c;
            return;
        }
        assert(m.type() == type() && m.cols == cols);
        Mat grown(rows + m.rows, cols, type());
        for (int y = 0; y < rows; y++) memcpy(grown.data + (size_t)y * grown.step.v, data + (size_t)y * step.v, (size_t)cols * elemSize());
        for (int y = 0; y < m.rows; y++)
            memcpy(grown.data + (size_t)(rows + y) * grown.step.v, m.data + (size_t)y * m.step.v, (size_t)cols * elemSize());
        *this = grown;
    }
    Mat rowRange(int startrow, int endrow) const { return Mat(*this, Rect(0, startrow, cols, endrow - startrow)); }
    Mat colRange(int startcol, int endcol) const { return Mat(*this, Rect(startcol, 0, endcol - startcol, rows)); }
    Mat row(int y) const { return Mat(*this, Rect(0, y, cols, 1)); }
    Mat col(int x) const { return Mat(*this, Rect(x, 0, 1, rows)); }
    Mat operator()(const Rect &roi) const { return Mat(*this, roi); }
    Mat operator()(Range rr, Range cr) const
    {
        return Mat(*this, Rect(cr.start, rr.start, cr.end - cr.start, rr.end - rr.start));
    }

    int type() const { return flags & 0xFFF; }
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return CV_MAT_CN(flags); }
    size_t elemSize1() const
    {
        static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 0};
        return (size_t)sz[depth()];
    }
    size_t elemSize() const { return elemSize1() * channels(); }
    size_t step1() const { return step.v / elemSize1(); }
    bool empty() const { return data == 0 || rows * cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return rows <= 1 || step.v == (size_t)cols * elemSize(); }
    bool isSubmatrix() const { return blk ? (rows != blk->rows || cols != blk->cols) : (rows != ext_rows || cols != ext_cols); }
    /* core/matrix.cpp Mat::locateROI */
    void locateROI(Size &whole, Point &ofs) const
    {
        const uchar *base = blk ? blk->base : ext_base;
        size_t st = step.v;
        ptrdiff_t d = data - base;
        ofs.y = st ? (int)(d / (ptrdiff_t)st) : 0;
        ofs.x = (int)((d - (ptrdiff_t)st * ofs.y) / (ptrdiff_t)elemSize());
        whole.height = blk ? blk->rows : ext_rows;
        whole.width = blk ? blk->cols : ext_cols;
    }
    Mat &adjustROI(int dtop, int dbottom, int dleft, int dright)
    {
        Size whole;
        Point ofs;
        locateROI(whole, ofs);
        int row1 = std::max(ofs.y - dtop, 0), row2 = std::min(ofs.y + rows + dbottom, whole.height);
        int col1 = std::max(ofs.x - dleft, 0), col2 = std::min(ofs.x + cols + dright, whole.width);
        data += (ptrdiff_t)(row1 - ofs.y) * (ptrdiff_t)step.v + (ptrdiff_t)(col1 - ofs.x) * (ptrdiff_t)elemSize();
        rows = row2 - row1;
        cols = col2 - col1;
        return *this;
    }

    uchar *ptr(int y = 0) { return data + (size_t)y * step.v; }
    const uchar *ptr(int y = 0) const { return data + (size_t)y * step.v; }
    template <typename T> T *ptr(int y = 0) { return (T *)(data + (size_t)y * step.v); }
    template <typename T> const T *ptr(int y = 0) const { return (const T *)(data + (size_t)y * step.v); }
    template <typename T> T &at(int y, int x) { return ((T *)(data + (size_t)y * step.v))[x]; }
    template <typename T> const T &at(int y, int x) const { return ((const T *)(data + (size_t)y * step.v))[x]; }
    /* single-index access on a vector-shaped matrix (core/mat.inl.hpp at(int i0)) */
    template <typename T> T &at(int i)
    {
        if (rows == 1) return ((T *)data)[i];
        if (cols == 1) return *(T *)(data + (size_t)i * step.v);
        return ((T *)(data + (size_t)(i / cols) * step.v))[i % cols];
    }
    template <typename T> const T &at(int i) const { return const_cast<Mat *>(this)->at<T>(i); }

    static MatExpr zeros(int rows, int cols, int type);
    static MatExpr zeros(Size size, int type);
    static MatExpr eye(int rows, int cols, int type);
    static MatExpr ones(int rows, int cols, int type);
    /* core/convert.cpp Mat::convertTo, only CV_8U / CV_32F -> CV_32F (alpha 1, beta 0); in-place use allocates a new
     * buffer because the type changes, as OpenCV does */
    void convertTo(Mat &dst, int rtype) const;

    /* float linear algebra used by the projection matchers of ORBmatcher.cc (eager; CV_32F only) */
    Mat t() const;
    double dot(const Mat &m) const;

    int flags, dims, rows, cols;
    uchar *data;
    MatStep step;

  private:
    static MatBlock *new_block() { return (MatBlock *)malloc(sizeof(MatBlock)); }
    static void delete_block(MatBlock *b) { free(b); }
    void set_elem(int y, int x, double v)
    {
        uchar *p = data + (size_t)y * step.v;
        switch (depth()) {
        case CV_8U: p[x] = (uchar)v; break;
        case CV_32S: ((int *)p)[x] = (int)v; break;
        case CV_32F: ((float *)p)[x] = (float)v; break;
        case CV_64F: ((double *)p)[x] = v; break;
        default: assert(!"cvstub: depth"); }
    }
    MatBlock *blk;
    int ext_rows, ext_cols;
    uchar *ext_base;
};

/* only the initializer expressions the reference uses */
class MatExpr
{
  public:
    enum { ZEROS = 0, EYE = 1, ONES = 2 };
    MatExpr(int k, int r, int c, int t) : kind(k), rows(r), cols(c), type(t) {}
    operator Mat() const
    {
        Mat m;
        assign(m);
        return m;
    }
    void assign(Mat &m) const
    {
        m.create(rows, cols, type);
        m.setTo(kind == ONES ? 1 : 0);
        if (kind == EYE)
            for (int i = 0; i < std::min(rows, cols); i++) {
                if (m.depth() == CV_32F) m.at<float>(i, i) = 1.f;
                else if (m.depth() == CV_64F) m.at<double>(i, i) = 1.0;
                else m.at<uchar>(i, i) = 1;
            }
    }
    int kind, rows, cols, type;
};
inline Mat::Mat(const MatExpr &e) : Mat() { e.assign(*this); }
inline Mat &Mat::operator=(const MatExpr &e)
{
    e.assign(*this);
    return *this;
}
inline MatExpr Mat::zeros(int r, int c, int t) { return MatExpr(MatExpr::ZEROS, r, c, t); }
inline MatExpr Mat::zeros(Size s, int t) { return MatExpr(MatExpr::ZEROS, s.height, s.width, t); }
inline MatExpr Mat::eye(int r, int c, int t) { return MatExpr(MatExpr::EYE, r, c, t); }
inline MatExpr Mat::ones(int r, int c, int t) { return MatExpr(MatExpr::ONES, r, c, t); }
inline void Mat::convertTo(Mat &dst, int rtype) const
{
    assert(CV_MAT_DEPTH(rtype) == CV_32F && channels() == 1 && (depth() == CV_8U || depth() == CV_32F));
    Mat out(rows, cols, CV_32F);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) out.at<float>(y, x) = depth() == CV_8U ? (float)at<uchar>(y, x) : at<float>(y, x);
    dst = out;
}

/* ---- eager CV_32F algebra (only what ORBmatcher.cc's projection family needs; not on any tested path that
 *      claims bit-parity with OpenCV's gemm) ---- */
namespace stubdetail
{
inline Mat newf(int r, int c) { return Mat(r, c, CV_32F); }

inline float gf(const Mat &m, int y, int x) { return m.at<float>(y, x); }
} // namespace stubdetail

/* cv::Scalar and cv::sum (core/stat.cpp: per-channel sums as doubles); perfect/src/Frame.cc:360 sums the 0 / 1 dynamic mask */
template <typename T> class Scalar_
{
  public:
    Scalar_() { val[0] = val[1] = val[2] = val[3] = 0; }
    Scalar_(T v0, T v1 = 0, T v2 = 0, T v3 = 0) { val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
    T val[4];
    const T &operator[](int i) const { return val[i]; }
    T &operator[](int i) { return val[i]; }
};
typedef Scalar_<double> Scalar;
inline Scalar sum(const Mat &m)
{
    Scalar s;
    const int cn = m.channels();
    assert(cn <= 4);
    for (int y = 0; y < m.rows; y++)
        for (int x = 0; x < m.cols; x++)
            for (int c = 0; c < cn; c++) {
                double v = 0;
                switch (m.depth()) {
                case CV_8U: v = m.ptr<uchar>(y)[x * cn + c]; break;
                case CV_32F: v = m.ptr<float>(y)[x * cn + c]; break;
                case CV_32S: v = m.ptr<int>(y)[x * cn + c]; break;
                default: assert(!"cv stub: sum() of this depth");
                }
                s.val[c] += v;
            }
    return s;
}
inline Mat Mat::t() const
{
    Mat r = stubdetail::newf(cols, rows);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) r.at<float>(x, y) = at<float>(y, x);
    return r;
}
inline double Mat::dot(const Mat &m) const
{
    double s = 0;
    assert(total() == m.total());
    const int n = (int)total();
    for (int i = 0; i < n; i++) s += (double)at<float>(i) * (double)m.at<float>(i);
    return s;
}
inline Mat operator*(const Mat &a, const Mat &b)
{
    assert(a.cols == b.rows);
    Mat r = stubdetail::newf(a.rows, b.cols);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < b.cols; x++) {
            float s = 0;
            for (int k = 0; k < a.cols; k++) s += a.at<float>(y, k) * b.at<float>(k, x);
            r.at<float>(y, x) = s;
        }
    return r;
}
inline Mat operator*(const Mat &a, double s)
{
    Mat r = stubdetail::newf(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++) r.at<float>(y, x) = (float)(a.at<float>(y, x) * s);
    return r;
}
inline Mat operator*(double s, const Mat &a) { return a * s; }
inline Mat operator/(const Mat &a, double s)
{
    Mat r = stubdetail::newf(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++) r.at<float>(y, x) = (float)(a.at<float>(y, x) / s);
    return r;
}
inline Mat operator+(const Mat &a, const Mat &b)
{
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat r = stubdetail::newf(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++) r.at<float>(y, x) = a.at<float>(y, x) + b.at<float>(y, x);
    return r;
}
inline Mat operator-(const Mat &a, const Mat &b)
{
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat r = stubdetail::newf(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++) r.at<float>(y, x) = a.at<float>(y, x) - b.at<float>(y, x);
    return r;
}
inline Mat operator-(const Mat &a) { return a * -1.0; }
inline double norm(const Mat &a) { return std::sqrt(a.dot(a)); }
enum { NORM_INF = 1, NORM_L1 = 2, NORM_L2 = 4 };
/* core/stat.cpp cv::norm(src1, src2, NORM_L1) for CV_32F: sum of |a - b| accumulated in double */
inline double norm(const Mat &a, const Mat &b, int normType)
{
    assert(normType == NORM_L1 && a.rows == b.rows && a.cols == b.cols && a.depth() == CV_32F && b.depth() == CV_32F);
    (void)normType;
    double s = 0;
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++) s += std::abs((double)(a.at<float>(y, x) - b.at<float>(y, x)));
    return s;
}

/* ---------------------------------------------------------------- InputArray / OutputArray proxies */
class _InputArray
{
  public:
    _InputArray() : m(0) {}
    _InputArray(const Mat &mat) : m(const_cast<Mat *>(&mat)) {}
    Mat getMat() const { return m ? *m : Mat(); }
    bool empty() const { return !m || m->empty(); }
    Mat *m;
};
class _OutputArray : public _InputArray
{
  public:
    _OutputArray() {}
    _OutputArray(Mat &mat) : _InputArray(mat) {}
    void create(int rows, int cols, int type) const { m->create(rows, cols, type); }
    void create(Size sz, int type) const { m->create(sz.height, sz.width, type); }
    void release() const { m->release(); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
inline InputArray noArray()
{
    static _InputArray none;
    return none;
}

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
       BORDER_REFLECT101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };

/* ---------------------------------------------------------------- the five algorithms (cvstub_impl.cpp) */
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0,
                  int borderType = BORDER_DEFAULT);
float fastAtan2(float y, float x);

/* calib3d: referenced by Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:578, :608) only when the
 * distortion coefficients are non-zero; rectified input (every stereo configuration the reference ships) never calls it */
inline void undistortPoints(const Mat &, Mat &, const Mat &, const Mat &, const Mat &, const Mat &)
{
    throw std::logic_error("cv stub: undistortPoints (calib3d) is not available");
}

/* referenced only by the dead ComputeKeyPointsOld (ORBextractor.cc:1015,1033), which operator() never calls */
class KeyPointsFilter
{
  public:
    static void retainBest(std::vector<KeyPoint> &keypoints, int npoints);
};

} // namespace cv

/* ---------------------------------------------------------------- call recorder for the stage-by-stage tests */
namespace cvstub
{
struct FastCall {
    const unsigned char *tile; /* data pointer of the ROI passed in */
    int w, h, threshold;
    std::vector<cv::KeyPoint> out;
};
struct BlurCall {
    int w, h;
    std::vector<unsigned char> out; /* dense w*h copy of the result */
};
struct Tap {
    bool enabled;
    std::vector<FastCall> fast;
    std::vector<BlurCall> blur;
    long fast_calls, resize_calls, blur_calls, border_calls, atan_calls;
    long blur_ties;
    void clear()
    {
        fast.clear();
        blur.clear();
        fast_calls = resize_calls = blur_calls = border_calls = atan_calls = 0;
        blur_ties = 0;
    }
};
Tap &tap();
extern int blur_mode; /* orc_gaussian_blur7 mode: 0 integer half-up, 1 SSE2 column kernel emulation */
} // namespace cvstub

#endif
