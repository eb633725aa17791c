/*
 * frame_stereo_api.cpp -- the reference's STEREO Frame constructor (src/Frame.cc:102-168) and the members it calls
 * (ExtractORB :337-343, UndistortKeyPoints :559-590, ComputeImageBounds :593-626; AssignFeaturesToGrid / PosInGrid /
 * ComputeStereoMatches come from ref_slices.cpp), cut VERBATIM at build time by slice.py and compiled against the mock
 * Frame of ref_mocks.h.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Built twice:
 *   -DFS_PREFIX=ref_      into oracle/_ref/libref_orb.so      against the reference's ORBextractor / ORBmatcher
 *   -DFS_PREFIX=shim_st_   into oracle/_ref/libshim_stereo.so  against the PRODUCT's shims (shim/ORBextractor.h first on
 *                         the include path, shim/ORBmatcher_orbfe.cc as the ORBmatcher translation unit): the unchanged
 *                         constructor reads the shim's public mvImagePyramid in ComputeStereoMatches (:649, :761-778)
 *                         with nothing set on the extractor -- the drop-in claim of INTEGRATION.md for stereo.
 * The constructor runs its two ExtractORB calls on two threads, as written (:121-124).
 *
 * The same translation unit carries the callers of the TUM path (BASELINE configs 1-3 are RGB-D): the RGB-D constructor
 * (src/Frame.cc:176-245) with ComputeStereoFromRGBD (:850-874), the monocular constructor (:247-311), and perfect/'s RGB-D
 * constructor with the dynamic-object mask (perfect/src/Frame.cc:328-420), behind FS_NAME(frame_ctor).
 */
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "ORBmatcher.h" /* reference header; MapPoint.h / KeyFrame.h / Frame.h inside it are ref_mocks.h */

using namespace std;

namespace ORB_SLAM2
{
float Frame::s_mb_before_ctor = 0.f;
long unsigned int Frame::nNextId = 0;
bool Frame::mbInitialComputations = true;
#ifdef FS_DEFINE_GRID_STATICS /* libref_orb.so gets these from ref_matcher_api.cpp */
float Frame::mnMinX = 0, Frame::mnMaxX = 640, Frame::mnMinY = 0, Frame::mnMaxY = 480;
float Frame::mfGridElementWidthInv = 0.1f, Frame::mfGridElementHeightInv = 0.1f;
#endif
#include "gen_frame_stereo_ctor.inc"
#include "gen_frame_masked_ctor.inc"
} // namespace ORB_SLAM2

using namespace ORB_SLAM2;

#define FS_CAT2(a, b) a##b
#define FS_CAT(a, b) FS_CAT2(a, b)
#define FS_NAME(n) FS_CAT(FS_PREFIX, n)

extern "C" {
#ifdef FS_EXT_IS_HANDLE
void *ref_ext_object(void *h); /* ref_extractor_api.cpp: the ORBextractor behind a ref_ext_* handle */
void ref_region_enter();      /* the configured allocator (bump arena = creation-order tie-break at ORBextractor.cc:686) */
void ref_region_leave();
#define FS_EXT(h) ((ORBextractor *)ref_ext_object(h))
#define FS_ENTER() ref_region_enter()
#define FS_LEAVE() ref_region_leave()
#else
#define FS_EXT(h) ((ORBextractor *)(h)) /* shimext-style handles are the objects themselves */
void *FS_NAME(ext_create)(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    return new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
}
void FS_NAME(ext_destroy)(void *h) { delete (ORBextractor *)h; }
void FS_NAME(ext_set_blur_rounding)(void *h, int mode) { ((ORBextractor *)h)->mnBlurRounding = mode; }
void FS_NAME(ext_set_reuse)(void *h, int on) { ((ORBextractor *)h)->mbReuseIdenticalInput = on != 0; }
long FS_NAME(ext_reused_calls)(void *h) { return ((ORBextractor *)h)->mnReusedCalls; }
void FS_NAME(ext_set_keep_pyramid)(void *h, int on) { ((ORBextractor *)h)->mbKeepPyramid = on != 0; }
#define FS_ENTER()
#define FS_LEAVE()
#endif

struct fs_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

/* Frame(imLeft, imRight, ts, extL, extR, voc = NULL, K, distCoef = 0, bf, thDepth).  Outputs, caller-sized by cap:
 * mvKeys / mvKeysUn / mDescriptors / mvuRight / mvDepth (N entries), mvKeysRight / mDescriptorsRight (NR), the grid as
 * CSR over cells ix * 48 + iy, scal[8] = {mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, ..HeightInv, mb, fx}.
 * Returns N, -2 if cap is too small, -3 on an exception. */
int FS_NAME(stereo_frame)(void *extL, void *extR, const uint8_t *left, const uint8_t *right, int w, int h, int stride, float fx,
                          float fy, float cx, float cy, float bf, float th_depth, float mb_before, fs_kp *keys, fs_kp *keys_un,
                          uint8_t *desc, float *u_right, float *depth, fs_kp *keys_r, uint8_t *desc_r, int cap, int *n_right,
                          uint32_t *cell_off, uint32_t *cell_idx, float *scal)
{
    cv::Mat imL(h, w, CV_8UC1, (void *)left, (size_t)stride), imR(h, w, CV_8UC1, (void *)right, (size_t)stride);
    cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
    K.at<float>(0, 0) = fx;
    K.at<float>(1, 1) = fy;
    K.at<float>(0, 2) = cx;
    K.at<float>(1, 2) = cy;
    cv::Mat dist = cv::Mat::zeros(4, 1, CV_32F);
    Frame::mbInitialComputations = true; /* "first frame": image bounds, grid constants and intrinsics are computed */
    Frame::s_mb_before_ctor = mb_before;
    FS_ENTER();
    int rc;
    try {
        Frame f(imL, imR, 0.0, FS_EXT(extL), FS_EXT(extR), (ORBVocabulary *)0, K, dist, bf, th_depth);
        const int n = f.N, nr = (int)f.mvKeysRight.size();
        *n_right = nr;
        if (n > cap || nr > cap) throw std::length_error("cap");
        if (n) {
            memcpy(keys, f.mvKeys.data(), sizeof(cv::KeyPoint) * (size_t)n);
            memcpy(keys_un, f.mvKeysUn.data(), sizeof(cv::KeyPoint) * (size_t)n);
        }
        for (int i = 0; i < n; i++) {
            memcpy(desc + (size_t)i * 32, f.mDescriptors.ptr(i), 32);
            u_right[i] = f.mvuRight[(size_t)i];
            depth[i] = f.mvDepth[(size_t)i];
        }
        if (nr) memcpy(keys_r, f.mvKeysRight.data(), sizeof(cv::KeyPoint) * (size_t)nr);
        for (int i = 0; i < nr; i++) memcpy(desc_r + (size_t)i * 32, f.mDescriptorsRight.ptr(i), 32);
        uint32_t k = 0;
        for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
            for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
                cell_off[ix * FRAME_GRID_ROWS + iy] = k;
                for (size_t j = 0; j < f.mGrid[ix][iy].size(); j++) cell_idx[k++] = (uint32_t)f.mGrid[ix][iy][j];
            }
        cell_off[FRAME_GRID_COLS * FRAME_GRID_ROWS] = k;
        scal[0] = Frame::mnMinX; scal[1] = Frame::mnMaxX; scal[2] = Frame::mnMinY; scal[3] = Frame::mnMaxY;
        scal[4] = Frame::mfGridElementWidthInv; scal[5] = Frame::mfGridElementHeightInv; scal[6] = f.mb; scal[7] = f.fx;
        rc = n;
    } catch (const std::length_error &) {
        rc = -2;
    } catch (const std::exception &) {
        rc = -3;
    }
    FS_LEAVE();
    return rc;
}

/* kind 0: Frame(imGray, imDepth, ts, ext, voc, K, distCoef, bf, thDepth)          src/Frame.cc:176-245
 * kind 1: Frame(imGray, ts, ext, voc, K, distCoef, bf, thDepth)                   src/Frame.cc:247-311
 * kind 2: Frame(imGray, imDepth, imMask, ts, ext, voc, K, distCoef, bf, thDepth)  perfect/src/Frame.cc:328-420
 * depth_img: w x h floats (the caller converted it, src/Tracking.cc:355-356), mask: w x h bytes 0 / 1.  Outputs as
 * FS_NAME(stereo_frame); scal[8] = {mnMinX, mnMaxX, mnMinY, mnMaxY, grid width inv, grid height inv, mb, fx}. */
int FS_NAME(frame_ctor)(void *ext, int kind, const uint8_t *gray, const float *depth_img, const uint8_t *mask, int w, int h, int stride,
                        float fx, float fy, float cx, float cy, float bf, float th_depth, fs_kp *keys, fs_kp *keys_un, uint8_t *desc,
                        float *u_right, float *depth, int cap, uint32_t *cell_off, uint32_t *cell_idx, float *scal)
{
    cv::Mat im(h, w, CV_8UC1, (void *)gray, (size_t)stride);
    cv::Mat imD, imM;
    if (depth_img) imD = cv::Mat(h, w, CV_32F, (void *)depth_img, (size_t)w * sizeof(float));
    if (mask) imM = cv::Mat(h, w, CV_8UC1, (void *)mask, (size_t)w);
    cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
    K.at<float>(0, 0) = fx;
    K.at<float>(1, 1) = fy;
    K.at<float>(0, 2) = cx;
    K.at<float>(1, 2) = cy;
    cv::Mat dist = cv::Mat::zeros(4, 1, CV_32F);
    Frame::mbInitialComputations = true;
    FS_ENTER();
    int rc;
    try {
        Frame *pf = 0;
        if (kind == 0) pf = new Frame(im, imD, 0.0, FS_EXT(ext), (ORBVocabulary *)0, K, dist, bf, th_depth);
        else if (kind == 1) pf = new Frame(im, 0.0, FS_EXT(ext), (ORBVocabulary *)0, K, dist, bf, th_depth);
        else pf = new Frame(im, imD, imM, 0.0, FS_EXT(ext), (ORBVocabulary *)0, K, dist, bf, th_depth);
        Frame &f = *pf;
        const int n = f.N;
        if (n > cap) { delete pf; throw std::length_error("cap"); }
        if ((int)f.mvKeys.size() != n || (n && ((int)f.mvKeysUn.size() != n || f.mDescriptors.rows != n || (int)f.mvuRight.size() != n ||
                                                (int)f.mvDepth.size() != n))) { delete pf; throw std::logic_error("sizes"); }
        if (n) {
            memcpy(keys, f.mvKeys.data(), sizeof(cv::KeyPoint) * (size_t)n);
            memcpy(keys_un, f.mvKeysUn.data(), sizeof(cv::KeyPoint) * (size_t)n);
        }
        for (int i = 0; i < n; i++) {
            memcpy(desc + (size_t)i * 32, f.mDescriptors.ptr(i), 32);
            u_right[i] = f.mvuRight[(size_t)i];
            depth[i] = f.mvDepth[(size_t)i];
        }
        uint32_t k = 0;
        for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
            for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
                cell_off[ix * FRAME_GRID_ROWS + iy] = k;
                for (size_t j = 0; j < f.mGrid[ix][iy].size(); j++) cell_idx[k++] = (uint32_t)f.mGrid[ix][iy][j];
            }
        cell_off[FRAME_GRID_COLS * FRAME_GRID_ROWS] = k;
        scal[0] = Frame::mnMinX; scal[1] = Frame::mnMaxX; scal[2] = Frame::mnMinY; scal[3] = Frame::mnMaxY;
        scal[4] = Frame::mfGridElementWidthInv; scal[5] = Frame::mfGridElementHeightInv; scal[6] = f.mb; scal[7] = f.fx;
        rc = n;
        delete pf;
    } catch (const std::length_error &) {
        rc = -2;
    } catch (const std::exception &) {
        rc = -3;
    }
    FS_LEAVE();
    return rc;
}
}
