// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h:35-116.
// Same namespace, class name, constructor, operator() signature, getters and public mvImagePyramid, so
// src/Frame.cc / src/Tracking.cc of the reference compile unchanged against it; all work is forwarded to the
// C-ABI in include/orbfe.h (HIP kernels, no CPU path).
#pragma once
#include <vector>

#ifdef ORBFE_WITH_OPENCV
#include <opencv/cv.h>
#else
#include "cv_stub/orbfe_cv_stub.h"
#endif

struct orbfe_handle;

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image.  Mask is ignored (as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Reference semantics by default: every operator() leaves mvImagePyramid[l] valid, as an ROI of the (w+38)x(h+38)
    // REFLECT_101-padded level (src/ORBextractor.cc:1128-1142), so the unchanged stereo Frame::ComputeStereoMatches
    // (src/Frame.cc:649,761-778) reads what it always read.  The pyramid itself lives on the device; keeping the host copy
    // costs one kernel + one 1.2 MB device-to-host copy per 640x480 call.  Mono / RGB-D tracking never reads it:
    // mbKeepPyramid = false after construction skips that (opt-out; SyncImagePyramid() fetches it on demand).
    std::vector<cv::Mat> mvImagePyramid;
    bool mbKeepPyramid = true;
    void SyncImagePyramid();

    // GaussianBlur's column rounding (SURVEY 9.4 ambiguity A).  The shim stands in for a reference BINARY, and every x86-64
    // OpenCV <= 3.3 build runs the SSE2 column kernel (SymmColumnVec_32s8u: fp32 sum, cvtps2dq = round half to EVEN on the
    // columns below width & ~3, half up on the scalar tail): 1 reproduces that, 0 is the generic C++ path's integer formula
    // (half up everywhere; what an ARM / non-SSE2 build computes and what the C-ABI's zero-initialised params select).  They
    // differ on about 13 pixels of a 640x480 frame's 8 blurred levels.  Set before the first operator().
#ifndef ORBFE_SHIM_BLUR_ROUNDING
#define ORBFE_SHIM_BLUR_ROUNDING 1
#endif
    int mnBlurRounding = ORBFE_SHIM_BLUR_ROUNDING;

    // perfect/src/Tracking.cc:685 and :716 construct two Frames from the SAME mImGray with the same extractor (the second with the
    // dynamic-object mask, which operator() ignores): with this flag the second operator() is answered from the first one's
    // results (ORBFE_OPT_REUSE_IDENTICAL_INPUT: a host compare against the staged previous frame, no GPU work, no pyramid copy).
    // On in a -DORBFE_SHIM_PERFECT build (the tree that has that call pattern), off otherwise.  Set before the first operator().
#ifndef ORBFE_SHIM_REUSE_IDENTICAL_INPUT
#ifdef ORBFE_SHIM_PERFECT
#define ORBFE_SHIM_REUSE_IDENTICAL_INPUT 1
#else
#define ORBFE_SHIM_REUSE_IDENTICAL_INPUT 0
#endif
#endif
    bool mbReuseIdenticalInput = ORBFE_SHIM_REUSE_IDENTICAL_INPUT != 0;
    long mnReusedCalls = 0;   // operator() calls answered that way

    int LastStatus() const { return mLastStatus; }  // orbfe_status of the last call (the reference has no error path)
    // the C handle, for calls that keep the pyramid on the device (orbfe_stereo_matches); null before the first operator()
    orbfe_handle *handle() const { return mpHandle; }

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    bool EnsureHandle(int w, int h);
    orbfe_handle *mpHandle = nullptr;
    int mPlanW = 0, mPlanH = 0, mPlanBlur = -1, mLastStatus = 0;
    bool mbPyramidSynced = false;   // mvImagePyramid holds the frame the handle's device pyramid holds
    std::vector<cv::Mat> mvPadded;
    cv::Mat mPadBlock;  // all padded levels of the last frame, one allocation, one device-to-host copy
    ORBextractor(const ORBextractor &);
    ORBextractor &operator=(const ORBextractor &);
};

}  // namespace ORB_SLAM2
