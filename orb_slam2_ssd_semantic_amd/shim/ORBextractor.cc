// ORBextractor.cc -- shim over the C-ABI.  Replaces the reference's src/ORBextractor.cc in libORB_SLAM2.
#include "ORBextractor.h"

#include <assert.h>
#include <stdio.h>
#include <string.h>

#include <stdexcept>
#include <string>

#include "orbfe.h"

namespace ORB_SLAM2 {

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    // scale tables are identical to the reference constructor (src/ORBextractor.cc:404-421); recomputed here so
    // the getters work before the first frame fixes the device plan
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = mvScaleFactor[i - 1] * _scaleFactor;
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
}

ORBextractor::~ORBextractor() { orbfe_destroy(mpHandle); }

bool ORBextractor::EnsureHandle(int w, int h)
{
    if (mpHandle && w <= mPlanW && h <= mPlanH && mPlanBlur == mnBlurRounding) return true;
    orbfe_destroy(mpHandle);
    mpHandle = nullptr;
    orbfe_params p;
    memset(&p, 0, sizeof(p));
    p.nfeatures = nfeatures;
    p.scale_factor = (float)scaleFactor;
    p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST;
    p.min_th_fast = minThFAST;
    p.max_width = w > mPlanW ? w : mPlanW;
    p.max_height = h > mPlanH ? h : mPlanH;
    p.max_batch = 1;
    p.device = -1;
    p.blur_rounding = mnBlurRounding;
    mLastStatus = orbfe_create(&p, &mpHandle);
    if (mLastStatus != ORBFE_OK) {
        // The reference has no error path (its operator() cannot fail); a tracker silently fed with empty frames is
        // worse than a crash, so device failures are loud: the exception is not caught anywhere in ORB-SLAM2.
        throw std::runtime_error(std::string("ORBextractor (orbfe): orbfe_create failed: ") + orbfe_strerror(mLastStatus) + " (" +
                                 orbfe_last_error() + ")");
    }
    mPlanW = p.max_width;
    mPlanH = p.max_height;
    mPlanBlur = mnBlurRounding;
    if (mbReuseIdenticalInput) orbfe_set_option(mpHandle, ORBFE_OPT_REUSE_IDENTICAL_INPUT, 1);
    orbfe_get_scales(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
    orbfe_get_features_per_level(mpHandle, mnFeaturesPerLevel.data());
    return true;
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors)
{
    if (_image.empty()) return;  // src/ORBextractor.cc:1055-1056
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);  // :1059
    if (!EnsureHandle(image.cols, image.rows)) return;
    const int cap = orbfe_keypoint_capacity(mpHandle);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    _keypoints.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0;
    mLastStatus = orbfe_extract(mpHandle, image.ptr<uint8_t>(0), image.cols, image.rows, (int)image.step,
                                reinterpret_cast<orbfe_keypoint *>(_keypoints.data()), desc.ptr<uint8_t>(0), cap, &n);
    if (mLastStatus != ORBFE_OK) {
        _keypoints.clear();
        throw std::runtime_error(std::string("ORBextractor (orbfe): orbfe_extract failed: ") + orbfe_strerror(mLastStatus) + " (" +
                                 orbfe_last_error() + ")");
    }
    _keypoints.resize(n);
    if (n == 0) {
        _descriptors.release();  // :1073-1074
    } else {
        _descriptors.create(n, 32, CV_8U);  // :1077
        cv::Mat out = _descriptors.getMat();
        for (int i = 0; i < n; ++i) memcpy(out.ptr<uint8_t>(i), desc.ptr<uint8_t>(i), 32);
    }
    // a call answered from the previous one's results left the device pyramid as it was: the host copy made then is still it
    const bool reused = orbfe_last_call_reused(mpHandle) != 0;
    if (reused) ++mnReusedCalls;
    if (mbKeepPyramid && !(reused && mbPyramidSynced)) SyncImagePyramid();
    if (!mbKeepPyramid && !reused) mbPyramidSynced = false;
}

void ORBextractor::SyncImagePyramid()
{
    if (!mpHandle) return;
    const int E = 19;  // EDGE_THRESHOLD
    // ONE device kernel frames every level with its 19-px BORDER_REFLECT_101 border and ONE copy brings the blocks over
    // (orbfe_get_pyramid_padded) into a staging block; every level then gets its OWN freshly allocated (w + 38) x (h + 38) matrix,
    // exactly the reference's memory shape -- `Mat temp(wholeSize, ...)` per level per call (:1126), mvImagePyramid[l] its ROI
    // (:1128, :1136-1142): Mat::adjustROI / locateROI on a level behave as on the reference's, and a cv::Mat copy of
    // mvImagePyramid[l] a caller kept stays valid and unchanged after the next operator() (reference-counted, never re-used).
    // Every failure is loud (the reference cannot fail here; a stale pyramid under ComputeStereoMatches would be silent).
    auto fail = [&](const char *what) {
        throw std::runtime_error(std::string("ORBextractor (orbfe): ") + what + " failed: " + orbfe_strerror(mLastStatus) + " (" +
                                 orbfe_last_error() + ")");
    };
    size_t off[16] = {0}, total = 0;
    mLastStatus = orbfe_get_pyramid_padded(mpHandle, 0, NULL, 0, off, &total);
    if (mLastStatus != ORBFE_OK) fail("orbfe_get_pyramid_padded (sizes)");
    mPadBlock.create(1, (int)total, CV_8U);   // staging only: re-used while the frame size stays
    mLastStatus = orbfe_get_pyramid_padded(mpHandle, 0, mPadBlock.ptr<uint8_t>(0), total, NULL, NULL);
    if (mLastStatus != ORBFE_OK) fail("orbfe_get_pyramid_padded");
    mvPadded.resize(nlevels);
    for (int l = 0; l < nlevels; ++l) {
        int w = 0, h = 0;
        mLastStatus = orbfe_get_level_size(mpHandle, l, &w, &h);
        if (mLastStatus != ORBFE_OK) fail("orbfe_get_level_size");
        const int pw = w + 2 * E, ph = h + 2 * E;
        cv::Mat pad(ph, pw, CV_8U);   // a new allocation: tight rows (pitch w + 38), like the staging block's
        memcpy(pad.ptr<uint8_t>(0), mPadBlock.ptr<uint8_t>(0) + off[l], (size_t)pw * ph);
        mvPadded[l] = pad;
#ifdef ORBFE_WITH_OPENCV
        mvImagePyramid[l] = mvPadded[l](cv::Rect(E, E, w, h));  // ROI inside the padded buffer (:1128)
#else
        mvImagePyramid[l] = mvPadded[l].roi(E, E, w, h);
#endif
    }
    mbPyramidSynced = true;
}

}  // namespace ORB_SLAM2
