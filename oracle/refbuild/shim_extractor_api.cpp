/*
 * shim_extractor_api.cpp -- the PRODUCT's extractor shim (orb_slam2_ssd_semantic_amd/shim/ORBextractor.{h,cc}, built with
 * -DORBFE_WITH_OPENCV, i.e. its real-OpenCV code path, against the cv stub of oracle/refbuild) driven by the REFERENCE's own
 * caller: Frame::ExtractORB (src/Frame.cc:337-343) is cut verbatim by slice.py and compiled against the shim's header, as
 * it would be inside ORB-SLAM2 after the swap described in INTEGRATION.md.  TEST INFRASTRUCTURE (oracle/_ref/libshim_ext.so).
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "ORBextractor.h" /* the SHIM's header: this TU is compiled with -I <repo>/orb_slam2_ssd_semantic_amd/shim first */

namespace ORB_SLAM2
{
/* the members Frame::ExtractORB touches (include/Frame.h:84, 131-145, 150) */
class Frame
{
  public:
    Frame() : mpORBextractorLeft(0), mpORBextractorRight(0) {}
    void ExtractORB(int flag, const cv::Mat &im);
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
};
#include "gen_frame_extract.inc"
} // namespace ORB_SLAM2

using namespace ORB_SLAM2;

extern "C" {
struct shimext_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

void *shimext_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    return new ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
}
void shimext_destroy(void *h) { delete (ORBextractor *)h; }
void shimext_set_blur_rounding(void *h, int mode) { ((ORBextractor *)h)->mnBlurRounding = mode; }
int shimext_get_blur_rounding(void *h) { return ((ORBextractor *)h)->mnBlurRounding; }
void shimext_set_reuse(void *h, int on) { ((ORBextractor *)h)->mbReuseIdenticalInput = on != 0; }
long shimext_reused_calls(void *h) { return ((ORBextractor *)h)->mnReusedCalls; }

/* left = 1: (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors) through Frame::ExtractORB(0, im); else the right
 * extractor / mvKeysRight.  Returns the keypoint count, -2 if cap is too small, -3 if the shim threw. */
int shimext_extract_via_frame(void *h, int left, const uint8_t *gray, int w, int hh, int stride, shimext_kp *kps, uint8_t *desc,
                              int cap, int keep_pyramid)
{
    ORBextractor *e = (ORBextractor *)h;
    e->mbKeepPyramid = keep_pyramid != 0;
    Frame f;
    f.mpORBextractorLeft = f.mpORBextractorRight = e;
    cv::Mat im(hh, w, CV_8UC1, (void *)gray, (size_t)stride);
    try {
        f.ExtractORB(left ? 0 : 1, im);
    } catch (const std::exception &) {
        return -3;
    }
    const std::vector<cv::KeyPoint> &k = left ? f.mvKeys : f.mvKeysRight;
    const cv::Mat &d = left ? f.mDescriptors : f.mDescriptorsRight;
    const int n = (int)k.size();
    if (n > cap) return -2;
    if (n) memcpy(kps, k.data(), sizeof(cv::KeyPoint) * (size_t)n);
    for (int i = 0; i < n; i++) memcpy(desc + (size_t)i * 32, d.ptr(i), 32);
    return n;
}

/* public mvImagePyramid[level] after a keep_pyramid call: the ROI and the (w+38) x (h+38) buffer around it */
int shimext_level(void *h, int level, int with_border, uint8_t *dst, int dst_cap, int *w, int *hh)
{
    ORBextractor *e = (ORBextractor *)h;
    cv::Mat m = e->mvImagePyramid[(size_t)level];
    if (m.empty()) return -1;
    if (with_border) m.adjustROI(19, 19, 19, 19);
    *w = m.cols;
    *hh = m.rows;
    if (m.cols * m.rows > dst_cap) return -2;
    for (int y = 0; y < m.rows; y++) memcpy(dst + (size_t)y * m.cols, m.ptr(y), (size_t)m.cols);
    return 0;
}

void shimext_getters(void *h, int *levels, float *scale_factor, float *scales, float *inv_scales, float *sigma2, float *inv_sigma2)
{
    ORBextractor *e = (ORBextractor *)h;
    *levels = e->GetLevels();
    *scale_factor = e->GetScaleFactor();
    std::vector<float> a = e->GetScaleFactors(), b = e->GetInverseScaleFactors(), c = e->GetScaleSigmaSquares(),
                       d = e->GetInverseScaleSigmaSquares();
    for (int i = 0; i < *levels; i++) {
        scales[i] = a[(size_t)i];
        inv_scales[i] = b[(size_t)i];
        sigma2[i] = c[(size_t)i];
        inv_sigma2[i] = d[(size_t)i];
    }
}
}
