/*
 * ref_slices.cpp -- reference function bodies that live in translation units which cannot be compiled whole here
 * (src/Frame.cc, src/KeyFrame.cc, src/MapPoint.cc: calib3d, DBoW2 templates, g2o ...), cut VERBATIM at build time by
 * oracle/refbuild/slice.py into oracle/_ref/gen_*.inc and compiled against the mock classes of ref_mocks.h.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Pins the (f) rows of the oracle to the reference's own code:
 *   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea      src/Frame.cc:319-334, 522-531, 465-518
 *   KeyFrame::GetFeaturesInArea                                      src/KeyFrame.cc:659-698
 *   Frame::ComputeStereoMatches                                      src/Frame.cc:642-846
 *   MapPoint::ComputeDistinctiveDescriptors, PredictScale x2         src/MapPoint.cc:284-345, 448-480
 * and gives the projection family of the compiled ORBmatcher.cc the grid queries it calls.
 */
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "ORBmatcher.h" /* reference header (its MapPoint.h / KeyFrame.h / Frame.h are ref_mocks.h) */

using namespace std;

namespace ORB_SLAM2
{
#include "gen_frame_grid.inc"
#include "gen_keyframe_grid.inc"
#include "gen_mappoint.inc"
} // namespace ORB_SLAM2

using namespace ORB_SLAM2;

extern "C" {

/* Frame::AssignFeaturesToGrid + PosInGrid: cell c = ix*48 + iy, CSR (cell_off[64*48+1], cell_idx[n]) in push_back order */
int ref_assign_grid(const float *xy, int n, float minx, float miny, float gw_inv, float gh_inv, uint32_t *cell_off,
                    uint32_t *cell_idx)
{
    Frame f;
    f.N = n;
    f.mvKeysUn.assign((size_t)n, cv::KeyPoint());
    for (int i = 0; i < n; i++) f.mvKeysUn[(size_t)i].pt = cv::Point2f(xy[2 * i], xy[2 * i + 1]);
    Frame::mnMinX = minx;
    Frame::mnMinY = miny;
    Frame::mfGridElementWidthInv = gw_inv;
    Frame::mfGridElementHeightInv = gh_inv;
    f.AssignFeaturesToGrid();
    uint32_t k = 0;
    for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
        for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
            cell_off[ix * FRAME_GRID_ROWS + iy] = k;
            for (size_t j = 0; j < f.mGrid[ix][iy].size(); j++) cell_idx[k++] = (uint32_t)f.mGrid[ix][iy][j];
        }
    cell_off[FRAME_GRID_COLS * FRAME_GRID_ROWS] = k;
    return (int)k;
}

/* Frame::GetFeaturesInArea on a frame whose grid is given as that CSR; returns the count (-1: cap too small) */
int ref_features_in_area(const float *xy, const int32_t *octave, int n, const uint32_t *cell_off, const uint32_t *cell_idx,
                         float minx, float miny, float gw_inv, float gh_inv, float x, float y, float r, int min_level,
                         int max_level, uint32_t *out, int cap)
{
    Frame f;
    f.N = n;
    f.mvKeysUn.assign((size_t)n, cv::KeyPoint());
    for (int i = 0; i < n; i++) {
        f.mvKeysUn[(size_t)i].pt = cv::Point2f(xy[2 * i], xy[2 * i + 1]);
        f.mvKeysUn[(size_t)i].octave = octave[i];
    }
    Frame::mnMinX = minx;
    Frame::mnMinY = miny;
    Frame::mfGridElementWidthInv = gw_inv;
    Frame::mfGridElementHeightInv = gh_inv;
    for (int c = 0; c < FRAME_GRID_COLS * FRAME_GRID_ROWS; c++)
        for (uint32_t k = cell_off[c]; k < cell_off[c + 1]; k++) f.mGrid[c / FRAME_GRID_ROWS][c % FRAME_GRID_ROWS].push_back(cell_idx[k]);
    const vector<size_t> v = f.GetFeaturesInArea(x, y, r, min_level, max_level);
    if ((int)v.size() > cap) return -1;
    for (size_t i = 0; i < v.size(); i++) out[i] = (uint32_t)v[i];
    return (int)v.size();
}

/* MapPoint::ComputeDistinctiveDescriptors for a batch of map points: point p observes pool[idx[off[p] .. off[p+1])]
 * (one mock KeyFrame per observation, allocated in one array so that the reference's std::map<KeyFrame*, size_t> iterates
 * them in list order).  best_desc[p][32] = the descriptor the reference stores, has[p] = 0 when it returns early. */
int ref_distinctive(const uint8_t *pool, int npool, const uint32_t *off, const uint32_t *idx, int npoints, uint8_t *best_desc,
                    uint8_t *has)
{
    for (int p = 0; p < npoints; p++) {
        const int m = (int)(off[p + 1] - off[p]);
        vector<KeyFrame> kfs((size_t)std::max(m, 1));
        MapPoint mp;
        for (int i = 0; i < m; i++) {
            kfs[(size_t)i].mDescriptors = cv::Mat(npool, 32, CV_8UC1, (void *)pool);
            mp.mObservations[&kfs[(size_t)i]] = idx[off[p] + i];
        }
        mp.ComputeDistinctiveDescriptors();
        has[p] = mp.mDescriptor.empty() ? 0 : 1;
        if (has[p]) memcpy(best_desc + (size_t)p * 32, mp.mDescriptor.ptr(0), 32);
    }
    return 0;
}

#ifndef REF_SLICES_NO_EXTRACTOR /* libshim_ref.so carries the matcher only */
/* Frame::ComputeStereoMatches on the pyramids two reference extractors built in their last operator() calls.
 * The reference reads `mb` before its constructor assigns it (:682); here the caller supplies it. */
void *ref_ext_object(void *h);
struct ref_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
int ref_stereo_matches(void *extL, void *extR, const ref_kp *kpsL, const uint8_t *descL, int nL, const ref_kp *kpsR,
                       const uint8_t *descR, int nR, float mbf, float mb, float *uRight, float *depth)
{
    Frame f;
    f.mpORBextractorLeft = (ORBextractor *)ref_ext_object(extL);
    f.mpORBextractorRight = (ORBextractor *)ref_ext_object(extR);
    f.N = nL;
    f.mvKeys.assign((const cv::KeyPoint *)kpsL, (const cv::KeyPoint *)kpsL + nL);
    f.mvKeysRight.assign((const cv::KeyPoint *)kpsR, (const cv::KeyPoint *)kpsR + nR);
    f.mDescriptors = cv::Mat(nL, 32, CV_8UC1, (void *)descL);
    f.mDescriptorsRight = cv::Mat(nR, 32, CV_8UC1, (void *)descR);
    f.mvScaleFactors = f.mpORBextractorLeft->GetScaleFactors();
    f.mvInvScaleFactors = f.mpORBextractorLeft->GetInverseScaleFactors();
    f.mbf = mbf;
    f.mb = mb;
    f.ComputeStereoMatches();
    for (int i = 0; i < nL; i++) {
        uRight[i] = f.mvuRight[(size_t)i];
        depth[i] = f.mvDepth[(size_t)i];
    }
    return 0;
}

#endif

int ref_predict_scale(float max_distance, float current_dist, float log_scale_factor, int nlevels)
{
    MapPoint mp;
    mp.mfMaxDistance = max_distance;
    Frame f;
    f.mfLogScaleFactor = log_scale_factor;
    f.mnScaleLevels = nlevels;
    return mp.PredictScale(current_dist, &f);
}
}
