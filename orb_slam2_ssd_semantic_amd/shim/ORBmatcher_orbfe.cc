// ORBmatcher_orbfe.cc -- link-level drop-in for the reference's ORBmatcher (src/ORBmatcher.cc).
//
// The reference's OWN include/ORBmatcher.h stays as it is, so Tracking.cc, LocalMapping.cc and LoopClosing.cc compile
// unchanged.  This file supplies EVERY public member of the class over the C-ABI of liborbfe.so:
//     static int DescriptorDistance(const cv::Mat&, const cv::Mat&)                                 src/ORBmatcher.cc:1968-1984
//     int SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&)                                   :217-363
//     int SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&)                                :665-812
//     int SearchByProjection(Frame&, const std::vector<MapPoint*>&, float)                          :63-157
//     int SearchByProjection(Frame&, const Frame&, float, bool)                                     :1578-1724
//       (with -DORBFE_SHIM_PERFECT, for the perfect/ tree) the overload that also returns the 2-D point pairs,
//                                                                                                    perfect/src/ORBmatcher.cc:1727-1911
//     int SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, float, int)                  :1757-1867
//     int SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>&, int)  :378-470
//     int SearchForInitialization(Frame&, Frame&, vector<cv::Point2f>&, vector<int>&, int)          :523-651
//     int SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool)  :827-1012
//     int SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>&, s12, R12, t12, th)                 :1334-1548
//     int Fuse(KeyFrame*, const std::vector<MapPoint*>&, float)                                     :1031-1182
//     int Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float, vector<MapPoint*>&)         :1198-1299
// and, with -DORBFE_SHIM_STANDALONE, the rest of the translation unit (constants, constructor, RadiusByViewingCos,
// CheckDistEpipolarLine, ComputeThreeMaxima), so that src/ORBmatcher.cc can leave the build altogether.
//
// One pattern throughout: the pose projection and its gates run here, on cv::Mat, statement for statement as in the
// reference; the candidate scans (GetFeaturesInArea + Hamming + whatever rule couples the queries) are ONE device call per
// invocation; the assignments / Replace / AddMapPoint decisions and the rotation histogram are replayed from its result, in
// the reference's order, on the live objects.  The device call is orbfe_search_by_projection(_chi2) for the four
// SearchByProjection members (it also resolves the "slot already taken by an earlier query" rule) and for Fuse x2 /
// SearchBySim3 (independent queries on a KeyFrame's grid; Fuse's reprojection-error gate is ORBFE_PROJ_CHI2_GATE),
// orbfe_search_for_triangulation, orbfe_search_by_bow, and orbfe_window_distances for SearchForInitialization (whose rule
// needs every distance of every window; only the in-order acceptance runs here).
// Integration: add this file to the ORB_SLAM2 library in place of src/ORBmatcher.cc (-DORBFE_SHIM_STANDALONE), or next to it
// with the bodies listed above deleted (or #if 0), and link liborbfe.so.  The class gets no new data member:
// the device matcher handle is per thread, which is also what the C-ABI asks for (Tracking, LocalMapping and LoopClosing call
// the matcher concurrently).
//
// Built and tested in this repo against the reference's unmodified header with mock KeyFrame / Frame / MapPoint types
// (oracle/refbuild: libshim_ref.so / libshim_full.so; tests/test_shim_ref.py and tests/test_projection.py compare it with the
// compiled reference bodies).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <ORBmatcher.h>  // the reference's own header, found on the include path (NOT the stand-alone template form next to this file)
#include "orbfe.h"

namespace
{
struct ThreadMatcher {
    orbfe_matcher *m = nullptr;
    ~ThreadMatcher() { orbfe_matcher_destroy(m); }
    orbfe_matcher *get()
    {
        // the reference's matcher cannot fail; a device failure must not look like "no matches": it throws
        if (!m && orbfe_matcher_create(-1, &m) != ORBFE_OK) {
            m = nullptr;
            throw std::runtime_error(std::string("ORBmatcher (orbfe): orbfe_matcher_create failed: ") + orbfe_last_error());
        }
        return m;
    }
};
thread_local ThreadMatcher t_matcher;

struct Csr {
    std::vector<uint32_t> node, off, idx;
};
// DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned int>>: ascending node ids, features in insertion order
void Flatten(const DBoW2::FeatureVector &fv, Csr &c)
{
    c.off.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        c.node.push_back((uint32_t)it->first);
        for (size_t k = 0; k < it->second.size(); ++k) c.idx.push_back((uint32_t)it->second[k]);
        c.off.push_back((uint32_t)c.idx.size());
    }
}
void Angles(const std::vector<cv::KeyPoint> &k, std::vector<float> &a)
{
    a.resize(k.size());
    for (size_t i = 0; i < k.size(); ++i) a[i] = k[i].angle;
}
void Valid(const std::vector<ORB_SLAM2::MapPoint *> &mp, std::vector<uint8_t> &v)
{
    v.resize(mp.size());
    for (size_t i = 0; i < mp.size(); ++i) v[i] = mp[i] && !mp[i]->isBad();  // :256-259
}
// descriptor rows as one dense N x 32 block (cv::Mat rows of mDescriptors are contiguous in the reference; a ROI is copied)
const uint8_t *Rows(const cv::Mat &d, std::vector<uint8_t> &tmp)
{
    if (d.rows == 0) return nullptr;
    if (d.isContinuous()) return d.ptr<uint8_t>(0);
    tmp.resize((size_t)d.rows * 32);
    for (int i = 0; i < d.rows; ++i) memcpy(&tmp[(size_t)i * 32], d.ptr<uint8_t>(i), 32);
    return tmp.data();
}
// The frame side of a projection search as flat arrays: mDescriptors, mvKeysUn (pt, octave), mvuRight, the slots that hold a
// MapPoint with Observations() > 0 (:108-110 / :1647-1649) and Frame::mGrid in the (ix, iy) order GetFeaturesInArea walks it.
struct FrameSide {
    std::vector<float> xy;
    std::vector<int32_t> oct;
    std::vector<uint8_t> blocked, tmp;
    std::vector<uint32_t> cell_off, cell_idx;
    const uint8_t *desc = nullptr;
    explicit FrameSide(ORB_SLAM2::Frame &F)
    {
        const size_t n = (size_t)F.N;
        xy.resize(2 * n);
        oct.resize(n);
        blocked.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            xy[2 * i] = F.mvKeysUn[i].pt.x;
            xy[2 * i + 1] = F.mvKeysUn[i].pt.y;
            oct[i] = F.mvKeysUn[i].octave;
            ORB_SLAM2::MapPoint *p = F.mvpMapPoints[i];
            blocked[i] = p && p->Observations() > 0;
        }
        cell_off.reserve(ORBFE_GRID_COLS * ORBFE_GRID_ROWS + 1);
        cell_off.push_back(0);
        for (int ix = 0; ix < ORBFE_GRID_COLS; ++ix)
            for (int iy = 0; iy < ORBFE_GRID_ROWS; ++iy) {
                const std::vector<std::size_t> &c = F.mGrid[ix][iy];
                for (size_t k = 0; k < c.size(); ++k) cell_idx.push_back((uint32_t)c[k]);
                cell_off.push_back((uint32_t)cell_idx.size());
            }
        desc = Rows(F.mDescriptors, tmp);
    }
    // a KeyFrame as the searched side (:378-470): slots taken = vpMatched[idx] != NULL (none for Fuse / SearchBySim3)
    FrameSide(ORB_SLAM2::KeyFrame *pKF, const std::vector<ORB_SLAM2::MapPoint *> *vpMatched)
    {
        const size_t n = pKF->mvKeysUn.size();
        xy.resize(2 * n);
        oct.resize(n);
        blocked.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            xy[2 * i] = pKF->mvKeysUn[i].pt.x;
            xy[2 * i + 1] = pKF->mvKeysUn[i].pt.y;
            oct[i] = pKF->mvKeysUn[i].octave;
            blocked[i] = vpMatched && (*vpMatched)[i] != NULL;
        }
        // KeyFrame::mGrid is PROTECTED in the reference (include/KeyFrame.h:223) and this helper is no friend of the class.
        // The grid is a pure function of public data: it is the Frame's, built by Frame::AssignFeaturesToGrid from mvKeysUn
        // with the Frame statics mnMinX / mnMinY (float; KeyFrame keeps truncated int copies) and the cell sizes the
        // KeyFrame copied; per cell in ascending keypoint order.  orbfe_assign_grid_host restates exactly that.
        cell_off.resize(ORBFE_GRID_COLS * ORBFE_GRID_ROWS + 1);
        cell_idx.resize(n);
        if (orbfe_assign_grid_host(xy.data(), (int32_t)n, ORB_SLAM2::Frame::mnMinX, ORB_SLAM2::Frame::mnMinY, pKF->mfGridElementWidthInv,
                                   pKF->mfGridElementHeightInv, cell_off.data(), cell_idx.data(), NULL) != ORBFE_OK)
            throw std::runtime_error(std::string("ORBmatcher (orbfe): orbfe_assign_grid_host: ") + orbfe_last_error());
        desc = Rows(pKF->mDescriptors, tmp);
    }
};

struct Queries {
    std::vector<orbfe_proj_query> q;
    std::vector<uint8_t> desc;
    std::vector<ORB_SLAM2::MapPoint *> mp;
    std::vector<int> src;  // index of the query in the caller's list
    // flags < 0: the Frame overloads' rule (right-image gate on, the slot is taken iff the point has observations)
    void Add(ORB_SLAM2::MapPoint *p, int from, float u, float v, float r, int minLevel, int maxLevel, float ur, int flags = -1)
    {
        orbfe_proj_query e;
        e.u = u; e.v = v; e.r = r; e.min_level = minLevel; e.max_level = maxLevel; e.ur = ur;
        e.flags = flags >= 0 ? flags : (ORBFE_PROJ_RIGHT_GATE | (p->Observations() > 0 ? ORBFE_PROJ_CLAIMS : 0));
        e.pad = 0;
        q.push_back(e);
        const cv::Mat d = p->GetDescriptor();
        desc.insert(desc.end(), d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + 32);
        mp.push_back(p);
        src.push_back(from);
    }
};

void RunSearch(ORB_SLAM2::Frame &F, const Queries &qs, int th, float nnratio, int ratio_rule, std::vector<int32_t> &match,
               bool any_point_blocks = false)
{
    match.assign(qs.q.size(), -1);
    if (qs.q.empty() || F.N == 0) return;
    FrameSide fs(F);
    if (any_point_blocks)  // :1812 `if(CurrentFrame.mvpMapPoints[i2]) continue;`
        for (int i = 0; i < F.N; ++i) fs.blocked[(size_t)i] = F.mvpMapPoints[(size_t)i] != NULL;
    const orbfe_status s = orbfe_search_by_projection(
        t_matcher.get(), fs.desc, fs.xy.data(), fs.oct.data(), F.N, fs.cell_off.data(), fs.cell_idx.data(), ORB_SLAM2::Frame::mnMinX,
        ORB_SLAM2::Frame::mnMinY, ORB_SLAM2::Frame::mfGridElementWidthInv, ORB_SLAM2::Frame::mfGridElementHeightInv,
        F.mvuRight.empty() ? NULL : F.mvuRight.data(), fs.blocked.data(), qs.q.data(), qs.desc.data(), (int32_t)qs.q.size(), th, nnratio,
        ratio_rule, match.data(), NULL, NULL);
    if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByProjection (orbfe): ") + orbfe_last_error());
}
// Independent queries on a KeyFrame's grid (Fuse x2, SearchBySim3): KeyFrame::GetFeaturesInArea, the level window, the
// optional reprojection-error gate and the Hamming scan in one device call; match[k] = the best candidate (first in the
// reference's order on ties) if its distance is <= th, else -1.  No query looks at another one's result.
void RunKeyFrameSearch(ORB_SLAM2::KeyFrame *pKF, const Queries &qs, int th, bool chi2_gate, std::vector<int32_t> &match, const char *who)
{
    match.assign(qs.q.size(), -1);
    if (qs.q.empty() || pKF->mvKeysUn.empty()) return;
    FrameSide fs(pKF, NULL);
    const orbfe_status s = orbfe_search_by_projection_chi2(
        t_matcher.get(), fs.desc, fs.xy.data(), fs.oct.data(), (int32_t)pKF->mvKeysUn.size(), fs.cell_off.data(), fs.cell_idx.data(),
        (float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv,
        chi2_gate ? pKF->mvuRight.data() : NULL, NULL, chi2_gate ? pKF->mvInvLevelSigma2.data() : NULL,
        chi2_gate ? (int32_t)pKF->mvInvLevelSigma2.size() : 0, qs.q.data(), qs.desc.data(), (int32_t)qs.q.size(), th, 0.f, 0, match.data(),
        NULL, NULL);
    if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::") + who + " (orbfe): " + orbfe_last_error());
}
}  // namespace

namespace ORB_SLAM2
{

// src/ORBmatcher.cc:63-157
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    const bool bFactor = th != 1.0;  // :67
    Queries qs;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint *pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;  // :73
        if (pMP->isBad()) continue;         // :76
        const int &nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);  // :84
        if (bFactor) r *= th;
        // :90-91 GetFeaturesInArea(mTrackProjX, mTrackProjY, r * mvScaleFactors[level], level - 1, level); the right-image
        // gate of :114-119 compares against the same product
        qs.Add(pMP, (int)iMP, pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
               nPredictedLevel, pMP->mTrackProjXR);
    }
    std::vector<int32_t> match;
    RunSearch(F, qs, TH_HIGH, mfNNratio, 1, match);
    int nmatches = 0;
    for (size_t k = 0; k < match.size(); ++k)
        if (match[k] >= 0) {  // :150-151
            F.mvpMapPoints[(size_t)match[k]] = qs.mp[k];
            nmatches++;
        }
    return nmatches;
}

// src/ORBmatcher.cc:1757-1867 (Tracking::Relocalization): the keyframe's MapPoints projected into the current frame; a slot of
// the frame is taken by ANY MapPoint (:1812), the rotation histogram uses the keyframe's keypoint angles
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                   const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const float factor = 1.0f / HISTO_LENGTH;
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    Queries qs;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP) continue;
        if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;  // :1778
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D = cv::norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;  // :1799
        int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        qs.Add(pMP, (int)i, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, 0.f, ORBFE_PROJ_CLAIMS);  // :1806
    }
    std::vector<int32_t> match;
    RunSearch(CurrentFrame, qs, ORBdist, 0.f, 0, match, /*any_point_blocks*/ true);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (size_t k = 0; k < match.size(); ++k) {
        if (match[k] < 0) continue;
        const int bestIdx2 = match[k], i = qs.src[k];
        CurrentFrame.mvpMapPoints[bestIdx2] = qs.mp[k];  // :1826
        nmatches++;
        if (mbCheckOrientation) {
            float rot = pKF->mvKeysUn[i].angle - CurrentFrame.mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {  // :1846-1863
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    CurrentFrame.mvpMapPoints[rotHist[i][j]] = NULL;
                    nmatches--;
                }
    }
    return nmatches;
}

// src/ORBmatcher.cc:378-470 (LoopClosing::ComputeSim3 / SearchAndFuse): map points projected into a keyframe with a Sim3
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                                   int th)
{
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());  // :393, not updated by the loop
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    Queries qs;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist = cv::norm(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist) continue;
        int nPredictedLevel = pMP->PredictScale(dist, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        // :434 KeyFrame::GetFeaturesInArea(u, v, radius) + the level test of :447-448 inside the candidate loop
        qs.Add(pMP, iMP, u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.f, ORBFE_PROJ_CLAIMS);
    }
    std::vector<int32_t> match(qs.q.size(), -1);
    if (!qs.q.empty() && !pKF->mvKeysUn.empty()) {
        FrameSide fs(pKF, &vpMatched);
        const orbfe_status s = orbfe_search_by_projection(
            t_matcher.get(), fs.desc, fs.xy.data(), fs.oct.data(), (int32_t)pKF->mvKeysUn.size(), fs.cell_off.data(), fs.cell_idx.data(),
            (float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, NULL, fs.blocked.data(),
            qs.q.data(), qs.desc.data(), (int32_t)qs.q.size(), TH_LOW, 0.f, 0, match.data(), NULL, NULL);
        if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByProjection (orbfe): ") + orbfe_last_error());
    }
    int nmatches = 0;
    for (size_t k = 0; k < match.size(); ++k)
        if (match[k] >= 0) {  // :463-464
            vpMatched[(size_t)match[k]] = qs.mp[k];
            nmatches++;
        }
    return nmatches;
}

// src/ORBmatcher.cc:1578-1724; points_last / points_current: the extra outputs of perfect/src/ORBmatcher.cc:1727-1911
typedef std::function<void(std::vector<int> *, int &, int &, int &)> ThreeMaxima;  // ORBmatcher::ComputeThreeMaxima is protected
static int SearchLastFrame(const bool mbCheckOrientation, Frame &CurrentFrame, const Frame &LastFrame, const float th,
                           const bool bMono, std::vector<cv::Point2f> *points_last, std::vector<cv::Point2f> *points_current,
                           const ThreeMaxima &three_maxima)
{
    const int HISTO_LENGTH = ORBmatcher::HISTO_LENGTH;
    const float factor = 1.0f / HISTO_LENGTH;  // :1586
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;    // :1604
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;  // :1607
    Queries qs;
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();  // :1620-1632
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        int nLastOctave = LastFrame.mvKeys[i].octave;
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];  // :1642
        int minLevel, maxLevel;
        if (bForward) { minLevel = nLastOctave; maxLevel = -1; }            // :1645 GetFeaturesInArea(u, v, radius, nLastOctave)
        else if (bBackward) { minLevel = 0; maxLevel = nLastOctave; }       // :1647
        else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }    // :1649
        const float ur = u - CurrentFrame.mbf * invzc;                      // :1656
        qs.Add(pMP, i, u, v, radius, minLevel, maxLevel, ur);
    }
    std::vector<int32_t> match;
    RunSearch(CurrentFrame, qs, ORBmatcher::TH_HIGH, 0.f, 0, match);
    int nmatches = 0;
    std::vector<int> rotHist[ORBmatcher::HISTO_LENGTH];
    for (size_t k = 0; k < match.size(); ++k) {
        if (match[k] < 0) continue;
        const int bestIdx2 = match[k], i = qs.src[k];
        CurrentFrame.mvpMapPoints[bestIdx2] = qs.mp[k];  // :1675-1676
        nmatches++;
        if (points_last) {
            points_last->push_back(LastFrame.mvKeys[i].pt);
            points_current->push_back(CurrentFrame.mvKeys[bestIdx2].pt);
        }
        if (mbCheckOrientation) {  // :1679-1689
            float rot = LastFrame.mvKeysUn[i].angle - CurrentFrame.mvKeysUn[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(bestIdx2);
        }
    }
    if (mbCheckOrientation) {  // :1696-1719
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, ind1, ind2, ind3);  // :1912-1957, the reference's own member
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    CurrentFrame.mvpMapPoints[rotHist[i][j]] = static_cast<MapPoint *>(NULL);
                    nmatches--;
                }
    }
    return nmatches;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    return SearchLastFrame(mbCheckOrientation, CurrentFrame, LastFrame, th, bMono, NULL, NULL,
                           [this](std::vector<int> *h, int &a, int &b, int &c) { ComputeThreeMaxima(h, HISTO_LENGTH, a, b, c); });
}

#ifdef ORBFE_SHIM_PERFECT
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono,
                                   std::vector<cv::Point2f> &points_last, std::vector<cv::Point2f> &points_current)
{
    return SearchLastFrame(mbCheckOrientation, CurrentFrame, LastFrame, th, bMono, &points_last, &points_current,
                           [this](std::vector<int> *h, int &a, int &b, int &c) { ComputeThreeMaxima(h, HISTO_LENGTH, a, b, c); });
}
#endif

// src/ORBmatcher.cc:827-1012 (LocalMapping::CreateNewMapPoints, once per neighbour keyframe): Hamming + epipolar gate of every
// unmatched keyframe-1 feature against its vocabulary node's keyframe-2 features in ONE orbfe_search_for_triangulation call;
// the epipole, the eligibility flags and the rotation histogram stay here
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo)
{
    // :833-843 the epipole of keyframe 1 in keyframe 2
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<int> vMatches12((size_t)n1, -1);
    int nmatches = 0;
    if (n1 > 0 && n2 > 0) {
        std::vector<float> xy1((size_t)n1 * 2), xy2((size_t)n2 * 2);
        std::vector<int32_t> oct2((size_t)n2);
        std::vector<uint8_t> e1((size_t)n1), s1((size_t)n1), e2((size_t)n2), s2((size_t)n2), t1, t2;
        for (int i = 0; i < n1; ++i) {
            xy1[2 * (size_t)i] = pKF1->mvKeysUn[(size_t)i].pt.x;
            xy1[2 * (size_t)i + 1] = pKF1->mvKeysUn[(size_t)i].pt.y;
            s1[(size_t)i] = pKF1->mvuRight[(size_t)i] >= 0;                                   // :866
            e1[(size_t)i] = !pKF1->GetMapPoint((size_t)i) && (!bOnlyStereo || s1[(size_t)i]);  // :860-870
        }
        for (int i = 0; i < n2; ++i) {
            xy2[2 * (size_t)i] = pKF2->mvKeysUn[(size_t)i].pt.x;
            xy2[2 * (size_t)i + 1] = pKF2->mvKeysUn[(size_t)i].pt.y;
            oct2[(size_t)i] = pKF2->mvKeysUn[(size_t)i].octave;
            s2[(size_t)i] = pKF2->mvuRight[(size_t)i] >= 0;                                   // :887
            e2[(size_t)i] = !pKF2->GetMapPoint((size_t)i) && (!bOnlyStereo || s2[(size_t)i]);  // :881-891 (vbMatched2 is never set)
        }
        Csr c1, c2;
        Flatten(pKF1->mFeatVec, c1);
        Flatten(pKF2->mFeatVec, c2);
        float F[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
        std::vector<int32_t> m12((size_t)n1, -1);
        const orbfe_status st = orbfe_search_for_triangulation(
            t_matcher.get(), Rows(pKF1->mDescriptors, t1), xy1.data(), e1.data(), s1.data(), n1, c1.node.data(), c1.off.data(), c1.idx.data(),
            (int)c1.node.size(), Rows(pKF2->mDescriptors, t2), xy2.data(), oct2.data(), e2.data(), s2.data(), n2, c2.node.data(),
            c2.off.data(), c2.idx.data(), (int)c2.node.size(), F, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
            (int)pKF2->mvScaleFactors.size(), TH_LOW, m12.data());
        if (st != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchForTriangulation (orbfe): ") + orbfe_last_error());
        std::vector<int> rotHist[HISTO_LENGTH];
        const float factor = 1.0f / HISTO_LENGTH;
        for (int i = 0; i < n1; ++i) {
            if (m12[(size_t)i] < 0) continue;
            vMatches12[(size_t)i] = m12[(size_t)i];  // :917-918
            nmatches++;
            if (mbCheckOrientation) {
                float rot = pKF1->mvKeysUn[(size_t)i].angle - pKF2->mvKeysUn[(size_t)m12[(size_t)i]].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(i);
            }
        }
        if (mbCheckOrientation) {  // :966-985
            int ind1 = -1, ind2 = -1, ind3 = -1;
            ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
            for (int i = 0; i < HISTO_LENGTH; i++) {
                if (i == ind1 || i == ind2 || i == ind3) continue;
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    vMatches12[(size_t)rotHist[i][j]] = -1;
                    nmatches--;
                }
            }
        }
    }
    vMatchedPairs.clear();  // :987-997
    vMatchedPairs.reserve((size_t)std::max(nmatches, 0));
    for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
        if (vMatches12[i] < 0) continue;
        vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
    }
    return nmatches;
}

// src/ORBmatcher.cc:1031-1182
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    const float &bf = pKF->mbf;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    // ---- phase 1: every gate that does not depend on what the loop mutates (:1060-1101); one query per point that passes ----
    Queries qs;
    std::vector<int> slot((size_t)nMPs, -1);  // query index of MapPoint i, -1 = gated out
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;  // :1068
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;  // :1080
        const float ur = u - bf * invz;  // :1124
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;  // :1090
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;  // :1096
        int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        // :1103 GetFeaturesInArea(u, v, radius), :1119 the level window, :1122-1145 the reprojection-error gate: on the device
        slot[(size_t)i] = (int)qs.q.size();
        qs.Add(pMP, i, u, v, radius, nPredictedLevel - 1, nPredictedLevel, ur, ORBFE_PROJ_CHI2_GATE);
    }
    // ---- the candidate scans: best candidate per point, first in list order on ties (:1149-1156), accepted at TH_LOW ----
    std::vector<int32_t> bestIdx;
    RunKeyFrameSearch(pKF, qs, TH_LOW, true, bestIdx, "Fuse");
    // ---- phase 2: the loop's decisions, in order, on the live objects (:1049-1056, :1159-1180) ----
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        if (slot[(size_t)i] < 0) continue;
        const int k = slot[(size_t)i];
        if (bestIdx[(size_t)k] >= 0) {  // :1159 bestDist <= TH_LOW
            MapPoint *pMPinKF = pKF->GetMapPoint((size_t)bestIdx[(size_t)k]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, (size_t)bestIdx[(size_t)k]);
                pKF->AddMapPoint(pMP, (size_t)bestIdx[(size_t)k]);
            }
            nFused++;
        }
    }
    return nFused;
}

// src/ORBmatcher.cc:1198-1299 (LoopClosing::SearchAndFuse): the Sim3 form -- no stereo gate, a hit on an occupied slot is
// reported in vpReplacePoint instead of replaced; the same two phases as the plain Fuse
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    const float &fx = pKF->fx;
    const float &fy = pKF->fy;
    const float &cx = pKF->cx;
    const float &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();  // :1212, fixed for the whole loop
    const int nPoints = (int)vpPoints.size();
    Queries qs;
    std::vector<int> slot((size_t)nPoints, -1);
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint *pMP = vpPoints[iMP];
        if (!pMP) continue;
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;  // :1224 (neither changes inside the loop)
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;
        const float invz = 1.0 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        // :1257 GetFeaturesInArea(u, v, radius) and the level window of :1268 run on the device
        slot[(size_t)iMP] = (int)qs.q.size();
        qs.Add(pMP, iMP, u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.f, 0);
    }
    std::vector<int32_t> bestIdx;
    RunKeyFrameSearch(pKF, qs, TH_LOW, false, bestIdx, "Fuse");
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {  // :1282-1296 on the live objects, in order
        if (slot[(size_t)iMP] < 0) continue;
        const int k = slot[(size_t)iMP];
        if (bestIdx[(size_t)k] >= 0) {  // :1281 bestDist <= TH_LOW
            MapPoint *pMP = vpPoints[iMP];
            MapPoint *pMPinKF = pKF->GetMapPoint((size_t)bestIdx[(size_t)k]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[(size_t)iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, (size_t)bestIdx[(size_t)k]);
                pKF->AddMapPoint(pMP, (size_t)bestIdx[(size_t)k]);
            }
            nFused++;
        }
    }
    return nFused;
}

// src/ORBmatcher.cc:1334-1516 (LoopClosing::ComputeSim3): the points of each keyframe are searched in the other one under the
// candidate similarity, a match is kept when both directions agree.  No decision depends on an earlier one: each direction
// is one device call (window search on the other keyframe's grid + Hamming).
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th)
{
    const float &fx = pKF1->fx;
    const float &fy = pKF1->fy;
    const float &cx = pKF1->cx;
    const float &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1((size_t)N1, false), vbAlreadyMatched2((size_t)N2, false);
    for (int i = 0; i < N1; i++) {
        MapPoint *pMP = vpMatches12[(size_t)i];
        if (pMP) {
            vbAlreadyMatched1[(size_t)i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[(size_t)idx2] = true;
        }
    }
    // one direction: the points of keyframe A (pose Raw, taw) are taken into keyframe B by (sRba, tba) and looked up there;
    // GetFeaturesInArea on B's grid, the level window and the Hamming scan are one device call per direction
    auto search = [&](const std::vector<MapPoint *> &pts, const std::vector<bool> &already, const cv::Mat &Raw, const cv::Mat &taw,
                      const cv::Mat &sRba, const cv::Mat &tba, KeyFrame *pKFb, std::vector<int> &vnMatch) {
        Queries qs;
        for (int i = 0; i < (int)pts.size(); i++) {
            MapPoint *pMP = pts[(size_t)i];
            if (!pMP || already[(size_t)i]) continue;
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();
            cv::Mat p3Da = Raw * p3Dw + taw;
            cv::Mat p3Db = sRba * p3Da + tba;
            if (p3Db.at<float>(2) < 0.0) continue;
            const float invz = 1.0 / p3Db.at<float>(2);
            const float x = p3Db.at<float>(0) * invz;
            const float y = p3Db.at<float>(1) * invz;
            const float u = fx * x + cx;
            const float v = fy * y + cy;
            if (!pKFb->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = cv::norm(p3Db);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKFb);
            const float radius = th * pKFb->mvScaleFactors[nPredictedLevel];
            qs.Add(pMP, i, u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.f, 0);
        }
        std::vector<int32_t> best;
        RunKeyFrameSearch(pKFb, qs, TH_HIGH, false, best, "SearchBySim3");
        for (size_t k = 0; k < best.size(); ++k)
            if (best[k] >= 0) vnMatch[(size_t)qs.src[k]] = best[k];  // bestDist <= TH_HIGH (:1451, :1527)
    };
    std::vector<int> vnMatch1((size_t)N1, -1), vnMatch2((size_t)N2, -1);
    search(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1);  // :1380-1453
    search(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, vnMatch2);  // :1455-1529
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {  // :1532-1545
        int idx2 = vnMatch1[(size_t)i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[(size_t)idx2];
            if (idx1 == i1) {
                vpMatches12[(size_t)i1] = vpMapPoints2[(size_t)idx2];
                nFound++;
            }
        }
    }
    return nFound;
}

// src/ORBmatcher.cc:523-651 (Tracking::MonocularInitialization).  A candidate is skipped when an earlier query already holds
// it at a distance <= its own (:573), so best / second-best of a query depend on the matches made before it: the device
// returns every window's candidates with their distances (orbfe_window_distances, one call), the in-order rule runs here.
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize)
{
    int nmatches = 0;
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(F2.mvKeysUn.size(), INT_MAX);
    std::vector<int> vnMatches21(F2.mvKeysUn.size(), -1);
    const size_t n1 = F1.mvKeysUn.size();
    // one query per level-0 feature of F1 (:543-545): GetFeaturesInArea(prev.x, prev.y, windowSize, 0, 0) on F2's grid and the
    // distance of every candidate come back as lists in the reference's candidate order
    std::vector<orbfe_proj_query> qs;
    std::vector<uint8_t> qdesc, t1;
    std::vector<int> slot(n1, -1);
    const uint8_t *d1 = Rows(F1.mDescriptors, t1);
    for (size_t i1 = 0; i1 < n1; i1++) {
        const int level1 = F1.mvKeysUn[i1].octave;
        if (level1 > 0) continue;
        orbfe_proj_query e;
        e.u = vbPrevMatched[i1].x; e.v = vbPrevMatched[i1].y; e.r = (float)windowSize;
        e.min_level = level1; e.max_level = level1; e.ur = 0.f; e.flags = 0; e.pad = 0;
        slot[i1] = (int)qs.size();
        qs.push_back(e);
        qdesc.insert(qdesc.end(), d1 + i1 * 32, d1 + i1 * 32 + 32);
    }
    const int nq = (int)qs.size();
    std::vector<uint32_t> off((size_t)nq + 1, 0), ent;
    if (nq > 0 && F2.N > 0) {
        FrameSide fs(F2);
        size_t cap = std::max<size_t>((size_t)nq * 64, 4096);
        for (int attempt = 0;; ++attempt) {
            ent.resize(cap);
            const orbfe_status st = orbfe_window_distances(
                t_matcher.get(), fs.desc, fs.xy.data(), fs.oct.data(), F2.N, fs.cell_off.data(), fs.cell_idx.data(), Frame::mnMinX,
                Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, qs.data(), qdesc.data(), nq, off.data(),
                ent.data(), (int32_t)cap);
            if (st == ORBFE_ERR_CAP && attempt == 0) { cap = off[(size_t)nq]; continue; }
            if (st != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchForInitialization (orbfe): ") + orbfe_last_error());
            break;
        }
    }
    for (size_t i1 = 0; i1 < n1; i1++) {  // :538-617 on the device's distances
        if (slot[i1] < 0) continue;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (uint32_t k = off[(size_t)slot[i1]]; k < off[(size_t)slot[i1] + 1]; k++) {
            const size_t i2 = ent[k] & 0xFFFFu;
            const int d = (int)(ent[k] >> 16);
            if (vMatchedDistance[i2] <= d) continue;
            if (d < bestDist) {
                bestDist2 = bestDist;
                bestDist = d;
                bestIdx2 = (int)i2;
            } else if (d < bestDist2) {
                bestDist2 = d;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[(size_t)bestIdx2] >= 0) {
                    vnMatches12[(size_t)vnMatches21[(size_t)bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[(size_t)bestIdx2] = (int)i1;
                vMatchedDistance[(size_t)bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = F1.mvKeysUn[i1].angle - F2.mvKeysUn[(size_t)bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[(size_t)idx1] >= 0) {
                    vnMatches12[(size_t)idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[(size_t)vnMatches12[i1]].pt;
    return nmatches;
}

#ifdef ORBFE_SHIM_STANDALONE
// With every public member supplied above, src/ORBmatcher.cc can leave the build altogether: -DORBFE_SHIM_STANDALONE adds
// the rest of the class -- the constants (src/ORBmatcher.cc:39-41), the constructor (:43) and the three protected helpers
// (:159-167, :175-191, :1912-1957) -- so that this file alone is the ORBmatcher translation unit.
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

float ORBmatcher::RadiusByViewingCos(const float &viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

// squared distance of kp2 to the epipolar line of kp1 against the 95 % chi-square bound of kp2's level (float arithmetic, the
// bound in double as `3.84 * float` makes it)
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint &kp1, const cv::KeyPoint &kp2, const cv::Mat &F12, const KeyFrame *pKF2)
{
    float l[3];
    for (int k = 0; k < 3; k++) l[k] = kp1.pt.x * F12.at<float>(0, k) + kp1.pt.y * F12.at<float>(1, k) + F12.at<float>(2, k);
    const float num = l[0] * kp2.pt.x + l[1] * kp2.pt.y + l[2];
    const float den = l[0] * l[0] + l[1] * l[1];
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave];
}

// the three fullest bins, first bin wins ties; the second / third are dropped when they hold less than a tenth of the first
void ORBmatcher::ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
{
    int top[3] = {0, 0, 0}, idx[3] = {-1, -1, -1};
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        int r = 3;
        while (r > 0 && s > top[r - 1]) r--;   // rank of bin i among the best so far (strictly greater moves up)
        if (r == 3) continue;
        for (int k = 2; k > r; k--) { top[k] = top[k - 1]; idx[k] = idx[k - 1]; }
        top[r] = s;
        idx[r] = i;
    }
    if (top[1] < 0.1f * (float)top[0]) idx[1] = idx[2] = -1;
    else if (top[2] < 0.1f * (float)top[0]) idx[2] = -1;
    ind1 = idx[0];
    ind2 = idx[1];
    ind3 = idx[2];
}
#endif  // ORBFE_SHIM_STANDALONE

int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    return orbfe_hamming(a.ptr<uint8_t>(0), b.ptr<uint8_t>(0));
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
{
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));  // :222
    orbfe_matcher *m = t_matcher.get();
    std::vector<uint8_t> validKF, tk, tf;
    Valid(vpMapPointsKF, validKF);
    Csr kf, f;
    Flatten(pKF->mFeatVec, kf);
    Flatten(F.mFeatVec, f);
    std::vector<float> angKF, angF;
    Angles(pKF->mvKeysUn, angKF);  // :304
    Angles(F.mvKeys, angF);        // :308
    std::vector<int32_t> match((size_t)std::max(F.N, 1), -1);
    int n = 0;
    const orbfe_status s = orbfe_search_by_bow(m, Rows(pKF->mDescriptors, tk), (int)validKF.size(), validKF.data(), angKF.data(),
                                               kf.node.data(), kf.off.data(), kf.idx.data(), (int)kf.node.size(),
                                               Rows(F.mDescriptors, tf), F.N, NULL, angF.data(), f.node.data(), f.off.data(),
                                               f.idx.data(), (int)f.node.size(), mfNNratio, TH_LOW, 0, mbCheckOrientation ? 1 : 0,
                                               match.data(), &n);
    if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByBoW (orbfe): ") + orbfe_last_error());
    for (int i = 0; i < F.N; ++i)
        if (match[(size_t)i] >= 0) vpMapPointMatches[(size_t)i] = vpMapPointsKF[(size_t)match[(size_t)i]];  // :298
    return n;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
{
    const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint *> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));  // :677
    orbfe_matcher *m = t_matcher.get();
    std::vector<uint8_t> v1, v2, t1, t2;
    Valid(vpMapPoints1, v1);
    Valid(vpMapPoints2, v2);
    Csr c1, c2;
    Flatten(pKF1->mFeatVec, c1);
    Flatten(pKF2->mFeatVec, c2);
    std::vector<float> a1, a2;
    Angles(pKF1->mvKeysUn, a1);
    Angles(pKF2->mvKeysUn, a2);
    std::vector<int32_t> match(std::max<size_t>(v2.size(), 1), -1);
    int n = 0;
    const orbfe_status s = orbfe_search_by_bow(m, Rows(pKF1->mDescriptors, t1), (int)v1.size(), v1.data(), a1.data(), c1.node.data(),
                                               c1.off.data(), c1.idx.data(), (int)c1.node.size(), Rows(pKF2->mDescriptors, t2),
                                               (int)v2.size(), v2.data(), a2.data(), c2.node.data(), c2.off.data(), c2.idx.data(),
                                               (int)c2.node.size(), mfNNratio, TH_LOW, 1, mbCheckOrientation ? 1 : 0,
                                               match.data(), &n);
    if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByBoW (orbfe): ") + orbfe_last_error());
    for (size_t i2 = 0; i2 < v2.size(); ++i2)  // the reference's output is indexed by the KF1 feature (:751)
        if (match[i2] >= 0) vpMatches12[(size_t)match[i2]] = vpMapPoints2[i2];
    return n;
}

}  // namespace ORB_SLAM2
