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
// One pattern throughout: this file FLATTENS the objects (poses, map-point positions / normals / distance ranges, keypoints,
// descriptors), the C-ABI does the work, and the decisions are REPLAYED on the live objects in the reference's order:
//   orbfe_project_points / orbfe_proj_queries_local_map   the per-point projection and its gates (host, csrc/orbfe_hostgeom.hip)
//   orbfe_search_by_projection(_chi2) etc.                the candidate scans (GetFeaturesInArea + Hamming + the rule that couples
//                                                         the queries): ONE device call per invocation
//   orbfe_rotation_consistency                            the rotation histogram + ComputeThreeMaxima
// What stays here on cv::Mat are the few POSE-level expressions (-Rcw.t() * tcw, Scw / scw, ...): evaluated through whatever
// cv::Mat the application links, they are exact in a real OpenCV build (double-accumulating gemm) and in the test stub alike.
// MapPoint::PredictScale is the application's own member and is called, not restated.  The device call is orbfe_search_by_projection(_chi2) for the four
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
    void Add(ORB_SLAM2::MapPoint *p, int from, float u, float v, float r, int minLevel, int maxLevel, float ur, int flags)
    {
        orbfe_proj_query e;
        e.u = u; e.v = v; e.r = r; e.min_level = minLevel; e.max_level = maxLevel; e.ur = ur; e.flags = flags; e.pad = 0;
        q.push_back(e);
        const cv::Mat d = p->GetDescriptor();
        desc.insert(desc.end(), d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + 32);
        mp.push_back(p);
        src.push_back(from);
    }
};

// 3x3 / 3x1 CV_32F matrices as the flat arrays the C-ABI takes
struct Pose {
    float R[9], t[3];
    Pose(const cv::Mat &Rm, const cv::Mat &tm)
    {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R[3 * r + c] = Rm.at<float>(r, c);
            t[r] = tm.at<float>(r);
        }
    }
};
struct Vec3 {
    float v[3];
    explicit Vec3(const cv::Mat &m) { for (int r = 0; r < 3; ++r) v[r] = m.at<float>(r); }
};

// The map points of one call that passed the object-level filters (null / bad / already found), flattened; Project() runs
// orbfe_project_points on them: everything per point that the reference computes between GetWorldPos() and PredictScale().
struct PointList {
    std::vector<ORB_SLAM2::MapPoint *> mp;
    std::vector<int> src;
    std::vector<float> pos, nrm, dmin, dmax;
    std::vector<float> u, v, invz, dist, ur;
    std::vector<uint8_t> ok;
    void Add(ORB_SLAM2::MapPoint *p, int from, bool with_normal, bool with_range)
    {
        mp.push_back(p);
        src.push_back(from);
        const Vec3 w(p->GetWorldPos());
        pos.insert(pos.end(), w.v, w.v + 3);
        if (with_normal) {
            const Vec3 n(p->GetNormal());
            nrm.insert(nrm.end(), n.v, n.v + 3);
        }
        if (with_range) {
            dmax.push_back(p->GetMaxDistanceInvariance());
            dmin.push_back(p->GetMinDistanceInvariance());
        }
    }
    size_t size() const { return mp.size(); }
    void Project(const Pose &P, const Pose *second, const float *Ow, float fx, float fy, float cx, float cy, float bf, float minx,
                 float maxx, float miny, float maxy, int flags)
    {
        const size_t n = mp.size();
        u.resize(n); v.resize(n); invz.resize(n); dist.resize(n); ur.resize(n);
        ok.assign(n, 0);
        if (n == 0) return;
        const orbfe_status s = orbfe_project_points(P.R, P.t, second ? second->R : NULL, second ? second->t : NULL, Ow, fx, fy, cx, cy, bf,
                                                    minx, maxx, miny, maxy, flags, (int32_t)n, pos.data(), nrm.empty() ? NULL : nrm.data(),
                                                    dmin.empty() ? NULL : dmin.data(), dmax.empty() ? NULL : dmax.data(), u.data(),
                                                    v.data(), invz.data(), dist.data(), ur.data(), ok.data());
        if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher (orbfe): orbfe_project_points: ") + orbfe_last_error());
    }
};

// rotation-consistency check over the accepted matches (angle pairs in acceptance order): which of them fall outside the
// three fullest histogram bins (src/ORBmatcher.cc:308-316, :338-360, :1912-1957)
void RotationOutliers(const std::vector<float> &a, const std::vector<float> &b, int histo_len, std::vector<uint8_t> &drop)
{
    drop.assign(a.size(), 0);
    if (a.empty()) return;
    if (orbfe_rotation_consistency(a.data(), b.data(), (int32_t)a.size(), histo_len, drop.data()) != ORBFE_OK)
        throw std::runtime_error(std::string("ORBmatcher (orbfe): orbfe_rotation_consistency: ") + orbfe_last_error());
}

void RunSearch(ORB_SLAM2::Frame &F, const Queries &qs, int th, float nnratio, int ratio_rule, std::vector<int32_t> &match,
               bool any_point_blocks = false)
{
    match.assign(qs.q.size(), -1);
    if (qs.q.empty() || F.N == 0) return;
    FrameSide fs(F);
    if (any_point_blocks)  // :1812 a slot holding ANY map point is taken
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

// the projected points that passed become window queries: level from the application's own MapPoint::PredictScale, search
// radius th * scale factor of that level, levels [level - 1, level + up]
template <class Target>
void LevelQueries(const PointList &pl, Target *where, const std::vector<float> &scale_factors, float th, int up, int flags, bool with_ur,
                  Queries &qs, std::vector<int> *slot_of_src)
{
    for (size_t k = 0; k < pl.size(); ++k) {
        if (!pl.ok[k]) continue;
        const int level = pl.mp[k]->PredictScale(pl.dist[k], where);
        if (slot_of_src) (*slot_of_src)[(size_t)pl.src[k]] = (int)qs.q.size();
        qs.Add(pl.mp[k], pl.src[k], pl.u[k], pl.v[k], th * scale_factors[(size_t)level], level - 1, level + up, with_ur ? pl.ur[k] : 0.f, flags);
    }
}
}  // namespace

namespace ORB_SLAM2
{

// src/ORBmatcher.cc:63-157 (Tracking::SearchLocalPoints)
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    const size_t n = vpMapPoints.size();
    std::vector<uint8_t> in_view(n), bad(n), obs(n);
    std::vector<int32_t> level(n);
    std::vector<float> view_cos(n), uvr(3 * n);
    for (size_t i = 0; i < n; ++i) {
        MapPoint *p = vpMapPoints[i];
        in_view[i] = p->mbTrackInView;
        bad[i] = p->isBad();
        obs[i] = p->Observations() > 0;
        level[i] = p->mnTrackScaleLevel;
        view_cos[i] = p->mTrackViewCos;
        uvr[3 * i] = p->mTrackProjX;
        uvr[3 * i + 1] = p->mTrackProjY;
        uvr[3 * i + 2] = p->mTrackProjXR;
    }
    Queries qs;
    qs.q.resize(n);
    qs.src.resize(n);
    int32_t nq = 0;
    if (orbfe_proj_queries_local_map(F.mvScaleFactors.data(), (int32_t)n, in_view.data(), bad.data(), level.data(), view_cos.data(), uvr.data(),
                                     obs.data(), th, qs.q.data(), qs.src.data(), &nq) != ORBFE_OK)
        throw std::runtime_error(std::string("ORBmatcher::SearchByProjection (orbfe): ") + orbfe_last_error());
    qs.q.resize((size_t)nq);
    qs.src.resize((size_t)nq);
    for (int k = 0; k < nq; ++k) {
        MapPoint *p = vpMapPoints[(size_t)qs.src[(size_t)k]];
        const cv::Mat d = p->GetDescriptor();
        qs.desc.insert(qs.desc.end(), d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + 32);
        qs.mp.push_back(p);
    }
    std::vector<int32_t> match;
    RunSearch(F, qs, TH_HIGH, mfNNratio, 1, match);
    int found = 0;
    for (size_t k = 0; k < match.size(); ++k)
        if (match[k] >= 0) {  // :150-151
            F.mvpMapPoints[(size_t)match[k]] = qs.mp[k];
            ++found;
        }
    return found;
}

// src/ORBmatcher.cc:1757-1867 (Tracking::Relocalization): the keyframe's MapPoints projected into the current frame; a slot of
// the frame is taken by ANY MapPoint (:1812), the rotation histogram uses the keyframe's keypoint angles
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th,
                                   const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint *> kfPoints = pKF->GetMapPointMatches();
    PointList pl;
    for (size_t i = 0; i < kfPoints.size(); ++i) {
        MapPoint *p = kfPoints[i];
        if (p && !p->isBad() && !sAlreadyFound.count(p)) pl.Add(p, (int)i, false, true);  // :1774-1778
    }
    const Vec3 centre(Ow);
    pl.Project(Pose(Rcw, tcw), NULL, centre.v, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, 0.f, CurrentFrame.mnMinX,
               CurrentFrame.mnMaxX, CurrentFrame.mnMinY, CurrentFrame.mnMaxY, ORBFE_PJ_UV_CHAINED | ORBFE_PJ_BOUNDS_CLOSED);
    Queries qs;
    LevelQueries(pl, &CurrentFrame, CurrentFrame.mvScaleFactors, th, 1, ORBFE_PROJ_CLAIMS, false, qs, NULL);  // :1803-1806
    std::vector<int32_t> match;
    RunSearch(CurrentFrame, qs, ORBdist, 0.f, 0, match, /*any_point_blocks*/ true);
    std::vector<float> angKF, angF;
    std::vector<int> hit;
    int found = 0;
    for (size_t k = 0; k < match.size(); ++k) {
        if (match[k] < 0) continue;
        CurrentFrame.mvpMapPoints[(size_t)match[k]] = qs.mp[k];  // :1826
        ++found;
        hit.push_back(match[k]);
        angKF.push_back(pKF->mvKeysUn[(size_t)qs.src[k]].angle);
        angF.push_back(CurrentFrame.mvKeysUn[(size_t)match[k]].angle);
    }
    if (mbCheckOrientation) {  // :1846-1863
        std::vector<uint8_t> drop;
        RotationOutliers(angKF, angF, HISTO_LENGTH, drop);
        for (size_t e = 0; e < drop.size(); ++e)
            if (drop[e]) {
                CurrentFrame.mvpMapPoints[(size_t)hit[e]] = NULL;
                --found;
            }
    }
    return found;
}

// src/ORBmatcher.cc:378-470 (LoopClosing::ComputeSim3 / SearchAndFuse): map points projected into a keyframe with a Sim3
int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched,
                                   int th)
{
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint *> known(vpMatched.begin(), vpMatched.end());  // :393, not updated by the loop
    known.erase(static_cast<MapPoint *>(NULL));
    PointList pl;
    for (size_t i = 0; i < vpPoints.size(); ++i) {
        MapPoint *p = vpPoints[i];
        if (!p->isBad() && !known.count(p)) pl.Add(p, (int)i, true, true);  // :403
    }
    const Vec3 centre(Ow);
    pl.Project(Pose(Rcw, tcw), NULL, centre.v, pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0.f, (float)pKF->mnMinX, (float)pKF->mnMaxX,
               (float)pKF->mnMinY, (float)pKF->mnMaxY, ORBFE_PJ_SKIP_NEG_DEPTH);
    Queries qs;
    // :434 KeyFrame::GetFeaturesInArea(u, v, radius) + the level test of :447-448 inside the candidate loop: on the device
    LevelQueries(pl, pKF, pKF->mvScaleFactors, (float)th, 0, ORBFE_PROJ_CLAIMS, false, qs, NULL);
    std::vector<int32_t> match(qs.q.size(), -1);
    if (!qs.q.empty() && !pKF->mvKeysUn.empty()) {
        FrameSide fs(pKF, &vpMatched);
        const orbfe_status s = orbfe_search_by_projection(
            t_matcher.get(), fs.desc, fs.xy.data(), fs.oct.data(), (int32_t)pKF->mvKeysUn.size(), fs.cell_off.data(), fs.cell_idx.data(),
            (float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, NULL, fs.blocked.data(),
            qs.q.data(), qs.desc.data(), (int32_t)qs.q.size(), TH_LOW, 0.f, 0, match.data(), NULL, NULL);
        if (s != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByProjection (orbfe): ") + orbfe_last_error());
    }
    int found = 0;
    for (size_t k = 0; k < match.size(); ++k)
        if (match[k] >= 0) {  // :463-464
            vpMatched[(size_t)match[k]] = qs.mp[k];
            ++found;
        }
    return found;
}

// src/ORBmatcher.cc:1578-1724 (Tracking::TrackWithMotionModel); points_last / points_current: the extra outputs of
// perfect/src/ORBmatcher.cc:1727-1911
static int SearchLastFrame(const bool check_orientation, Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono,
                           std::vector<cv::Point2f> *points_last, std::vector<cv::Point2f> *points_current)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool forward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;    // :1604
    const bool backward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;  // :1607
    PointList pl;
    for (int i = 0; i < LastFrame.N; ++i)
        if (LastFrame.mvpMapPoints[(size_t)i] && !LastFrame.mvbOutlier[(size_t)i]) pl.Add(LastFrame.mvpMapPoints[(size_t)i], i, false, false);
    pl.Project(Pose(Rcw, tcw), NULL, NULL, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mbf,
               CurrentFrame.mnMinX, CurrentFrame.mnMaxX, CurrentFrame.mnMinY, CurrentFrame.mnMaxY,
               ORBFE_PJ_SKIP_NEG_INVZ | ORBFE_PJ_UV_CHAINED | ORBFE_PJ_BOUNDS_CLOSED);
    Queries qs;
    for (size_t k = 0; k < pl.size(); ++k) {
        if (!pl.ok[k]) continue;
        const int octave = LastFrame.mvKeys[(size_t)pl.src[k]].octave;
        // :1642-1649 the level window follows the motion: ahead -> finer or equal levels only, backwards -> coarser or equal
        const int lo = forward ? octave : (backward ? 0 : octave - 1), hi = forward ? -1 : (backward ? octave : octave + 1);
        MapPoint *p = pl.mp[k];
        qs.Add(p, pl.src[k], pl.u[k], pl.v[k], th * CurrentFrame.mvScaleFactors[(size_t)octave], lo, hi, pl.ur[k],
               ORBFE_PROJ_RIGHT_GATE | (p->Observations() > 0 ? ORBFE_PROJ_CLAIMS : 0));
    }
    std::vector<int32_t> match;
    RunSearch(CurrentFrame, qs, ORBmatcher::TH_HIGH, 0.f, 0, match);
    std::vector<float> angL, angC;
    std::vector<int> hit;
    int found = 0;
    for (size_t k = 0; k < match.size(); ++k) {
        if (match[k] < 0) continue;
        const size_t c = (size_t)match[k], l = (size_t)qs.src[k];
        CurrentFrame.mvpMapPoints[c] = qs.mp[k];  // :1675-1676
        ++found;
        if (points_last) {
            points_last->push_back(LastFrame.mvKeys[l].pt);
            points_current->push_back(CurrentFrame.mvKeys[c].pt);
        }
        hit.push_back(match[k]);
        angL.push_back(LastFrame.mvKeysUn[l].angle);
        angC.push_back(CurrentFrame.mvKeysUn[c].angle);
    }
    if (check_orientation) {  // :1696-1719
        std::vector<uint8_t> drop;
        RotationOutliers(angL, angC, ORBmatcher::HISTO_LENGTH, drop);
        for (size_t e = 0; e < drop.size(); ++e)
            if (drop[e]) {
                CurrentFrame.mvpMapPoints[(size_t)hit[e]] = NULL;
                --found;
            }
    }
    return found;
}

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    return SearchLastFrame(mbCheckOrientation, CurrentFrame, LastFrame, th, bMono, NULL, NULL);
}

#ifdef ORBFE_SHIM_PERFECT
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono,
                                   std::vector<cv::Point2f> &points_last, std::vector<cv::Point2f> &points_current)
{
    return SearchLastFrame(mbCheckOrientation, CurrentFrame, LastFrame, th, bMono, &points_last, &points_current);
}
#endif

// src/ORBmatcher.cc:827-1012 (LocalMapping::CreateNewMapPoints, once per neighbour keyframe): Hamming + epipolar gate of every
// unmatched keyframe-1 feature against its vocabulary node's keyframe-2 features in ONE orbfe_search_for_triangulation call;
// the eligibility flags and the rotation check are assembled / replayed here
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo)
{
    // :833-843 the epipole = keyframe 1's camera centre seen by keyframe 2: one point through the projection call, no gates
    float ex = 0.f, ey = 0.f;
    {
        const Vec3 c1(pKF1->GetCameraCenter());
        const Pose P2(pKF2->GetRotation(), pKF2->GetTranslation());
        uint8_t ok = 0;
        const float big = INFINITY;  // closed bounds at +-inf: nothing is gated, whatever the projection yields
        if (orbfe_project_points(P2.R, P2.t, NULL, NULL, NULL, pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy, 0.f, -big, big, -big, big,
                                 ORBFE_PJ_UV_CHAINED | ORBFE_PJ_BOUNDS_CLOSED, 1, c1.v, NULL, NULL, NULL, &ex, &ey, NULL, NULL, NULL, &ok) != ORBFE_OK)
            throw std::runtime_error(std::string("ORBmatcher::SearchForTriangulation (orbfe): ") + orbfe_last_error());
    }
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<int32_t> m12((size_t)std::max(n1, 1), -1);
    int found = 0;
    if (n1 > 0 && n2 > 0) {
        std::vector<float> xy1((size_t)n1 * 2), xy2((size_t)n2 * 2);
        std::vector<int32_t> oct2((size_t)n2);
        std::vector<uint8_t> e1((size_t)n1), s1((size_t)n1), e2((size_t)n2), s2((size_t)n2), t1, t2;
        for (size_t i = 0; i < (size_t)n1; ++i) {
            xy1[2 * i] = pKF1->mvKeysUn[i].pt.x;
            xy1[2 * i + 1] = pKF1->mvKeysUn[i].pt.y;
            s1[i] = pKF1->mvuRight[i] >= 0;                        // :866
            e1[i] = !pKF1->GetMapPoint(i) && (!bOnlyStereo || s1[i]);  // :860-870
        }
        for (size_t i = 0; i < (size_t)n2; ++i) {
            xy2[2 * i] = pKF2->mvKeysUn[i].pt.x;
            xy2[2 * i + 1] = pKF2->mvKeysUn[i].pt.y;
            oct2[i] = pKF2->mvKeysUn[i].octave;
            s2[i] = pKF2->mvuRight[i] >= 0;                        // :887
            e2[i] = !pKF2->GetMapPoint(i) && (!bOnlyStereo || s2[i]);  // :881-891 (vbMatched2 is never set)
        }
        Csr c1, c2;
        Flatten(pKF1->mFeatVec, c1);
        Flatten(pKF2->mFeatVec, c2);
        float F[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.at<float>(r, c);
        const orbfe_status st = orbfe_search_for_triangulation(
            t_matcher.get(), Rows(pKF1->mDescriptors, t1), xy1.data(), e1.data(), s1.data(), n1, c1.node.data(), c1.off.data(), c1.idx.data(),
            (int)c1.node.size(), Rows(pKF2->mDescriptors, t2), xy2.data(), oct2.data(), e2.data(), s2.data(), n2, c2.node.data(),
            c2.off.data(), c2.idx.data(), (int)c2.node.size(), F, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
            (int)pKF2->mvScaleFactors.size(), TH_LOW, m12.data());
        if (st != ORBFE_OK) throw std::runtime_error(std::string("ORBmatcher::SearchForTriangulation (orbfe): ") + orbfe_last_error());
        std::vector<float> a1, a2;
        std::vector<int> hit;
        for (int i = 0; i < n1; ++i) {
            if (m12[(size_t)i] < 0) continue;  // :917-918
            ++found;
            hit.push_back(i);
            a1.push_back(pKF1->mvKeysUn[(size_t)i].angle);
            a2.push_back(pKF2->mvKeysUn[(size_t)m12[(size_t)i]].angle);
        }
        if (mbCheckOrientation) {  // :966-985
            std::vector<uint8_t> drop;
            RotationOutliers(a1, a2, HISTO_LENGTH, drop);
            for (size_t e = 0; e < drop.size(); ++e)
                if (drop[e]) {
                    m12[(size_t)hit[e]] = -1;
                    --found;
                }
        }
    }
    vMatchedPairs.clear();  // :987-997
    vMatchedPairs.reserve((size_t)std::max(found, 0));
    for (int i = 0; i < n1; ++i)
        if (m12[(size_t)i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[(size_t)i]));
    return found;
}

// src/ORBmatcher.cc:1031-1182 (LocalMapping::SearchInNeighbors)
int ORBmatcher::Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th)
{
    const Pose P(pKF->GetRotation(), pKF->GetTranslation());
    const Vec3 centre(pKF->GetCameraCenter());
    const int nMPs = (int)vpMapPoints.size();
    // ---- phase 1: every gate that does not depend on what the loop mutates (:1060-1101); one query per point that passes ----
    PointList pl;
    for (int i = 0; i < nMPs; ++i)
        if (vpMapPoints[(size_t)i]) pl.Add(vpMapPoints[(size_t)i], i, true, true);
    pl.Project(P, NULL, centre.v, pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, (float)pKF->mnMinX, (float)pKF->mnMaxX, (float)pKF->mnMinY,
               (float)pKF->mnMaxY, ORBFE_PJ_SKIP_NEG_DEPTH);
    Queries qs;
    std::vector<int> slot((size_t)nMPs, -1);  // query index of MapPoint i, -1 = gated out
    // :1103 GetFeaturesInArea(u, v, radius), :1119 the level window, :1122-1145 the reprojection-error gate: on the device
    LevelQueries(pl, pKF, pKF->mvScaleFactors, th, 0, ORBFE_PROJ_CHI2_GATE, true, qs, &slot);
    // ---- the candidate scans: best candidate per point, first in list order on ties (:1149-1156), accepted at TH_LOW ----
    std::vector<int32_t> bestIdx;
    RunKeyFrameSearch(pKF, qs, TH_LOW, true, bestIdx, "Fuse");
    // ---- phase 2: the loop's decisions, in order, on the live objects (:1049-1056, :1159-1180) ----
    int fused = 0;
    for (int i = 0; i < nMPs; ++i) {
        MapPoint *pMP = vpMapPoints[(size_t)i];
        if (!pMP || slot[(size_t)i] < 0) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const int32_t at = bestIdx[(size_t)slot[(size_t)i]];
        if (at < 0) continue;  // :1159 bestDist <= TH_LOW
        MapPoint *held = pKF->GetMapPoint((size_t)at);
        if (!held) {
            pMP->AddObservation(pKF, (size_t)at);
            pKF->AddMapPoint(pMP, (size_t)at);
        } else if (!held->isBad()) {
            if (held->Observations() > pMP->Observations()) pMP->Replace(held);
            else held->Replace(pMP);
        }
        ++fused;
    }
    return fused;
}

// src/ORBmatcher.cc:1198-1299 (LoopClosing::SearchAndFuse): the Sim3 form -- no stereo gate, a hit on an occupied slot is
// reported in vpReplacePoint instead of replaced; the same two phases as the plain Fuse
int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
{
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint *> known = pKF->GetMapPoints();  // :1212, fixed for the whole loop
    const int nPoints = (int)vpPoints.size();
    PointList pl;
    for (int i = 0; i < nPoints; ++i) {
        MapPoint *p = vpPoints[(size_t)i];
        if (p && !p->isBad() && !known.count(p)) pl.Add(p, i, true, true);  // :1224 (neither changes inside the loop)
    }
    const Vec3 centre(Ow);
    pl.Project(Pose(Rcw, tcw), NULL, centre.v, pKF->fx, pKF->fy, pKF->cx, pKF->cy, 0.f, (float)pKF->mnMinX, (float)pKF->mnMaxX,
               (float)pKF->mnMinY, (float)pKF->mnMaxY, ORBFE_PJ_SKIP_NEG_DEPTH);
    Queries qs;
    std::vector<int> slot((size_t)nPoints, -1);
    // :1257 GetFeaturesInArea(u, v, radius) and the level window of :1268 run on the device
    LevelQueries(pl, pKF, pKF->mvScaleFactors, th, 0, 0, false, qs, &slot);
    std::vector<int32_t> bestIdx;
    RunKeyFrameSearch(pKF, qs, TH_LOW, false, bestIdx, "Fuse");
    int fused = 0;
    for (int i = 0; i < nPoints; ++i) {  // :1282-1296 on the live objects, in order
        if (slot[(size_t)i] < 0) continue;
        const int32_t at = bestIdx[(size_t)slot[(size_t)i]];
        if (at < 0) continue;  // :1281 bestDist <= TH_LOW
        MapPoint *held = pKF->GetMapPoint((size_t)at);
        if (!held) {
            vpPoints[(size_t)i]->AddObservation(pKF, (size_t)at);
            pKF->AddMapPoint(vpPoints[(size_t)i], (size_t)at);
        } else if (!held->isBad()) {
            vpReplacePoint[(size_t)i] = held;
        }
        ++fused;
    }
    return fused;
}

// src/ORBmatcher.cc:1334-1516 (LoopClosing::ComputeSim3): the points of each keyframe are searched in the other one under the
// candidate similarity, a match is kept when both directions agree.  No decision depends on an earlier one: each direction
// is one projection call + one device call (window search on the other keyframe's grid + Hamming).
int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                             const cv::Mat &t12, const float th)
{
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const Pose W1(pKF1->GetRotation(), pKF1->GetTranslation()), W2(pKF2->GetRotation(), pKF2->GetTranslation());
    const Pose S21(sR21, t21), S12(sR12, t12);
    const std::vector<MapPoint *> pts1 = pKF1->GetMapPointMatches(), pts2 = pKF2->GetMapPointMatches();
    const int N1 = (int)pts1.size(), N2 = (int)pts2.size();
    std::vector<uint8_t> done1((size_t)N1, 0), done2((size_t)N2, 0);
    for (int i = 0; i < N1; ++i) {  // :1360-1371
        MapPoint *p = vpMatches12[(size_t)i];
        if (!p) continue;
        done1[(size_t)i] = 1;
        const int at2 = p->GetIndexInKeyFrame(pKF2);
        if (at2 >= 0 && at2 < N2) done2[(size_t)at2] = 1;
    }
    // one direction: the points of keyframe A (world -> A by Wa) are taken into keyframe B by the similarity Sba and looked up there
    auto search = [&](const std::vector<MapPoint *> &pts, const std::vector<uint8_t> &done, const Pose &Wa, const Pose &Sba, KeyFrame *pKFb,
                      std::vector<int> &found_at) {
        PointList pl;
        for (size_t i = 0; i < pts.size(); ++i)
            if (pts[i] && !done[i] && !pts[i]->isBad()) pl.Add(pts[i], (int)i, false, true);
        // the reference projects with keyframe 1's intrinsics in BOTH directions (:1337-1340, :1399, :1476)
        pl.Project(Wa, &Sba, NULL, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, 0.f, (float)pKFb->mnMinX, (float)pKFb->mnMaxX, (float)pKFb->mnMinY,
                   (float)pKFb->mnMaxY, ORBFE_PJ_SKIP_NEG_DEPTH);
        Queries qs;
        LevelQueries(pl, pKFb, pKFb->mvScaleFactors, th, 0, 0, false, qs, NULL);
        std::vector<int32_t> best;
        RunKeyFrameSearch(pKFb, qs, TH_HIGH, false, best, "SearchBySim3");
        for (size_t k = 0; k < best.size(); ++k)
            if (best[k] >= 0) found_at[(size_t)qs.src[k]] = best[k];  // bestDist <= TH_HIGH (:1451, :1527)
    };
    std::vector<int> in2((size_t)N1, -1), in1((size_t)N2, -1);
    search(pts1, done1, W1, S21, pKF2, in2);  // :1380-1453
    search(pts2, done2, W2, S12, pKF1, in1);  // :1455-1529
    int agreed = 0;
    for (int i = 0; i < N1; ++i) {  // :1532-1545 both directions have to agree
        const int j = in2[(size_t)i];
        if (j >= 0 && in1[(size_t)j] == i) {
            vpMatches12[(size_t)i] = pts2[(size_t)j];
            ++agreed;
        }
    }
    return agreed;
}

// src/ORBmatcher.cc:523-651 (Tracking::MonocularInitialization).  A candidate is skipped when an earlier query already holds
// it at a distance <= its own (:573), so best / second-best of a query depend on the matches made before it: the device
// returns every window's candidates with their distances (orbfe_window_distances, one call), orbfe_initialization_resolve runs
// the in-order rule on those lists, and the outcome is written back here.
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize)
{
    const size_t n1 = F1.mvKeysUn.size(), n2 = F2.mvKeysUn.size();
    vnMatches12.assign(n1, -1);
    // one query per level-0 feature of F1 (:543-545): GetFeaturesInArea(prev.x, prev.y, windowSize, 0, 0) on F2's grid
    std::vector<orbfe_proj_query> qs;
    std::vector<uint8_t> qdesc, t1;
    std::vector<int> feat_of;  // F1 feature of a query
    const uint8_t *d1 = Rows(F1.mDescriptors, t1);
    for (size_t i = 0; i < n1; ++i) {
        if (F1.mvKeysUn[i].octave > 0) continue;
        orbfe_proj_query e;
        e.u = vbPrevMatched[i].x; e.v = vbPrevMatched[i].y; e.r = (float)windowSize;
        e.min_level = 0; e.max_level = 0; e.ur = 0.f; e.flags = 0; e.pad = 0;
        feat_of.push_back((int)i);
        qs.push_back(e);
        qdesc.insert(qdesc.end(), d1 + i * 32, d1 + i * 32 + 32);
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
    std::vector<int32_t> accepted((size_t)std::max(nq, 1), -1), holder(std::max<size_t>(n2, 1), -1);
    if (orbfe_initialization_resolve(off.data(), ent.data(), nq, (int32_t)n2, TH_LOW, mfNNratio, accepted.data(), holder.data()) != ORBFE_OK)
        throw std::runtime_error(std::string("ORBmatcher::SearchForInitialization (orbfe): ") + orbfe_last_error());
    // a query's match stands while it is still the holder of its feature (:587-592 a later, closer query takes it over)
    int found = 0;
    std::vector<float> a1, a2;
    std::vector<int> voter;  // F1 features in acceptance order: every accepted match votes (:600-608), taken over later or not
    for (int k = 0; k < nq; ++k) {
        const int j = accepted[(size_t)k];
        if (j < 0) continue;
        const size_t i = (size_t)feat_of[(size_t)k];
        if (holder[(size_t)j] == k) {
            vnMatches12[i] = j;
            ++found;
        }
        voter.push_back((int)i);
        a1.push_back(F1.mvKeysUn[i].angle);
        a2.push_back(F2.mvKeysUn[(size_t)j].angle);
    }
    if (mbCheckOrientation) {  // :621-641
        std::vector<uint8_t> drop;
        RotationOutliers(a1, a2, HISTO_LENGTH, drop);
        for (size_t e = 0; e < drop.size(); ++e)
            if (drop[e] && vnMatches12[(size_t)voter[e]] >= 0) {
                vnMatches12[(size_t)voter[e]] = -1;
                --found;
            }
    }
    for (size_t i = 0; i < n1; ++i)  // :644-647
        if (vnMatches12[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[(size_t)vnMatches12[i]].pt;
    return found;
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
