// ORBmatcher_orbfe.cc -- link-level drop-in for the Hamming core of the reference's ORBmatcher.
//
// The reference's OWN include/ORBmatcher.h stays as it is (so Tracking.cc, LocalMapping.cc, LoopClosing.cc and the
// projection family SearchByProjection x4 / SearchForInitialization / SearchForTriangulation / SearchBySim3 / Fuse x2
// in src/ORBmatcher.cc compile unchanged).  This file supplies the three members that carry the Hamming work, over
// the C-ABI of liborbfe.so:
//     static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)              src/ORBmatcher.cc:1968-1984
//     int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&)                src/ORBmatcher.cc:217-363
//     int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&)             src/ORBmatcher.cc:665-812
// Integration: add this file to the ORB_SLAM2 library, delete (or #if 0) those three bodies in src/ORBmatcher.cc, link
// liborbfe.so.  The class gets no new data member: the device matcher handle is per thread, which is also
// what the C-ABI asks for (Tracking, LocalMapping and LoopClosing call the matcher concurrently).
//
// Built and tested in this repo against the reference's unmodified header with mock KeyFrame / Frame / MapPoint types
// (oracle/refbuild: libshim_ref.so; tests/test_gpu_shim_ref.py compares it with the compiled reference bodies).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <string>
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
}  // namespace

namespace ORB_SLAM2
{

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
