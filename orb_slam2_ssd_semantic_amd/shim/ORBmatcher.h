// ORBmatcher.h -- drop-in core of the reference's include/ORBmatcher.h:33-134 over the C-ABI.
// DescriptorDistance, the constants, the constructor and both SearchByBoW overloads keep their signatures.  The
// SearchByBoW bodies are templates over the reference's KeyFrame / Frame / MapPoint types so that this header can be
// unit-tested with mock types here (DBoW2 / the SLAM classes are not in this repo) and instantiated with the real
// ones inside the ORB-SLAM2 tree.  The geometry-gated family (SearchByProjection, Fuse, ...) stays
// the reference's own code; it can call HammingCSR() for its inner loops.
#pragma once
#include <stdint.h>

#include <vector>

#ifdef ORBFE_WITH_OPENCV
#include <opencv2/core/core.hpp>
#else
#include "cv_stub/orbfe_cv_stub.h"
#endif
#include "orbfe.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    ~ORBmatcher() { orbfe_matcher_destroy(mpMatcher); }

    // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1968-1984)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbfe_hamming(a.ptr<uint8_t>(0), b.ptr<uint8_t>(0)); }

    // SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)   src/ORBmatcher.cc:217-363
    template <class KeyFrameT, class FrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches)
    {
        const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT *>(F.N, static_cast<MapPointT *>(NULL));
        std::vector<uint8_t> validKF(vpMapPointsKF.size());
        for (size_t i = 0; i < validKF.size(); ++i) validKF[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
        Csr kf, f;
        Flatten(pKF->mFeatVec, kf);
        Flatten(F.mFeatVec, f);
        std::vector<float> angKF, angF;
        Angles(pKF->mvKeysUn, angKF);
        Angles(F.mvKeys, angF);
        std::vector<int32_t> m(F.N, -1);
        int n = 0;
        if (!Ready()) return 0;
        mLastStatus = orbfe_search_by_bow(mpMatcher, pKF->mDescriptors.template ptr<uint8_t>(0), (int)validKF.size(),
                                          validKF.data(), angKF.data(), kf.node.data(), kf.off.data(), kf.idx.data(),
                                          (int)kf.node.size(), F.mDescriptors.template ptr<uint8_t>(0), F.N, NULL,
                                          angF.data(), f.node.data(), f.off.data(), f.idx.data(), (int)f.node.size(),
                                          mfNNratio, TH_LOW, 0, mbCheckOrientation ? 1 : 0, m.data(), &n);
        if (mLastStatus != ORBFE_OK) return 0;
        for (int i = 0; i < F.N; ++i)
            if (m[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[m[i]];
        return n;
    }

    // SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)   src/ORBmatcher.cc:665-812
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12)
    {
        const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches();
        const std::vector<MapPointT *> vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPointT *>(vpMapPoints1.size(), static_cast<MapPointT *>(NULL));
        std::vector<uint8_t> v1(vpMapPoints1.size()), v2(vpMapPoints2.size());
        for (size_t i = 0; i < v1.size(); ++i) v1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
        for (size_t i = 0; i < v2.size(); ++i) v2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
        Csr c1, c2;
        Flatten(pKF1->mFeatVec, c1);
        Flatten(pKF2->mFeatVec, c2);
        std::vector<float> a1, a2;
        Angles(pKF1->mvKeysUn, a1);
        Angles(pKF2->mvKeysUn, a2);
        std::vector<int32_t> m(v2.size(), -1);
        int n = 0;
        if (!Ready()) return 0;
        mLastStatus = orbfe_search_by_bow(mpMatcher, pKF1->mDescriptors.template ptr<uint8_t>(0), (int)v1.size(), v1.data(),
                                          a1.data(), c1.node.data(), c1.off.data(), c1.idx.data(), (int)c1.node.size(),
                                          pKF2->mDescriptors.template ptr<uint8_t>(0), (int)v2.size(), v2.data(), a2.data(),
                                          c2.node.data(), c2.off.data(), c2.idx.data(), (int)c2.node.size(), mfNNratio,
                                          TH_LOW, 1, mbCheckOrientation ? 1 : 0, m.data(), &n);
        if (mLastStatus != ORBFE_OK) return 0;
        for (size_t i2 = 0; i2 < v2.size(); ++i2)  // reference output is indexed by KF1 feature (:751)
            if (m[i2] >= 0) vpMatches12[m[i2]] = vpMapPoints2[i2];
        return n;
    }

    // BASELINE config 3: all-pairs best / second best + ratio + rotation histogram
    int MatchBruteForce(const cv::Mat &descQ, const std::vector<cv::KeyPoint> &kpQ, const cv::Mat &descT,
                        const std::vector<cv::KeyPoint> &kpT, std::vector<int> &vnMatchesQ2T, int th = TH_HIGH)
    {
        std::vector<float> aq, at;
        Angles(kpQ, aq);
        Angles(kpT, at);
        vnMatchesQ2T.assign(descQ.rows, -1);
        int n = 0;
        if (!Ready()) return 0;
        mLastStatus = orbfe_match_bf(mpMatcher, descQ.ptr<uint8_t>(0), descQ.rows, descT.ptr<uint8_t>(0), descT.rows,
                                     aq.data(), at.data(), mfNNratio, th, mbCheckOrientation ? 1 : 0, vnMatchesQ2T.data(),
                                     NULL, NULL, &n);
        return mLastStatus == ORBFE_OK ? n : 0;
    }

    // inner loop of the SearchByProjection / Fuse / SearchForTriangulation family
    int HammingCSR(const cv::Mat &descQ, const cv::Mat &descT, const std::vector<uint32_t> &off,
                   const std::vector<uint32_t> &cand, std::vector<int> &bestIdx, std::vector<int> &best,
                   std::vector<int> &second)
    {
        bestIdx.assign(descQ.rows, -1);
        best.assign(descQ.rows, 256);
        second.assign(descQ.rows, 256);
        if (!Ready()) return ORBFE_ERR_NODEVICE;
        mLastStatus = orbfe_hamming_csr(mpMatcher, descQ.ptr<uint8_t>(0), descQ.rows, descT.ptr<uint8_t>(0), descT.rows,
                                        off.data(), cand.data(), bestIdx.data(), best.data(), second.data());
        return mLastStatus;
    }

    int LastStatus() const { return mLastStatus; }

public:
    static const int TH_LOW = 50;        // src/ORBmatcher.cc:40
    static const int TH_HIGH = 100;      // src/ORBmatcher.cc:39
    static const int HISTO_LENGTH = 30;  // src/ORBmatcher.cc:41

protected:
    struct Csr {
        std::vector<uint32_t> node, off, idx;
    };
    template <class FeatureVectorT> static void Flatten(const FeatureVectorT &fv, Csr &c)
    {
        c.off.push_back(0);
        for (typename FeatureVectorT::const_iterator it = fv.begin(); it != fv.end(); ++it) {  // std::map: ascending ids
            c.node.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); ++k) c.idx.push_back((uint32_t)it->second[k]);
            c.off.push_back((uint32_t)c.idx.size());
        }
    }
    static void Angles(const std::vector<cv::KeyPoint> &k, std::vector<float> &a)
    {
        a.resize(k.size());
        for (size_t i = 0; i < k.size(); ++i) a[i] = k[i].angle;
    }
    bool Ready()
    {
        if (mpMatcher) return true;
        mLastStatus = orbfe_matcher_create(-1, &mpMatcher);
        return mLastStatus == ORBFE_OK;
    }

    float mfNNratio;
    bool mbCheckOrientation;
    orbfe_matcher *mpMatcher = nullptr;
    int mLastStatus = 0;
};

}  // namespace ORB_SLAM2
