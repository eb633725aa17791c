/*
 * ref_mocks.h -- mock MapPoint / KeyFrame / Frame for compiling the UNMODIFIED /root/reference/src/ORBmatcher.cc
 * (TEST INFRASTRUCTURE for oracle/_ref, NOT PRODUCT CODE).
 *
 * The reference's own MapPoint.h / KeyFrame.h / Frame.h pull in Map, KeyFrameDatabase, ORBVocabulary (DBoW2
 * templates), g2o and Eigen, none of which exist here.  This header is force-included (-include) with the
 * reference's include guards pre-defined (-DMAPPOINT_H -DKEYFRAME_H -DFRAME_H), so `#include "MapPoint.h"` etc.
 * in include/ORBmatcher.h become empty and the matcher sees these plain data holders instead.  They declare the
 * members ORBmatcher.cc touches, with the reference's names, types and signatures (include/Frame.h:63-190,
 * include/KeyFrame.h:43-197, include/MapPoint.h:46-108); no behaviour of the matcher is restated here.
 */
#ifndef ORBFE_REF_MOCKS_H
#define ORBFE_REF_MOCKS_H

#include <map>
#include <mutex>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>

#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "ORBextractor.h" /* the reference's own header (compiles against the cv stub): Frame holds two of them */

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

/* The members below that are `protected:` in the reference's real headers (include/KeyFrame.h:203-241, include/MapPoint.h:
 * 124-150) are declared behind REF_MOCKS_PROTECTED.  The glue files need them public (they build the mock objects); the
 * PRODUCT's matcher shim is compiled with -DREF_MOCKS_STRICT, where they ARE protected -- a compile-time check that the shim
 * reaches the objects only through what the real classes make public. */
#ifdef REF_MOCKS_STRICT
#define REF_MOCKS_PROTECTED protected
#else
#define REF_MOCKS_PROTECTED public
#endif

using std::pair; /* include/ORBmatcher.h:82 writes std::vector<pair<size_t,size_t> > without std:: */
using std::vector;

namespace ORB_SLAM2
{
class KeyFrame;
class Frame;
class ORBVocabulary; /* only ever held as a pointer by the code compiled here */

class MapPoint
{
  public:
    MapPoint()
        : mTrackProjX(0), mTrackProjY(0), mTrackProjXR(0), mbTrackInView(false), mnTrackScaleLevel(0),
          mTrackViewCos(0), mnLastFrameSeen(0), mnFuseCandidateForKF(0), mbBad(false), nObs(0), mfMinDistance(0),
          mfMaxDistance(1e9f), replaced(0)
    {
    }
    MapPoint(const MapPoint &o)
        : mTrackProjX(o.mTrackProjX), mTrackProjY(o.mTrackProjY), mTrackProjXR(o.mTrackProjXR),
          mbTrackInView(o.mbTrackInView), mnTrackScaleLevel(o.mnTrackScaleLevel), mTrackViewCos(o.mTrackViewCos),
          mnLastFrameSeen(o.mnLastFrameSeen), mnFuseCandidateForKF(o.mnFuseCandidateForKF), mbBad(o.mbBad), nObs(o.nObs),
          mfMinDistance(o.mfMinDistance), mfMaxDistance(o.mfMaxDistance), replaced(o.replaced), world_pos(o.world_pos),
          normal(o.normal), mDescriptor(o.mDescriptor), mObservations(o.mObservations)
    {
    }
    MapPoint &operator=(const MapPoint &o) /* the mutexes are not state */
    {
        mTrackProjX = o.mTrackProjX; mTrackProjY = o.mTrackProjY; mTrackProjXR = o.mTrackProjXR;
        mbTrackInView = o.mbTrackInView; mnTrackScaleLevel = o.mnTrackScaleLevel; mTrackViewCos = o.mTrackViewCos;
        mnLastFrameSeen = o.mnLastFrameSeen; mnFuseCandidateForKF = o.mnFuseCandidateForKF; mbBad = o.mbBad; nObs = o.nObs;
        mfMinDistance = o.mfMinDistance; mfMaxDistance = o.mfMaxDistance; replaced = o.replaced; world_pos = o.world_pos;
        normal = o.normal; mDescriptor = o.mDescriptor; mObservations = o.mObservations;
        return *this;
    }
    cv::Mat GetWorldPos() { return world_pos.clone(); }
    cv::Mat GetNormal() { return normal.clone(); }
    int Observations() { return nObs; }
    void AddObservation(KeyFrame *pKF, size_t idx)
    {
        if (!mObservations.count(pKF)) nObs++;
        mObservations[pKF] = idx;
    }
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    bool isBad() { return mbBad; }
    void Replace(MapPoint *pMP) { replaced = pMP; mbBad = true; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    /* bodies sliced verbatim from src/MapPoint.cc by oracle/refbuild/slice.py (ref_slices.cpp) */
    int PredictScale(const float &currentDist, KeyFrame *pKF);
    int PredictScale(const float &currentDist, Frame *pF);
    void ComputeDistinctiveDescriptors();

    float mTrackProjX, mTrackProjY, mTrackProjXR;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;
    long unsigned int mnLastFrameSeen;
    long unsigned int mnFuseCandidateForKF;

    /* state under the reference's member names (include/MapPoint.h:118-153) where the sliced bodies touch it */
  REF_MOCKS_PROTECTED:
    bool mbBad;
    int nObs;
    float mfMinDistance, mfMaxDistance;
    MapPoint *replaced;
    cv::Mat world_pos, normal, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    std::mutex mMutexFeatures, mMutexPos;
  public:
};

class Frame
{
  public:
    Frame()
        : mpORBextractorLeft(0), mpORBextractorRight(0), fx(1), fy(1), cx(0), cy(0), mbf(0), mb(0), N(0), mnScaleLevels(8),
          mfLogScaleFactor(0.18232156f)
    {
    }
    /* bodies sliced verbatim from src/Frame.cc by oracle/refbuild/slice.py (ref_slices.cpp) */
    vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1,
                                     const int maxLevel = -1) const;
    bool PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY);
    void AssignFeaturesToGrid();
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    void ComputeStereoMatches(); /* src/Frame.cc:642-846, sliced */
    /* the stereo constructor and what it calls (src/Frame.cc:102-168, 337-343, 559-590, 593-626), sliced into
     * frame_stereo_api.cpp.  The reference reads `mb` in ComputeStereoMatches (:682) BEFORE the constructor assigns it
     * (:161), i.e. whatever the memory held; here that value is the default member initialiser below. */
    Frame(const cv::Mat &imLeft, const cv::Mat &imRight, const double &timeStamp, ORBextractor *extractorLeft,
          ORBextractor *extractorRight, ORBVocabulary *voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);
    /* the RGB-D constructor (src/Frame.cc:176-245; BASELINE configs 1-3), the monocular one (:247-311), the depth look-up
     * they call (:850-874) and perfect/'s RGB-D constructor with the dynamic-object mask (perfect/src/Frame.cc:328-420), all
     * sliced into frame_stereo_api.cpp */
    Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const double &timeStamp, ORBextractor *extractor, ORBVocabulary *voc,
          cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);
    Frame(const cv::Mat &imGray, const double &timeStamp, ORBextractor *extractor, ORBVocabulary *voc, cv::Mat &K,
          cv::Mat &distCoef, const float &bf, const float &thDepth);
    Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const cv::Mat &imMask, const double &timeStamp, ORBextractor *extractor,
          ORBVocabulary *voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth);
    void ComputeStereoFromRGBD(const cv::Mat &imDepth);
    cv::Mat mImDepth, mImDynm_mask; /* perfect/include/Frame.h:164-166 */
    void ExtractORB(int flag, const cv::Mat &im);
    void UndistortKeyPoints();
    void ComputeImageBounds(const cv::Mat &imLeft);
    static float s_mb_before_ctor;
    ORBVocabulary *mpORBvocabulary = 0;
    double mTimeStamp = 0;
    cv::Mat mK, mDistCoef;
    float mThDepth = 0;
    KeyFrame *mpReferenceKF = 0;
    long unsigned int mnId = 0;
    static long unsigned int nNextId;
    static bool mbInitialComputations;
    float mfScaleFactor = 1.2f;
    float invfx = 1, invfy = 1;
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    std::vector<cv::KeyPoint> mvKeysRight;
    cv::Mat mDescriptorsRight;
    float fx = 1, fy = 1, cx = 0, cy = 0;
    float mbf = 0, mb = s_mb_before_ctor;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0.18232156f;
    vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};

class KeyFrame
{
  public:
    KeyFrame()
        : fx(1), fy(1), cx(0), cy(0), mbf(0), mb(0), N(0), mnScaleLevels(8), mfLogScaleFactor(0.18232156f), mnMinX(0),
          mnMinY(0), mnMaxX(640), mnMaxY(480), mnId(0), mnGridCols(FRAME_GRID_COLS), mnGridRows(FRAME_GRID_ROWS),
          mfGridElementWidthInv(0.1f), mfGridElementHeightInv(0.1f), mbBad(false)
    {
    }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints()
    {
        std::set<MapPoint *> s;
        for (size_t i = 0; i < mvpMapPoints.size(); i++)
            if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
        return s;
    }
    /* body sliced verbatim from src/KeyFrame.cc (ref_slices.cpp) */
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r) const;
    bool isBad() { return mbBad; }
    bool IsInImage(const float &x, const float &y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }

    float fx, fy, cx, cy, mbf, mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    int mnScaleLevels;
    float mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnMinX, mnMinY, mnMaxX, mnMaxY;
    long unsigned int mnId;
    int mnGridCols, mnGridRows;
    float mfGridElementWidthInv, mfGridElementHeightInv;
  REF_MOCKS_PROTECTED:
    std::vector<std::vector<std::vector<size_t> > > mGrid;
    bool mbBad;

    /* mock state */
    std::vector<MapPoint *> mvpMapPoints;
    cv::Mat Ow, Rcw, tcw;
  public:
};

} // namespace ORB_SLAM2
#endif
