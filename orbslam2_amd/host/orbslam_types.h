// orbslam_types.h -- the members of Frame / KeyFrame / MapPoint that ORBmatcher reads and writes, with the
// reference's names (Frame.h:235-291, KeyFrame.h:131-282, MapPoint.h:96-198).  In an ORB-SLAM2 tree these
// come from the project's own Frame.h / KeyFrame.h / MapPoint.h; this header exists so that the drop-in
// ORBmatcher can be compiled and tested without the rest of the system (its constructors, extractor,
// vocabulary, covisibility graph and mutexes are outside the hot path and are not mirrored).
#pragma once
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "cvlite.h"

namespace DBoW2 {
// FeatureVector.h of DBoW2: node id -> indices of the local features in that node
class FeatureVector : public std::map<unsigned int, std::vector<unsigned int>> {};

typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;

enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };

// BowVector.h of DBoW2: word id -> value (methods in ORBVocabulary.cc)
class BowVector : public std::map<WordId, WordValue> {
public:
    void addWeight(WordId id, WordValue v);
    void addIfNotExist(WordId id, WordValue v);
    void normalize(LNorm norm_type);
};
}  // namespace DBoW2

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;

class MapPoint {
public:
    // --- the interface the matcher calls (MapPoint.cc:115-127, 212, 315, 453, 541-602) ---
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, KeyFrame *pKF);
    int PredictScale(const float &currentDist, Frame *pF);
    int GetIndexInKeyFrame(KeyFrame *pKF) {
        auto it = mObservations.find(pKF);
        return it == mObservations.end() ? -1 : (int)it->second;
    }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    // the two mutators Fuse calls (MapPoint.cc:141-158, 255-311); defined after KeyFrame below
    void AddObservation(KeyFrame *pKF, size_t idx);
    void Replace(MapPoint *pMP);
    MapPoint *GetReplaced() { return mpReplaced; }
    void IncreaseVisible(int n = 1) { mnVisible += n; }
    void IncreaseFound(int n = 1) { mnFound += n; }
    // MapPoint.cc:323-398 (median-distance descriptor over all observations) belongs to the map, not the matcher:
    // the stand-in keeps mDescriptor as it is.
    void ComputeDistinctiveDescriptors() {}

    MapPoint() { static long unsigned int nNextId = 0; mnId = nNextId++; }
    long unsigned int mnId;

    // tracking scratch fields written by Frame::isInFrustum (MapPoint.h:183-198)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;

    // state
    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    int nObs = 0;
    bool mbBad = false;
    float mfMinDistance = 0, mfMaxDistance = 0;
    int mnVisible = 1, mnFound = 1;
    MapPoint *mpReplaced = nullptr;
};

class Frame {
public:
    long unsigned int mnId = 0;
    DBoW2::BowVector mBowVec;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    float mbf = 0, mb = 0;
    int mnScaleLevels = 8;
    float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    // statics, as in Frame.h:203-206, 265-266, 295-298
    static float fx, fy, cx, cy;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
};

class KeyFrame {
public:
    long unsigned int mnId = 0;
    DBoW2::BowVector mBowVec;
    // KeyFrameDatabase bookkeeping (KeyFrame.h:152-160)
    long unsigned int mnLoopQuery = 0;
    int mnLoopWords = 0;
    float mLoopScore = 0;
    long unsigned int mnRelocQuery = 0;
    int mnRelocWords = 0;
    float mRelocScore = 0;
    // covisibility graph as the database reads it (KeyFrame.cc:193-236): the stand-in stores the ordered list directly
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    std::set<KeyFrame *> GetConnectedKeyFrames() { return std::set<KeyFrame *>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.end()); }
    std::vector<KeyFrame *> GetBestCovisibilityKeyFrames(const int &N_) {
        if ((int)mvpOrderedConnectedKeyFrames.size() < N_) return mvpOrderedConnectedKeyFrames;
        return std::vector<KeyFrame *>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + N_);
    }
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int mnScaleLevels = 8;
    float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;  // const int in KeyFrame.h:251-254 (truncated from Frame's floats)
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;

    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints() {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }                    // KeyFrame.cc:285
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = static_cast<MapPoint *>(NULL); }  // KeyFrame.cc:294
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }           // KeyFrame.cc:309
    cv::Mat GetRotation() { return Rcw.clone(); }       // 3x3 CV_32F
    cv::Mat GetTranslation() { return tcw.clone(); }    // 3x1
    cv::Mat GetCameraCenter() { return Ow.clone(); }    // 3x1
    bool isBad() { return mbBad; }                          // KeyFrame.cc:703
    bool mbBad = false;
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

    std::vector<MapPoint *> mvpMapPoints;
    cv::Mat Rcw, tcw, Ow;
};

// MapPoint.cc:141-158
inline void MapPoint::AddObservation(KeyFrame *pKF, size_t idx) {
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    if (pKF->mvuRight[idx] >= 0) nObs += 2;
    else nObs++;
}

// MapPoint.cc:255-311 (without the Map bookkeeping of :310)
inline void MapPoint::Replace(MapPoint *pMP) {
    if (pMP->mnId == this->mnId) return;
    std::map<KeyFrame *, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    const int nvisible = mnVisible, nfound = mnFound;
    mpReplaced = pMP;
    for (auto mit = obs.begin(), mend = obs.end(); mit != mend; mit++) {
        KeyFrame *pKF = mit->first;
        if (!pMP->IsInKeyFrame(pKF)) {
            pKF->ReplaceMapPointMatch(mit->second, pMP);
            pMP->AddObservation(pKF, mit->second);
        } else {
            pKF->EraseMapPointMatch(mit->second);
        }
    }
    pMP->IncreaseFound(nfound);
    pMP->IncreaseVisible(nvisible);
    pMP->ComputeDistinctiveDescriptors();
}

// MapPoint.cc:569-602.  `using namespace std` reaches MapPoint.cc through the DBoW2 headers, so log(float)
// and ceil(float) are the float overloads.
inline int MapPoint::PredictScale(const float &currentDist, KeyFrame *pKF) {
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
    return nScale;
}
inline int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}

}  // namespace ORB_SLAM2
