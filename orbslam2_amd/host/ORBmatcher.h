// ORBmatcher.h -- drop-in for ORB_SLAM2::ORBmatcher (reference ORBmatcher.h:39-257) whose search methods run
// on MI355X through the C ABI of include/orbmatch.h.  Same class name, namespace, constructor, method
// signatures, public constants and return values as the reference; callers (Tracking.cc:768-1879,
// LocalMapping.cc:416, LoopClosing.cc:417-659) compile unchanged.
//
// Each method snapshots the pointer graph into flat arrays (descriptors, keypoints, FeatureVector CSR, map
// point flags), calls ONE batched entry point of liborbmatch_hip.so and re-applies the pointer side effects
// the reference makes (F.mvpMapPoints[...] = pMP, vpMatched[...] = pMP, vbPrevMatched, ...).  The snapshot
// equals what the reference reads because its callers hold Map::mMutexMapUpdate or work on local copies
// ("Pointer graph <-> flat arrays").
//
// Fuse mutates the map while it iterates (MapPoint::Replace / AddObservation).  The search itself reads none of
// that state, so it runs as one launch; the host then replays the reference's loop in order with the gates that
// do depend on it (isBad, IsInKeyFrame, the slot's current map point) evaluated at replay time.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <atomic>
#include <cstdint>
#include <set>
#include <utility>
#include <vector>

// In an ORB-SLAM2 tree (built with -DORBMATCH_USE_OPENCV) the types come from the project's own headers, exactly as
// the reference's ORBmatcher.h:29-31 includes them; stand-alone builds and the tests use the stand-ins.
#ifdef ORBMATCH_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#else
#include "orbslam_types.h"
#endif

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Hamming distance of two 32-byte descriptors (ORBmatcher.cc:2184-2200).  A scalar host helper for the
    // callers outside the matcher (Frame.cc:720, MapPoint.cc:418); the searches below never call it.
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // ORBmatcher.cc:57-166 (Tracking::SearchLocalPoints)
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3);
    // ORBmatcher.cc:1768-1950 (Tracking::TrackWithMotionModel)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);
    // ORBmatcher.cc:1963-2123 (Tracking::Relocalization)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    // ORBmatcher.cc:401-542 (LoopClosing::ComputeSim3)
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);

    // ORBmatcher.cc:1468-1752 (LoopClosing::ComputeSim3)
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12,
                     const cv::Mat &t12, const float th);
    // ORBmatcher.cc:1111-1288 (LocalMapping::SearchInNeighbors)
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    // ORBmatcher.cc:1301-1443 (LoopClosing::SearchAndFuse)
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

    // ORBmatcher.cc:218-385 and 703-854
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);

    // ORBmatcher.cc:562-689
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);

    // ORBmatcher.cc:868-1093
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    // ---- extensions (not in the reference) -----------------------------------------------------------------------
    // GPU the searches run on (contexts created after the call); default 0 or $ORBMATCH_DEVICE.
    static void SetDevice(int device);
    // Several GPUs behind the batched SearchByBoW overloads below (the reference's candidate loops LoopClosing.cc:433-477 /
    // Tracking.cc:1778-1801 over shards): the candidate list is cut into one contiguous slice per listed device, every slice
    // runs on a worker thread with the calling thread's own context and residency table for that device, and the results
    // are put together on the host -- pairs are independent, no collective is involved.  The same device may be listed
    // more than once (two contexts on one GPU).  Call it before the calling thread's first search; an empty list or a list
    // of one restores the single-context behaviour.
    static void SetDevices(const std::vector<int> &devices);

    // Device residency.  Every Frame / KeyFrame a thread searches is uploaded ONCE into that thread's context
    // (descriptors, keypoints, FeatureVector; the grid is built on the device) and found again by
    // (mnId, descriptor buffer, N) + a content fingerprint; later calls only refresh the map-point flags when they
    // changed.  Prefetch uploads ahead of the first search -- the natural place is keyframe creation, after
    // ComputeBoW (KeyFrame.cc:46-78, LocalMapping.cc:ProcessNewKeyFrame).  Evict drops the calling thread's copy
    // (call it from KeyFrame::SetBadFlag / before deleting a keyframe; copies are also dropped least-recently-used
    // first beyond the limits, default 16384 keyframes and 8 frames per thread).
    static void Prefetch(KeyFrame *pKF);
    static void Evict(KeyFrame *pKF);
    static void Evict(const Frame &F);
    static void SetResidentLimits(size_t nKeyFrames, size_t nFrames);
    // How a resident copy is checked against the host object on every call.  0: by key only (mnId, descriptor buffer, N);
    // 1 (default): + a fingerprint of three sampled rows / keypoints; 2: + a hash of EVERY descriptor row and keypoint
    // (taken at upload, recomputed per call: ~6 us per 2000-keypoint frame) -- for host projects that rewrite
    // mDescriptors / mvKeysUn in place, which ORB-SLAM2 itself never does (Frame.h:235-253 are written by the constructor).
    static void SetResidencyVerify(int level);
    // Optional: a counter the host project bumps whenever a MapPoint's bad flag or observation count, or any Frame's /
    // KeyFrame's map-point vector, changes.  While it stands still the per-call sweep over the N map points of a resident
    // frame (isBad() + Observations(), both mutex-guarded: MapPoint.cc:212,315) is skipped.  nullptr (default): the
    // reference-exact sweep on every call.
    static void SetMapPointVersion(const std::atomic<uint64_t> *counter);

    // The candidate loops around SearchByBoW as ONE launch: Tracking::Relocalization (Tracking.cc:1778-1801,
    // `matcher.SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i])` for every candidate) and
    // LoopClosing::ComputeSim3 (LoopClosing.cc:433-477, `matcher.SearchByBoW(mpCurrentKF, pKF, vvpMapPointMatches[i])`).
    // Entry i of the result / of vvp... is exactly what the i-th call of the loop returns; a NULL or bad candidate
    // (the loops `continue` on isBad()) gets 0 and an empty vector.
    std::vector<int> SearchByBoW(const std::vector<KeyFrame *> &vpKFs, Frame &F, std::vector<std::vector<MapPoint *>> &vvpMapPointMatches);
    std::vector<int> SearchByBoW(KeyFrame *pKF1, const std::vector<KeyFrame *> &vpKF2s, std::vector<std::vector<MapPoint *>> &vvpMatches12);

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

// Extension (not in the reference): MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:363-449) for many map points
// in one launch; sets mDescriptor of every point exactly as the per-point method would.  Use it where the reference
// loops over points (LocalMapping.cc:289,780,926, LoopClosing.cc:918).
void ComputeDistinctiveDescriptorsBatch(const std::vector<MapPoint *> &vpMapPoints);

}  // namespace ORB_SLAM2

#endif  // ORBMATCHER_H
