// KeyFrameDatabase.h -- drop-in for ORB_SLAM2::KeyFrameDatabase (KeyFrameDatabase.h:42-75): same constructor and public
// methods.  The inverted file and the list logic are the reference's (pointer work, kept on the host in its order);
// the `mpVoc->score(...)` loop over the keyframes that share enough words (KeyFrameDataBase.cc:179-196, 347-361) is
// ONE batched call on the GPU (orbm_bow_score_batch): each keyframe's BowVector is uploaded once, at add().
// The reference calls add/erase from LocalMapping and Detect* from LoopClosing / Tracking; so does this class: the
// device BowVectors live in a GPU context owned by the database (created at the first add), every use of it is under
// mMutex, and the destructor releases it.
#pragma once
#include <list>
#include <map>
#include <mutex>
#include <vector>

#include "ORBVocabulary.h"
#include "orbslam_types.h"

namespace ORB_SLAM2 {

class KeyFrameDatabase {
public:
    KeyFrameDatabase(const ORBVocabulary &voc);
    ~KeyFrameDatabase();

    void add(KeyFrame *pKF);
    void erase(KeyFrame *pKF);
    void clear();

    // Loop Detection
    std::vector<KeyFrame *> DetectLoopCandidates(KeyFrame *pKF, float minScore);
    // Relocalization
    std::vector<KeyFrame *> DetectRelocalizationCandidates(Frame *F);

protected:
    // float si = mpVoc->score(query, pKFi->mBowVec) for every listed keyframe, in one launch
    std::vector<float> ScoreAgainst(const DBoW2::BowVector &query, const std::vector<KeyFrame *> &kfs);

    const ORBVocabulary *mpVoc;
    std::vector<std::list<KeyFrame *>> mvInvertedFile;
    std::map<KeyFrame *, int> mDeviceBow;   // keyframe -> device BowVector id
    std::mutex mMutex;
    void *mCtx = nullptr;                   // orbm_ctx of this database (not a thread's): see the header comment
};

}  // namespace ORB_SLAM2
