// KeyFrameDatabase.cc -- see KeyFrameDatabase.h.  The score arithmetic happens on the GPU only: without the library the
// Detect* methods throw.
#include "KeyFrameDatabase.h"

#include <set>
#include <stdexcept>

#include "host_ctx.h"

using namespace std;

namespace ORB_SLAM2 {

namespace {
struct FlatBow {
    vector<uint32_t> word;
    vector<double> value;
    orbm_bowvec_view view;
    explicit FlatBow(const DBoW2::BowVector &v) {
        word.reserve(v.size()); value.reserve(v.size());
        for (const auto &kv : v) { word.push_back(kv.first); value.push_back(kv.second); }
        view.n = (int32_t)word.size(); view.word = word.data(); view.value = value.data();
    }
};
}  // namespace

KeyFrameDatabase::KeyFrameDatabase(const ORBVocabulary &voc) : mpVoc(&voc) {
    if (voc.getScoringType() != DBoW2::L1_NORM)
        throw runtime_error("KeyFrameDatabase: the GPU score is DBoW2's L1 scoring (ORBvoc.txt's type); this vocabulary uses another");
    mvInvertedFile.resize(voc.size());
}

KeyFrameDatabase::~KeyFrameDatabase() {
    if (mCtx) orbm_destroy(static_cast<orbm_ctx *>(mCtx));   // frees every device BowVector with it
}

// The database is shared between threads (add / erase from LocalMapping, DetectLoopCandidates from LoopClosing,
// DetectRelocalizationCandidates from Tracking: LocalMapping.cc:ProcessNewKeyFrame, LoopClosing.cc:DetectLoop,
// Tracking.cc:1738), so its device BowVectors live in a context of its own, used only under mMutex.
static orbm_ctx *DbCtx(void *&slot) {
    if (!slot) slot = detail::NewCtx();
    return static_cast<orbm_ctx *>(slot);
}

void KeyFrameDatabase::add(KeyFrame *pKF) {
    unique_lock<mutex> lock(mMutex);
    for (auto vit = pKF->mBowVec.begin(), vend = pKF->mBowVec.end(); vit != vend; vit++) mvInvertedFile[vit->first].push_back(pKF);
    if (!mDeviceBow.count(pKF)) {
        FlatBow fb(pKF->mBowVec);
        int32_t id = -1;
        orbm_ctx *ctx = DbCtx(mCtx);
        detail::CheckOn(ctx, orbm_bowvecs_upload(ctx, 1, &fb.view, &id), "orbm_bowvecs_upload");
        mDeviceBow[pKF] = id;
    }
}

void KeyFrameDatabase::erase(KeyFrame *pKF) {
    unique_lock<mutex> lock(mMutex);
    for (auto vit = pKF->mBowVec.begin(), vend = pKF->mBowVec.end(); vit != vend; vit++) {
        list<KeyFrame *> &lKFs = mvInvertedFile[vit->first];
        for (auto lit = lKFs.begin(), lend = lKFs.end(); lit != lend; lit++)
            if (pKF == *lit) {
                lKFs.erase(lit);
                break;
            }
    }
    auto it = mDeviceBow.find(pKF);
    if (it != mDeviceBow.end()) {
        int32_t id = it->second;
        orbm_ctx *ctx = DbCtx(mCtx);
        detail::CheckOn(ctx, orbm_bowvecs_free(ctx, 1, &id), "orbm_bowvecs_free");
        mDeviceBow.erase(it);
    }
}

void KeyFrameDatabase::clear() {
    unique_lock<mutex> lock(mMutex);
    mvInvertedFile.clear();
    mvInvertedFile.resize(mpVoc->size());
    vector<int32_t> ids;
    for (auto &kv : mDeviceBow) ids.push_back(kv.second);
    if (!ids.empty()) {
        orbm_ctx *ctx = DbCtx(mCtx);
        detail::CheckOn(ctx, orbm_bowvecs_free(ctx, (int)ids.size(), ids.data()), "orbm_bowvecs_free");
    }
    mDeviceBow.clear();
}

vector<float> KeyFrameDatabase::ScoreAgainst(const DBoW2::BowVector &query, const vector<KeyFrame *> &kfs) {
    vector<float> out(kfs.size());
    if (kfs.empty()) return out;
    orbm_ctx *ctx = DbCtx(mCtx);          // the callers hold mMutex
    FlatBow fq(query);
    int32_t qid = -1;
    detail::CheckOn(ctx, orbm_bowvecs_upload(ctx, 1, &fq.view, &qid), "orbm_bowvecs_upload");
    vector<int32_t> a(kfs.size(), qid), b(kfs.size());
    for (size_t i = 0; i < kfs.size(); i++) {
        auto it = mDeviceBow.find(kfs[i]);
        if (it == mDeviceBow.end()) { orbm_bowvecs_free(ctx, 1, &qid); throw runtime_error("KeyFrameDatabase: a keyframe of the inverted file has no device BowVector"); }
        b[i] = it->second;
    }
    vector<double> score(kfs.size());
    const int rc = orbm_bow_score_batch(ctx, (int)kfs.size(), a.data(), b.data(), score.data(), nullptr, 0);
    orbm_bowvecs_free(ctx, 1, &qid);
    detail::CheckOn(ctx, rc, "orbm_bow_score_batch");
    for (size_t i = 0; i < kfs.size(); i++) out[i] = (float)score[i];      // float si = mpVoc->score(...)
    return out;
}

namespace {

// The two queries differ only in which bookkeeping fields of KeyFrame they use, in whether keyframes connected to the
// query are left out, in the score floor and in how a covisible neighbour qualifies for the accumulation.
struct LoopFields {
    static long unsigned int &Query(KeyFrame *k) { return k->mnLoopQuery; }
    static int &Words(KeyFrame *k) { return k->mnLoopWords; }
    static float &Score(KeyFrame *k) { return k->mLoopScore; }
    // :216: a neighbour counts if this query reached it AND it is above the word threshold
    static bool NeighbourCounts(KeyFrame *k, long unsigned int id, int minWords) { return k->mnLoopQuery == id && k->mnLoopWords > minWords; }
};
struct RelocFields {
    static long unsigned int &Query(KeyFrame *k) { return k->mnRelocQuery; }
    static int &Words(KeyFrame *k) { return k->mnRelocWords; }
    static float &Score(KeyFrame *k) { return k->mRelocScore; }
    // :383: being reached by this query is enough -- a neighbour below the word threshold adds the score it LAST got
    static bool NeighbourCounts(KeyFrame *k, long unsigned int id, int) { return k->mnRelocQuery == id; }
};

// Walk of the inverted file in word order (:125-146, :305-325): keyframes in order of first encounter, with their
// shared-word counters.  A keyframe in `skip` never gets the query id, so its counter restarts at every encounter.
template <class Fld>
vector<KeyFrame *> SharingWords(const vector<list<KeyFrame *>> &inverted, const DBoW2::BowVector &bow, long unsigned int id,
                                const set<KeyFrame *> *skip) {
    vector<KeyFrame *> found;
    for (const auto &entry : bow)
        for (KeyFrame *k : inverted[entry.first]) {
            if (Fld::Query(k) != id) {
                Fld::Words(k) = 0;
                if (!skip || !skip->count(k)) {
                    Fld::Query(k) = id;
                    found.push_back(k);
                }
            }
            Fld::Words(k)++;
        }
    return found;
}

// Covisibility accumulation and the 0.75 cut (:205-262, :370-431)
template <class Fld>
vector<KeyFrame *> BestGroups(const vector<pair<float, KeyFrame *>> &scored, long unsigned int id, int minWords, float floorScore) {
    vector<pair<float, KeyFrame *>> groups;
    float bestAcc = floorScore;
    for (const auto &sm : scored) {
        float top = sm.first, acc = sm.first;
        KeyFrame *rep = sm.second;
        for (KeyFrame *nb : sm.second->GetBestCovisibilityKeyFrames(10)) {
            if (!Fld::NeighbourCounts(nb, id, minWords)) continue;
            acc += Fld::Score(nb);
            if (Fld::Score(nb) > top) {
                rep = nb;
                top = Fld::Score(nb);
            }
        }
        groups.push_back(make_pair(acc, rep));
        if (acc > bestAcc) bestAcc = acc;
    }
    const float retain = 0.75f * bestAcc;
    set<KeyFrame *> taken;
    vector<KeyFrame *> out;
    out.reserve(groups.size());
    for (const auto &g : groups)
        if (g.first > retain && taken.insert(g.second).second) out.push_back(g.second);
    return out;
}

template <class Fld>
int WordThreshold(const vector<KeyFrame *> &kfs) {
    int most = 0;
    for (KeyFrame *k : kfs)
        if (Fld::Words(k) > most) most = Fld::Words(k);
    return most * 0.8f;            // int <- float product, as :172 / :340
}

}  // namespace

vector<KeyFrame *> KeyFrameDatabase::DetectLoopCandidates(KeyFrame *pKF, float minScore) {
    const set<KeyFrame *> connected = pKF->GetConnectedKeyFrames();
    unique_lock<mutex> lock(mMutex);       // held to the end: the device table belongs to this object
    const vector<KeyFrame *> sharing = SharingWords<LoopFields>(mvInvertedFile, pKF->mBowVec, pKF->mnId, &connected);
    if (sharing.empty()) return vector<KeyFrame *>();
    const int minCommonWords = WordThreshold<LoopFields>(sharing);
    vector<KeyFrame *> toScore;
    for (KeyFrame *k : sharing)
        if (k->mnLoopWords > minCommonWords) toScore.push_back(k);
    const vector<float> si = ScoreAgainst(pKF->mBowVec, toScore);      // :179-196 in one launch
    vector<pair<float, KeyFrame *>> scored;
    for (size_t i = 0; i < toScore.size(); i++) {
        toScore[i]->mLoopScore = si[i];
        if (si[i] >= minScore) scored.push_back(make_pair(si[i], toScore[i]));
    }
    if (scored.empty()) return vector<KeyFrame *>();
    return BestGroups<LoopFields>(scored, pKF->mnId, minCommonWords, minScore);
}

vector<KeyFrame *> KeyFrameDatabase::DetectRelocalizationCandidates(Frame *F) {
    unique_lock<mutex> lock(mMutex);
    const vector<KeyFrame *> sharing = SharingWords<RelocFields>(mvInvertedFile, F->mBowVec, F->mnId, nullptr);
    if (sharing.empty()) return vector<KeyFrame *>();
    const int minCommonWords = WordThreshold<RelocFields>(sharing);
    vector<KeyFrame *> toScore;
    for (KeyFrame *k : sharing)
        if (k->mnRelocWords > minCommonWords) toScore.push_back(k);
    const vector<float> si = ScoreAgainst(F->mBowVec, toScore);         // :347-361 in one launch
    vector<pair<float, KeyFrame *>> scored;
    for (size_t i = 0; i < toScore.size(); i++) {
        toScore[i]->mRelocScore = si[i];
        scored.push_back(make_pair(si[i], toScore[i]));
    }
    return BestGroups<RelocFields>(scored, F->mnId, minCommonWords, 0.f);
}

}  // namespace ORB_SLAM2
