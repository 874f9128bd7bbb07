// ORBVocabulary.h -- drop-in for ORB_SLAM2::ORBVocabulary (ORBVocabulary.h: a typedef of
// DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB>) for the calls the reference makes on it:
//   loadFromTextFile   System.cc:80
//   transform(,,,4)    Frame.cc:531, KeyFrame.cc:76
//   size               KeyFrameDataBase.cc:92
//   score              KeyFrameDataBase.cc:188,356  LoopClosing.cc:229
// The tree descent (k Hamming distances per level and feature) runs on the GPU through orbm_vocab_transform; the
// BowVector / FeatureVector maps are filled on the host from the returned word / node ids, in feature order, with
// DBoW2's own arithmetic (double).  DBoW2 is a third-party dependency that is not in the reference tree: its
// published algorithm is restated here and the parity of this class is unpinned.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "orbslam_types.h"

namespace ORB_SLAM2 {

class ORBVocabulary {
public:
    ORBVocabulary() {}
    ~ORBVocabulary();
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // the text format ORB-SLAM2 ships its vocabulary in: "k L scoring weighting", then one node per line
    // "parent is_leaf d0 .. d31 weight" (node ids are line numbers, the root is node 0 and has no line)
    bool loadFromTextFile(const std::string &filename);
    void saveToTextFile(const std::string &filename) const;

    // features: one 1x32 CV_8U row per keypoint (Converter::toDescriptorVector)
    void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const;
    double score(const DBoW2::BowVector &a, const DBoW2::BowVector &b) const;

    unsigned int size() const { return (unsigned int)m_words.size(); }
    bool empty() const { return m_words.empty(); }
    int getBranchingFactor() const { return m_k; }
    int getDepthLevels() const { return m_L; }
    DBoW2::WeightingType getWeightingType() const { return m_weighting; }
    DBoW2::ScoringType getScoringType() const { return m_scoring; }

    // flat form (what orbm_vocab_upload takes); valid until the next load
    struct FlatTree {
        std::vector<uint32_t> child_off, child_idx;
        std::vector<uint8_t> desc;
        std::vector<int32_t> word_id;
        std::vector<double> weight;
    };
    const FlatTree &flat() const { return m_flat; }

private:
    int DeviceId() const;   // uploads on first use in the calling thread's context
    void Flatten();
    bool MustNormalize(DBoW2::LNorm &norm) const;

    struct Node {
        unsigned int parent = 0;
        std::vector<unsigned int> children;
        uint8_t descriptor[32] = {0};
        double weight = 0;
        unsigned int word_id = 0;
        bool isLeaf() const { return children.empty(); }
    };
    int m_k = 10, m_L = 5;
    DBoW2::WeightingType m_weighting = DBoW2::TF_IDF;
    DBoW2::ScoringType m_scoring = DBoW2::L1_NORM;
    std::vector<Node> m_nodes;
    std::vector<unsigned int> m_words;   // word id -> node id
    FlatTree m_flat;
    unsigned long m_generation = 0;      // bumped by every load: stale device copies are replaced
};

}  // namespace ORB_SLAM2
