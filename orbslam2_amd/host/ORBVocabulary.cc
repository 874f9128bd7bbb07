// ORBVocabulary.cc -- see ORBVocabulary.h.  No Hamming distance is computed on the host: without the GPU library
// transform() throws.
#include "ORBVocabulary.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "host_ctx.h"

using namespace std;

namespace DBoW2 {

void BowVector::addWeight(WordId id, WordValue v) {
    auto vit = this->lower_bound(id);
    if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
    else this->insert(vit, value_type(id, v));
}

void BowVector::addIfNotExist(WordId id, WordValue v) {
    auto vit = this->lower_bound(id);
    if (vit == this->end() || (this->key_comp()(id, vit->first))) this->insert(vit, value_type(id, v));
}

void BowVector::normalize(LNorm norm_type) {
    double norm = 0.0;
    if (norm_type == DBoW2::L1) {
        for (auto it = begin(); it != end(); ++it) norm += fabs(it->second);
    } else {
        for (auto it = begin(); it != end(); ++it) norm += it->second * it->second;
        norm = sqrt(norm);
    }
    if (norm > 0.0)
        for (auto it = begin(); it != end(); ++it) it->second /= norm;
}

}  // namespace DBoW2

namespace ORB_SLAM2 {

namespace {
// device copies per calling thread (contexts are per thread): vocabulary object -> {generation, device id}
struct DevCopies {
    struct Entry { unsigned long generation; int id; };
    map<const ORBVocabulary *, Entry> m;
};
DevCopies &Copies() {
    static thread_local DevCopies c;
    return c;
}
}  // namespace

// Device copies are not released here: a vocabulary may be destroyed after the calling thread's context (static
// storage, another thread).  They go with their context, or are replaced when an object at the same address is loaded.
ORBVocabulary::~ORBVocabulary() {}

bool ORBVocabulary::loadFromTextFile(const string &filename) {
    ifstream f;
    f.open(filename.c_str());
    if (!f.is_open() || f.eof()) return false;
    m_words.clear();
    m_nodes.clear();
    string s;
    getline(f, s);
    stringstream ss;
    ss << s;
    int n1 = -1, n2 = -1;
    m_k = -1; m_L = -1;
    ss >> m_k;
    ss >> m_L;
    ss >> n1;
    ss >> n2;
    if (m_k < 0 || m_k > 20 || m_L < 1 || m_L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        fprintf(stderr, "Vocabulary loading failure: This is not a correct text file!\n");
        return false;
    }
    m_scoring = (DBoW2::ScoringType)n1;
    m_weighting = (DBoW2::WeightingType)n2;
    m_nodes.resize(1);   // the root
    // One node per line until end-of-file is SEEN: like the reference's loader this reads one more (empty) line after
    // a final newline and appends a node from it -- all fields 0: a childless child of the root with an all-zero
    // descriptor, weight 0 and word id 0 that is not in the word list.  A descriptor nearer to zero than to every
    // level-1 cluster therefore falls into a weight-0 "word" and is left out, exactly as in the reference.
    while (!f.eof()) {
        string snode;
        getline(f, snode);
        stringstream ssnode;
        ssnode << snode;
        const unsigned int nid = (unsigned int)m_nodes.size();
        m_nodes.resize(nid + 1);
        int pid = 0;
        ssnode >> pid;
        if (pid < 0 || (unsigned int)pid >= nid) return false;     // a parent must be an earlier node
        m_nodes[nid].parent = (unsigned int)pid;
        m_nodes[pid].children.push_back(nid);
        int nIsLeaf = 0;
        ssnode >> nIsLeaf;
        for (int iD = 0; iD < 32; iD++) {
            int e = 0;
            ssnode >> e;
            m_nodes[nid].descriptor[iD] = (uint8_t)e;
        }
        double w = 0;
        ssnode >> w;
        m_nodes[nid].weight = w;
        if (nIsLeaf > 0) {
            const unsigned int wid = (unsigned int)m_words.size();
            m_words.resize(wid + 1);
            m_nodes[nid].word_id = wid;
            m_words[wid] = nid;
        }
    }
    Flatten();
    return true;
}

void ORBVocabulary::saveToTextFile(const string &filename) const {
    ofstream f(filename.c_str());
    f << m_k << " " << m_L << " " << " " << (int)m_scoring << " " << (int)m_weighting << "\n";
    for (size_t i = 1; i < m_nodes.size(); i++) {
        const Node &nd = m_nodes[i];
        f << nd.parent << " " << (nd.isLeaf() ? 1 : 0) << " ";
        for (int j = 0; j < 32; j++) f << (int)nd.descriptor[j] << " ";
        f.precision(17);
        f << nd.weight << "\n";
    }
}

void ORBVocabulary::Flatten() {
    const size_t nn = m_nodes.size();
    m_flat = FlatTree();
    m_flat.child_off.assign(nn + 1, 0);
    m_flat.desc.resize(nn * 32);
    m_flat.word_id.assign(nn, -1);
    m_flat.weight.resize(nn);
    for (size_t i = 0; i < nn; i++) {
        const Node &nd = m_nodes[i];
        m_flat.child_off[i + 1] = m_flat.child_off[i] + (uint32_t)nd.children.size();
        for (unsigned int c : nd.children) m_flat.child_idx.push_back(c);
        memcpy(&m_flat.desc[i * 32], nd.descriptor, 32);
        if (nd.isLeaf()) m_flat.word_id[i] = (int32_t)nd.word_id;
        m_flat.weight[i] = nd.weight;
    }
    static std::atomic<unsigned long> next_generation{0};
    m_generation = ++next_generation;      // unique over all objects: a stale device copy is never mistaken for this tree
}

int ORBVocabulary::DeviceId() const {
    auto &m = Copies().m;
    auto it = m.find(this);
    if (it != m.end() && it->second.generation == m_generation) return it->second.id;
    orbm_ctx *ctx = detail::HostCtx();
    if (it != m.end()) { orbm_vocab_free(ctx, it->second.id); m.erase(it); }
    orbm_vocab_view v;
    v.n_nodes = (int32_t)m_nodes.size();
    v.depth = m_L;
    v.child_off = m_flat.child_off.data();
    v.child_idx = m_flat.child_idx.data();
    v.desc = m_flat.desc.data();
    v.word_id = m_flat.word_id.data();
    v.weight = m_flat.weight.data();
    v.weighting = (int32_t)m_weighting;
    v.scoring = (int32_t)m_scoring;
    int32_t id = -1;
    detail::HostCheck(orbm_vocab_upload(ctx, &v, &id), "orbm_vocab_upload");
    m[this] = {m_generation, id};
    return id;
}

bool ORBVocabulary::MustNormalize(DBoW2::LNorm &norm) const {
    switch (m_scoring) {
        case DBoW2::L2_NORM: norm = DBoW2::L2; return true;
        case DBoW2::DOT_PRODUCT: return false;
        default: norm = DBoW2::L1; return true;    // L1_NORM, CHI_SQUARE, KL, BHATTACHARYYA
    }
}

void ORBVocabulary::transform(const vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (empty()) return;
    const int n = (int)features.size();
    vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) memcpy(&desc[(size_t)i * 32], features[i].ptr<uint8_t>(), 32);
    vector<int32_t> word(n), node(n);
    if (n > 0)
        detail::HostCheck(orbm_vocab_transform(detail::HostCtx(), DeviceId(), n, desc.data(), levelsup, word.data(), node.data(), 0),
                          "orbm_vocab_transform");
    DBoW2::LNorm norm = DBoW2::L1;
    const bool must = MustNormalize(norm);
    const bool tf = m_weighting == DBoW2::TF || m_weighting == DBoW2::TF_IDF;
    for (int i_feature = 0; i_feature < n; i_feature++) {
        if (node[i_feature] < 0) continue;                          // the word's weight is not positive: "stopped"
        const DBoW2::WordId id = (DBoW2::WordId)word[i_feature];
        const DBoW2::WordValue w = m_nodes[m_words[id]].weight;
        if (tf) v.addWeight(id, w);
        else v.addIfNotExist(id, w);
        // FeatureVector::addFeature
        const unsigned int nid = (unsigned int)node[i_feature];
        auto vit = fv.lower_bound(nid);
        if (vit != fv.end() && vit->first == nid) vit->second.push_back(i_feature);
        else {
            vit = fv.insert(vit, DBoW2::FeatureVector::value_type(nid, vector<unsigned int>()));
            vit->second.push_back(i_feature);
        }
    }
    if (tf && !v.empty() && !must) {
        const double nd = v.size();
        for (auto vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
    }
    if (must) v.normalize(norm);
}

double ORBVocabulary::score(const DBoW2::BowVector &v1, const DBoW2::BowVector &v2) const {
    if (m_scoring != DBoW2::L1_NORM && m_scoring != DBoW2::L2_NORM && m_scoring != DBoW2::DOT_PRODUCT)
        throw runtime_error("ORBVocabulary::score: only L1_NORM, L2_NORM and DOT_PRODUCT are implemented");
    auto v1_it = v1.begin(), v2_it = v2.begin();
    const auto v1_end = v1.end(), v2_end = v2.end();
    double score = 0;
    while (v1_it != v1_end && v2_it != v2_end) {
        const DBoW2::WordValue &vi = v1_it->second;
        const DBoW2::WordValue &wi = v2_it->second;
        if (v1_it->first == v2_it->first) {
            if (m_scoring == DBoW2::L1_NORM) score += fabs(vi - wi) - fabs(vi) - fabs(wi);
            else score += vi * wi;
            ++v1_it;
            ++v2_it;
        } else if (v1_it->first < v2_it->first) {
            v1_it = v1.lower_bound(v2_it->first);
        } else {
            v2_it = v2.lower_bound(v1_it->first);
        }
    }
    if (m_scoring == DBoW2::L1_NORM) return -score / 2.0;           // ||v - w||_L1 = 2 + sum(|vi - wi| - |vi| - |wi|)
    if (m_scoring == DBoW2::L2_NORM) return score >= 1 ? 1.0 : 1.0 - sqrt(1.0 - score);   // ||v - w||_L2 = sqrt(2 - 2 v.w)
    return score;
}

}  // namespace ORB_SLAM2
