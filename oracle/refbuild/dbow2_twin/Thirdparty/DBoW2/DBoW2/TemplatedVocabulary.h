/*
 * DBoW2::TemplatedVocabulary of the DBoW2 twin (see FeatureVector.h): the vocabulary tree with the members ORB-SLAM2 uses --
 * loadFromTextFile (src/System.cc:126), loadFromBinaryFile (:129) / saveToBinaryFile (tool/text2binary.cc:23-45),
 * transform(features, BowVector&, FeatureVector&, levelsup) (src/Frame.cc:553, src/KeyFrame.cc:82), score
 * (src/LoopClosing.cc:156), size / empty -- restated from the published DBoW2 + ORB-SLAM2 code in DBoW2's class shape.
 * Text file: "k L scoring weighting", then one line per non-root node in id order: "parent is_leaf d0 .. d31 weight".
 * Binary file: u32 nb_nodes | u32 size_node | i32 k | i32 L | i32 scoring | i32 weighting, then per non-root node
 * i32 parent | 32 descriptor bytes | f32 weight | u8 is_leaf (size_node = 41).
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 */
#ifndef DBOW2_TWIN_TEMPLATEDVOCABULARY_H
#define DBOW2_TWIN_TEMPLATEDVOCABULARY_H
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "BowVector.h"
#include "FeatureVector.h"
namespace DBoW2
{
template <class TDescriptor, class F> class TemplatedVocabulary
{
  public:
    TemplatedVocabulary(int k = 10, int L = 5, WeightingType weighting = TF_IDF, ScoringType scoring = L1_NORM)
        : m_k(k), m_L(L), m_weighting(weighting), m_scoring(scoring)
    {
    }
    virtual ~TemplatedVocabulary() {}

    virtual inline unsigned int size() const { return (unsigned int)m_words.size(); }
    virtual inline bool empty() const { return m_words.empty(); }
    int getBranchingFactor() const { return m_k; }
    int getDepthLevels() const { return m_L; }
    WeightingType getWeightingType() const { return m_weighting; }
    ScoringType getScoringType() const { return m_scoring; }
    unsigned int nodes() const { return (unsigned int)m_nodes.size(); }

    /* features -> BowVector + FeatureVector (node ids levelsup levels above the leaves) */
    virtual void transform(const std::vector<TDescriptor> &features, BowVector &v, FeatureVector &fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        if (empty()) return;
        /* the scoring object decides: L1_NORM, L2_NORM normalise with their own norm; the others leave the vector alone or,
         * for TF / TF_IDF, divide by the number of words */
        LNorm norm = L1;
        const bool must = must_normalize(norm);
        if (m_weighting == TF || m_weighting == TF_IDF) {
            unsigned int i_feature = 0;
            for (typename std::vector<TDescriptor>::const_iterator fit = features.begin(); fit < features.end(); ++fit, ++i_feature) {
                WordId id;
                NodeId nid;
                WordValue w; /* idf value with TF_IDF, 1 with TF */
                transform(*fit, id, w, &nid, levelsup);
                if (w > 0) { /* not stopped */
                    v.addWeight(id, w);
                    fv.addFeature(nid, i_feature);
                }
            }
            if (!v.empty() && !must) { /* unnecessary when normalising */
                const double nd = (double)v.size();
                for (BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
            }
        } else { /* IDF || BINARY */
            unsigned int i_feature = 0;
            for (typename std::vector<TDescriptor>::const_iterator fit = features.begin(); fit < features.end(); ++fit, ++i_feature) {
                WordId id;
                NodeId nid;
                WordValue w; /* idf value with IDF, 1 with BINARY */
                transform(*fit, id, w, &nid, levelsup);
                if (w > 0) {
                    v.addIfNotExist(id, w);
                    fv.addFeature(nid, i_feature);
                }
            }
        }
        if (must) v.normalize(norm);
    }

    /* L1 score of two normalised vectors (the scoring ORBvoc uses): 1 - 0.5 * sum |a - b| over all words, written over
     * the common words as -sum(|a - b| - |a| - |b|) / 2 */
    inline double score(const BowVector &v1, const BowVector &v2) const
    {
        assert(m_scoring == L1_NORM);
        BowVector::const_iterator v1_it = v1.begin(), v2_it = v2.begin();
        const BowVector::const_iterator v1_end = v1.end(), v2_end = v2.end();
        double score = 0;
        while (v1_it != v1_end && v2_it != v2_end) {
            const WordValue &vi = v1_it->second, &wi = v2_it->second;
            if (v1_it->first == v2_it->first) {
                score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
                ++v1_it;
                ++v2_it;
            } else if (v1_it->first < v2_it->first) {
                v1_it = v1.lower_bound(v2_it->first);
            } else {
                v2_it = v2.lower_bound(v1_it->first);
            }
        }
        return -score / 2.0;
    }

    bool loadFromTextFile(const std::string &filename)
    {
        std::ifstream f;
        f.open(filename.c_str());
        if (f.eof()) return false;
        m_words.clear();
        m_nodes.clear();
        std::string s;
        std::getline(f, s);
        std::stringstream ss;
        ss << s;
        ss >> m_k;
        ss >> m_L;
        int n1, n2;
        ss >> n1;
        ss >> n2;
        if (m_k < 0 || m_k > 20 || m_L < 1 || m_L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
            std::cerr << "Vocabulary loading failure: This is not a correct text file!" << std::endl;
            return false;
        }
        m_scoring = (ScoringType)n1;
        m_weighting = (WeightingType)n2;
        /* nodes: at most (k^(L+1) - 1) / (k - 1) */
        const int expected_nodes = (int)((std::pow((double)m_k, (double)m_L + 1) - 1) / (m_k - 1));
        m_nodes.reserve(expected_nodes);
        m_words.reserve((size_t)std::pow((double)m_k, (double)m_L + 1));
        m_nodes.resize(1);
        m_nodes[0].id = 0;
        while (!f.eof()) {
            std::string snode;
            std::getline(f, snode);
            if (snode.find_first_not_of(" \t\r\n") == std::string::npos) continue; /* the empty line behind the last node */
            std::stringstream ssnode;
            ssnode << snode;
            const int nid = (int)m_nodes.size();
            m_nodes.resize(m_nodes.size() + 1);
            m_nodes[nid].id = nid;
            int pid;
            ssnode >> pid;
            m_nodes[nid].parent = pid;
            m_nodes[pid].children.push_back(nid);
            int nIsLeaf;
            ssnode >> nIsLeaf;
            std::stringstream ssd;
            for (int iD = 0; iD < F::L; iD++) {
                std::string sElement;
                ssnode >> sElement;
                ssd << sElement << " ";
            }
            F::fromString(m_nodes[nid].descriptor, ssd.str());
            ssnode >> m_nodes[nid].weight;
            if (nIsLeaf > 0) {
                const int wid = (int)m_words.size();
                m_words.resize(wid + 1);
                m_nodes[nid].word_id = wid;
                m_words[wid] = nid;
            } else {
                m_nodes[nid].children.reserve(m_k);
            }
        }
        return true;
    }

    bool loadFromBinaryFile(const std::string &filename)
    {
        std::fstream f;
        f.open(filename.c_str(), std::ios_base::in | std::ios::binary);
        if (!f.is_open()) return false;
        unsigned int nb_nodes, size_node;
        f.read((char *)&nb_nodes, sizeof(nb_nodes));
        f.read((char *)&size_node, sizeof(size_node));
        f.read((char *)&m_k, sizeof(m_k));
        f.read((char *)&m_L, sizeof(m_L));
        int sc, we;
        f.read((char *)&sc, sizeof(sc));
        f.read((char *)&we, sizeof(we));
        m_scoring = (ScoringType)sc;
        m_weighting = (WeightingType)we;
        if (!f || size_node != sizeof(int) + F::L + sizeof(float) + sizeof(bool) || nb_nodes < 1) return false;
        m_words.clear();
        m_words.reserve((size_t)std::pow((double)m_k, (double)m_L + 1));
        m_nodes.clear();
        m_nodes.resize(nb_nodes);
        m_nodes[0].id = 0;
        std::vector<char> buf(size_node);
        for (unsigned int nid = 1; nid < nb_nodes; ++nid) {
            f.read(&buf[0], size_node);
            if (!f) return false;
            m_nodes[nid].id = nid;
            int parent;
            std::memcpy(&parent, &buf[0], sizeof(int));
            m_nodes[nid].parent = (NodeId)parent;
            m_nodes[parent].children.push_back(nid);
            m_nodes[nid].descriptor.create(1, F::L, CV_8U);
            std::memcpy(m_nodes[nid].descriptor.data, &buf[4], F::L);
            float w;
            std::memcpy(&w, &buf[4 + F::L], sizeof(float));
            m_nodes[nid].weight = w;
            if (buf[8 + F::L]) { /* is leaf */
                const int wid = (int)m_words.size();
                m_words.resize(wid + 1);
                m_nodes[nid].word_id = wid;
                m_words[wid] = nid;
            } else {
                m_nodes[nid].children.reserve(m_k);
            }
        }
        f.close();
        return true;
    }

    void saveToBinaryFile(const std::string &filename) const
    {
        std::fstream f;
        f.open(filename.c_str(), std::ios_base::out | std::ios::binary);
        const unsigned int nb_nodes = (unsigned int)m_nodes.size();
        float _weight;
        const unsigned int size_node = sizeof(int) + F::L * sizeof(char) + sizeof(_weight) + sizeof(bool);
        f.write((const char *)&nb_nodes, sizeof(nb_nodes));
        f.write((const char *)&size_node, sizeof(size_node));
        f.write((const char *)&m_k, sizeof(m_k));
        f.write((const char *)&m_L, sizeof(m_L));
        const int sc = (int)m_scoring, we = (int)m_weighting;
        f.write((const char *)&sc, sizeof(sc));
        f.write((const char *)&we, sizeof(we));
        for (size_t i = 1; i < nb_nodes; i++) {
            const Node &node = m_nodes[i];
            const int parent = (int)node.parent;
            f.write((const char *)&parent, sizeof(parent));
            f.write((const char *)node.descriptor.data, F::L);
            _weight = (float)node.weight;
            f.write((const char *)&_weight, sizeof(_weight));
            const bool is_leaf = node.isLeaf();
            f.write((const char *)&is_leaf, sizeof(is_leaf)); /* last, as the original: no alignment to keep */
        }
        f.close();
    }

    /* read access for the glue (dbow2_twin_api.cpp) */
    struct Node {
        NodeId id;
        WordValue weight;
        std::vector<NodeId> children;
        NodeId parent;
        TDescriptor descriptor;
        WordId word_id;
        Node() : id(0), weight(0), parent(0), word_id(0) {}
        inline bool isLeaf() const { return children.empty(); }
    };
    const std::vector<Node> &getNodes() const { return m_nodes; }

  protected:
    /* one feature down the tree: the closest child at every level (first wins on equal distance) */
    virtual void transform(const TDescriptor &feature, WordId &word_id, WordValue &weight, NodeId *nid, int levelsup) const
    {
        const int nid_level = m_L - levelsup;
        if (nid_level <= 0 && nid != NULL) *nid = 0; /* root */
        NodeId final_id = 0;
        int current_level = 0;
        do {
            ++current_level;
            const std::vector<NodeId> &nodes = m_nodes[final_id].children;
            final_id = nodes[0];
            double best_d = F::distance(feature, m_nodes[final_id].descriptor);
            for (typename std::vector<NodeId>::const_iterator nit = nodes.begin() + 1; nit != nodes.end(); ++nit) {
                const NodeId id = *nit;
                const double d = F::distance(feature, m_nodes[id].descriptor);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (nid != NULL && current_level == nid_level) *nid = final_id;
        } while (!m_nodes[final_id].isLeaf());
        word_id = m_nodes[final_id].word_id;
        weight = m_nodes[final_id].weight;
    }
    bool must_normalize(LNorm &norm) const
    {
        if (m_scoring == L1_NORM) { norm = L1; return true; }
        if (m_scoring == L2_NORM) { norm = L2; return true; }
        return false;
    }

    int m_k, m_L;
    WeightingType m_weighting;
    ScoringType m_scoring;
    std::vector<Node> m_nodes;
    std::vector<NodeId> m_words; /* word id -> node id (the original keeps Node pointers) */
};
} // namespace DBoW2
#endif
