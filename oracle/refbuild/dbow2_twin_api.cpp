/*
 * dbow2_twin_api.cpp -- flat C entry points over the DBoW2 twin's ORB_SLAM2::ORBVocabulary (the reference's own typedef,
 * include/ORBVocabulary.h:16-17, instantiated over dbow2_twin/Thirdparty/DBoW2/DBoW2/*.h): load / save in the two file
 * layouts, the tree as arrays, transform() as Frame::ComputeBoW calls it (src/Frame.cc:546-555).  -> oracle/_ref/libdbow2_twin.so
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ORBVocabulary.h" /* the reference's header */

using ORB_SLAM2::ORBVocabulary;

extern "C" {
void *twin_voc_load(const char *path, int binary)
{
    ORBVocabulary *v = new ORBVocabulary();
    const bool ok = binary ? v->loadFromBinaryFile(path) : v->loadFromTextFile(path);
    if (!ok) {
        delete v;
        return 0;
    }
    return v;
}
void twin_voc_free(void *h) { delete (ORBVocabulary *)h; }
void twin_voc_save_binary(void *h, const char *path) { ((ORBVocabulary *)h)->saveToBinaryFile(path); }
void twin_voc_info(void *h, int *k, int *L, int *nnodes, int *nwords, int *scoring, int *weighting)
{
    const ORBVocabulary *v = (const ORBVocabulary *)h;
    *k = v->getBranchingFactor();
    *L = v->getDepthLevels();
    *nnodes = (int)v->nodes();
    *nwords = (int)v->size();
    *scoring = (int)v->getScoringType();
    *weighting = (int)v->getWeightingType();
}
/* parent[nnodes], is_leaf[nnodes], desc[nnodes * 32], weight[nnodes], word_id[nnodes] (0 for inner nodes) */
void twin_voc_arrays(void *h, uint32_t *parent, uint8_t *is_leaf, uint8_t *desc, double *weight, uint32_t *word_id)
{
    const ORBVocabulary *v = (const ORBVocabulary *)h;
    const std::vector<ORBVocabulary::Node> &nodes = v->getNodes();
    for (size_t i = 0; i < nodes.size(); i++) {
        parent[i] = nodes[i].parent;
        is_leaf[i] = i > 0 && nodes[i].isLeaf();
        if (i > 0) memcpy(desc + i * 32, nodes[i].descriptor.data, 32);
        else memset(desc, 0, 32);
        weight[i] = nodes[i].weight;
        word_id[i] = is_leaf[i] ? nodes[i].word_id : 0;
    }
}
/* Frame::ComputeBoW (src/Frame.cc:546-555): vector<cv::Mat> of the descriptor rows -> transform(v, BowVec, FeatVec, levelsup).
 * BowVector as (id, value) ascending; FeatureVector as CSR (node ascending, off, idx). */
void twin_voc_transform(void *h, const uint8_t *desc, int n, int levelsup, uint32_t *bow_id, double *bow_val, int *nbow,
                        uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx, int *nfv)
{
    const ORBVocabulary *v = (const ORBVocabulary *)h;
    std::vector<cv::Mat> feats;
    feats.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        cv::Mat row(1, 32, CV_8U);
        memcpy(row.data, desc + (size_t)i * 32, 32);
        feats.push_back(row);
    }
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    v->transform(feats, bv, fv, levelsup);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++k) {
        bow_id[k] = it->first;
        bow_val[k] = it->second;
    }
    *nbow = k;
    k = 0;
    uint32_t o = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        fv_node[k] = it->first;
        fv_off[k] = o;
        for (size_t j = 0; j < it->second.size(); j++) fv_idx[o++] = it->second[j];
    }
    fv_off[k] = o;
    *nfv = k;
}
/* ORBVocabulary::score of two BowVectors given as (id, value) lists (src/LoopClosing.cc:156) */
double twin_voc_score(void *h, const uint32_t *id1, const double *v1, int n1, const uint32_t *id2, const double *v2, int n2)
{
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; i++) a.addWeight(id1[i], v1[i]);
    for (int i = 0; i < n2; i++) b.addWeight(id2[i], v2[i]);
    return ((const ORBVocabulary *)h)->score(a, b);
}
}
