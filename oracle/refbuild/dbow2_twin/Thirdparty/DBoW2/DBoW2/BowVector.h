/* DBoW2::BowVector of the DBoW2 twin (see FeatureVector.h).  TEST INFRASTRUCTURE, NOT PRODUCT CODE. */
#ifndef DBOW2_TWIN_BOWVECTOR_H
#define DBOW2_TWIN_BOWVECTOR_H
#include <cmath>
#include <map>

#include "FeatureVector.h"
namespace DBoW2
{
enum LNorm { L1, L2 };
enum WeightingType { TF_IDF, TF, IDF, BINARY };
enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };

/* word id -> value */
class BowVector : public std::map<WordId, WordValue>
{
  public:
    void addWeight(WordId id, WordValue v)
    {
        BowVector::iterator vit = this->lower_bound(id);
        if (vit != this->end() && !(this->key_comp()(id, vit->first))) vit->second += v;
        else this->insert(vit, BowVector::value_type(id, v));
    }
    void addIfNotExist(WordId id, WordValue v)
    {
        BowVector::iterator vit = this->lower_bound(id);
        if (vit == this->end() || (this->key_comp()(id, vit->first))) this->insert(vit, BowVector::value_type(id, v));
    }
    void normalize(LNorm norm_type)
    {
        double norm = 0.0;
        BowVector::iterator it;
        if (norm_type == DBoW2::L1) {
            for (it = begin(); it != end(); ++it) norm += std::fabs(it->second);
        } else {
            for (it = begin(); it != end(); ++it) norm += it->second * it->second;
            norm = std::sqrt(norm);
        }
        if (norm > 0.0)
            for (it = begin(); it != end(); ++it) it->second /= norm;
    }
};
} // namespace DBoW2
#endif
