/*
 * DBoW2::FeatureVector -- stand-in for the header the reference includes (src/ORBmatcher.cc:30) but does not
 * vendor (perfect/Thirdparty/DBoW2 holds a readme only).  The published DBoW2 definition is exactly this: a
 * std::map from vocabulary node id to the indices of the features that descend through that node.
 * TEST INFRASTRUCTURE (oracle/_ref), NOT PRODUCT CODE.
 */
#ifndef ORBFE_STUB_DBOW2_FEATUREVECTOR_H
#define ORBFE_STUB_DBOW2_FEATUREVECTOR_H
#include <map>
#include <vector>
namespace DBoW2
{
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> >
{
  public:
    void addFeature(NodeId id, unsigned int i_feature)
    {
        FeatureVector::iterator vit = this->lower_bound(id);
        if (vit != this->end() && vit->first == id) vit->second.push_back(i_feature);
        else {
            vit = this->insert(vit, FeatureVector::value_type(id, std::vector<unsigned int>()));
            vit->second.push_back(i_feature);
        }
    }
};
class BowVector : public std::map<WordId, WordValue>
{
};
} // namespace DBoW2
#endif
