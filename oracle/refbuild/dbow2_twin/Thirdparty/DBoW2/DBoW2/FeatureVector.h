/*
 * DBoW2::FeatureVector -- part of the DBoW2 TWIN (oracle/refbuild/dbow2_twin): a second, independent restatement of the
 * published DBoW2 / ORB-SLAM2 vocabulary code in DBoW2's own class shape, so that the one DBoW2 client in the reference
 * tree (tool/text2binary.cc) compiles and runs UNCHANGED and the product's vocabulary file IO and transform can be checked
 * against compiled C++ (not against the oracle's numpy / C restatement).  DBoW2 itself is NOT in /root/reference
 * (perfect/Thirdparty/DBoW2 holds a readme only): this pins nothing to the original library -- parity stays "published
 * algorithm".  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 */
#ifndef DBOW2_TWIN_FEATUREVECTOR_H
#define DBOW2_TWIN_FEATUREVECTOR_H
#include <map>
#include <vector>
namespace DBoW2
{
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;

/* node id at the chosen tree level -> indices of the features that descend through it */
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> >
{
  public:
    void addFeature(NodeId id, unsigned int i_feature)
    {
        FeatureVector::iterator vit = this->lower_bound(id);
        if (vit != this->end() && vit->first == id) {
            vit->second.push_back(i_feature);
        } else {
            vit = this->insert(vit, FeatureVector::value_type(id, std::vector<unsigned int>()));
            vit->second.push_back(i_feature);
        }
    }
};
} // namespace DBoW2
#endif
