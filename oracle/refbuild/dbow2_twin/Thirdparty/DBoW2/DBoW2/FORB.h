/* DBoW2::FORB of the DBoW2 twin (see FeatureVector.h): the ORB descriptor functions TemplatedVocabulary is instantiated with
 * (reference include/ORBVocabulary.h:16-17).  TEST INFRASTRUCTURE, NOT PRODUCT CODE. */
#ifndef DBOW2_TWIN_FORB_H
#define DBOW2_TWIN_FORB_H
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
namespace DBoW2
{
class FORB
{
  public:
    typedef cv::Mat TDescriptor; /* 1 x 32, CV_8U */
    typedef const TDescriptor *pDescriptor;
    static const int L = 32; /* descriptor length in bytes */

    /* Hamming distance of two 256-bit rows (the bit-trick popcount on 32-bit words) */
    static int distance(const TDescriptor &a, const TDescriptor &b)
    {
        const uint32_t *pa = a.ptr<uint32_t>(), *pb = b.ptr<uint32_t>();
        int dist = 0;
        for (int i = 0; i < 8; i++, pa++, pb++) {
            uint32_t v = *pa ^ *pb;
            v = v - ((v >> 1) & 0x55555555u);
            v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
            dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
        }
        return dist;
    }
    /* "d0 d1 ... d31 " */
    static std::string toString(const TDescriptor &a)
    {
        std::stringstream ss;
        const unsigned char *p = a.ptr<unsigned char>();
        for (int i = 0; i < a.cols; ++i, ++p) ss << (int)*p << " ";
        return ss.str();
    }
    static void fromString(TDescriptor &a, const std::string &s)
    {
        a.create(1, FORB::L, CV_8U);
        unsigned char *p = a.ptr<unsigned char>();
        std::stringstream ss(s);
        for (int i = 0; i < FORB::L; ++i, ++p) {
            int n;
            ss >> n;
            if (!ss.fail()) *p = (unsigned char)n;
        }
    }
};
} // namespace DBoW2
#endif
