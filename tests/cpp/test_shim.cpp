// test_shim.cpp -- the C++ shim keeps the reference's call shapes.  Reads a raw gray frame, calls
//   ORB_SLAM2::ORBextractor(nf, 1.2f, 8, 20, 7)(image, cv::Mat(), keys, descriptors)
// exactly like Frame::ExtractORB (reference src/Frame.cc:337-343), runs SearchByBoW on mock KeyFrame / Frame types
// carrying the extractor output, and dumps everything for the Python test to compare with the oracle.
//   usage: test_shim in.raw W H nfeatures out.bin
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"

struct MockMapPoint {
    bool bad;
    bool isBad() const { return bad; }
};
typedef std::map<unsigned, std::vector<unsigned> > FeatureVector;
struct MockFrame {
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors;
    FeatureVector mFeatVec;
    std::vector<MockMapPoint *> mvpMapPoints;
    std::vector<MockMapPoint *> GetMapPointMatches() { return mvpMapPoints; }
};

static void fill(MockFrame &f, std::vector<MockMapPoint> &pool)
{
    f.N = (int)f.mvKeys.size();
    f.mvKeysUn = f.mvKeys;
    pool.resize(f.N);
    f.mvpMapPoints.resize(f.N);
    for (int i = 0; i < f.N; ++i) {
        pool[i].bad = (i % 11 == 0);
        f.mvpMapPoints[i] = (i % 7 == 0) ? NULL : &pool[i];
        f.mFeatVec[(unsigned)(f.mDescriptors.ptr(i)[0] >> 2)].push_back(i);  // toy vocabulary: 64 nodes from byte 0
    }
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]), nf = atoi(argv[4]);
    std::vector<uint8_t> buf((size_t)W * H * 2);
    FILE *fi = fopen(argv[1], "rb");
    if (!fi || fread(buf.data(), 1, buf.size(), fi) != buf.size()) return 3;
    fclose(fi);
    ORB_SLAM2::ORBextractor ext(nf, 1.2f, 8, 20, 7);
    ext.mbKeepPyramid = true;
    MockFrame fr[2];
    std::vector<MockMapPoint> pool[2];
    for (int k = 0; k < 2; ++k) {
        cv::Mat im(H, W, CV_8UC1, buf.data() + (size_t)k * W * H, (size_t)W);
        ext(im, cv::Mat(), fr[k].mvKeys, fr[k].mDescriptors);
        if (ext.LastStatus() != 0) return 4;
        fill(fr[k], pool[k]);
    }
    cv::Mat empty;
    std::vector<cv::KeyPoint> kk(3);
    cv::Mat dd;
    ext(empty, cv::Mat(), kk, dd);  // empty image leaves the outputs untouched
    if (kk.size() != 3) return 5;

    ORB_SLAM2::ORBmatcher m1(0.7f, true), m2(0.75f, true), m3(0.9f, true);
    std::vector<MockMapPoint *> mp1, mp2;
    const int n1 = m1.SearchByBoW(&fr[0], fr[1], mp1);          // (KeyFrame*, Frame&)
    const int n2 = m2.SearchByBoW(&fr[0], &fr[1], mp2);         // (KeyFrame*, KeyFrame*)
    std::vector<int> bf;
    const int n3 = m3.MatchBruteForce(fr[1].mDescriptors, fr[1].mvKeys, fr[0].mDescriptors, fr[0].mvKeys, bf);
    const int d01 = ORB_SLAM2::ORBmatcher::DescriptorDistance(fr[0].mDescriptors.row(0), fr[0].mDescriptors.row(1));

    FILE *fo = fopen(argv[5], "wb");
    if (!fo) return 6;
    for (int k = 0; k < 2; ++k) {
        int n = fr[k].N;
        fwrite(&n, 4, 1, fo);
        fwrite(fr[k].mvKeys.data(), sizeof(cv::KeyPoint), n, fo);
        for (int i = 0; i < n; ++i) fwrite(fr[k].mDescriptors.ptr(i), 1, 32, fo);
    }
    int hdr[4] = {n1, n2, n3, d01};
    fwrite(hdr, 4, 4, fo);
    for (int i = 0; i < fr[1].N; ++i) { int v = mp1[i] ? (int)(mp1[i] - &pool[0][0]) : -1; fwrite(&v, 4, 1, fo); }
    for (int i = 0; i < fr[0].N; ++i) { int v = mp2[i] ? (int)(mp2[i] - &pool[1][0]) : -1; fwrite(&v, 4, 1, fo); }
    fwrite(bf.data(), 4, bf.size(), fo);
    int lw = ext.mvImagePyramid[7].cols, lh = ext.mvImagePyramid[7].rows;
    fwrite(&lw, 4, 1, fo);
    fwrite(&lh, 4, 1, fo);
    for (int y = -19; y < lh + 19; ++y) fwrite(ext.mvImagePyramid[7].data + (long)y * (long)ext.mvImagePyramid[7].step - 19, 1, lw + 38, fo);
    fclose(fo);
    printf("shim ok: %d %d keypoints, bow %d / %d, bf %d\n", fr[0].N, fr[1].N, n1, n2, n3);
    return 0;
}
