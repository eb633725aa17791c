#!/usr/bin/env python3
"""oracle/refbuild/slice.py -- cut whole function definitions VERBATIM out of reference translation units that cannot be
compiled as a whole here (src/Frame.cc, src/MapPoint.cc, src/KeyFrame.cc need OpenCV's calib3d / DBoW2 templates / g2o),
into generated include files under oracle/_ref/ (git-ignored; nothing of the reference enters the repository).
TEST INFRASTRUCTURE, NOT PRODUCT CODE.

usage: slice.py <reference root> <out dir>
Each slice is located by the exact text that opens the definition and runs to the closing brace of that function (brace
matching on the raw text); the script fails loudly if a signature is not found exactly once, so a changed reference
cannot be sliced silently wrong.  ref_slices.cpp #includes the generated files inside `namespace ORB_SLAM2 { ... }`.
"""
import os
import sys

SLICES = {
    "gen_frame_grid.inc": ("src/Frame.cc", [
        "void Frame::AssignFeaturesToGrid()",
        ("vector<size_t> Frame::GetFeaturesInArea(const float &x, const float  &y, const float  &r, const int minLevel, const int maxLevel) const",
         "vector<size_t> Frame::GetFeaturesInArea(\n"),  # perfect/src/Frame.cc breaks the parameter list over four lines
        "bool Frame::PosInGrid(const cv::KeyPoint &kp, int &posX, int &posY)",
        "void Frame::ComputeStereoMatches()",
    ]),
    "gen_frame_extract.inc": ("src/Frame.cc", [
        "void Frame::ExtractORB(int flag, const cv::Mat &im)",
    ]),
    "gen_frame_stereo_ctor.inc": ("src/Frame.cc", [
        "Frame::Frame(const cv::Mat &imLeft, const cv::Mat &imRight, const double &timeStamp, ORBextractor* extractorLeft, ORBextractor* extractorRight, ORBVocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth)",
        "void Frame::ExtractORB(int flag, const cv::Mat &im)",
        "void Frame::UndistortKeyPoints()",
        "void Frame::ComputeImageBounds(const cv::Mat &imLeft)",
        # the RGB-D constructor (BASELINE configs 1-3 are RGB-D: src/Frame.cc:176-245), the monocular one (:247-...) and the
        # depth look-up they call (:850)
        ("Frame::Frame(const cv::Mat &imGray, const cv::Mat &imDepth, const double &timeStamp, ORBextractor* extractor,ORBVocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth)",
         "Frame::Frame(const cv::Mat &imGray, const cv::Mat &imDepth, \n"),   # perfect/src/Frame.cc: three lines, + mImDepth(imDepth)
        "Frame::Frame(const cv::Mat &imGray, const double &timeStamp, ORBextractor* extractor,ORBVocabulary* voc, cv::Mat &K, cv::Mat &distCoef, const float &bf, const float &thDepth)",
        "void Frame::ComputeStereoFromRGBD(const cv::Mat &imDepth)",
    ]),
    # perfect/ only: the RGB-D constructor with the dynamic-object mask (perfect/src/Frame.cc:328-420): keypoints whose mask
    # pixel is not 1 are dropped after extraction.  Always cut from the perfect copy, whichever root this run was given.
    "gen_frame_masked_ctor.inc": (("perfect/src/Frame.cc", "src/Frame.cc"), [
        "Frame::Frame(const cv::Mat &imGray, const cv::Mat &imDepth,\n",
    ]),
    "gen_mappoint.inc": ("src/MapPoint.cc", [
        "void MapPoint::ComputeDistinctiveDescriptors()",
        "int MapPoint::PredictScale(const float &currentDist, KeyFrame* pKF)",
        "int MapPoint::PredictScale(const float &currentDist, Frame* pF)",
    ]),
    # perfect/ only: the binary map file (perfect/src/Map.cc:143-430); absent from the top-level copy -> skipped there
    "gen_map_io.inc": ("src/Map.cc", [
        "KeyFrame* Map::_ReadKeyFrame(ifstream &f,",
        "MapPoint* Map::_ReadMapPoint(ifstream &f)",
        "bool Map::Load(const string &filename, ORBVocabulary &voc)",
        "void Map::_WriteMapPoint(ofstream &f, MapPoint* mp)",
        "void Map::_WriteKeyFrame(ofstream &f, KeyFrame* kf, map<MapPoint*, unsigned long int>& idx_of_mp)",
        "bool Map::Save(const string &filename)",
    ], "optional"),
    "gen_keyframe_grid.inc": ("src/KeyFrame.cc", [
        "vector<size_t> KeyFrame::GetFeaturesInArea(const float &x, const float &y, const float &r) const",
    ]),
}


def cut(text, sigs):
    """sigs: one signature or a tuple of alternative spellings (src/ and perfect/src/ differ in layout only)"""
    if isinstance(sigs, str):
        sigs = (sigs,)
    hits = [s for s in sigs if text.count(s) == 1]
    if len(hits) != 1:
        raise SystemExit(f"slice.py: {len(hits)} of the alternative signatures found exactly once (expected 1): {sigs}")
    sig = hits[0]
    start = text.index(sig)
    i = text.index("{", start)
    depth = 0
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[start:i + 1]
        i += 1


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    for name, entry in SLICES.items():
        src, sigs = entry[0], entry[1]
        if not isinstance(src, str):   # alternative locations: the first that exists under this root
            src = [c for c in src if os.path.exists(os.path.join(ref, c))][0]
        text = open(os.path.join(ref, src), encoding="utf-8", errors="replace").read()
        if len(entry) > 2 and entry[2] == "optional" and all(text.count(s if isinstance(s, str) else s[0]) == 0 for s in sigs):
            continue  # this copy of the reference does not have these functions at all
        body = [f"/* GENERATED by oracle/refbuild/slice.py from {src} (verbatim function definitions); do not commit */"]
        for s in sigs:
            body.append(cut(text, s))
        open(os.path.join(out, name), "w", encoding="utf-8").write("\n\n".join(body) + "\n")


if __name__ == "__main__":
    main()
