// orbfe_cv_stub.h -- the handful of OpenCV types the ORB front-end's public signatures mention, for building
// and testing the shim where OpenCV is absent (this container, the GPU box).  With real OpenCV define
// ORBFE_WITH_OPENCV and this file is never included.  Only what the shim touches is modelled.
#pragma once
#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {
struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {  // field order of cv::KeyPoint (28 bytes)
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int /*type*/, void *ext, size_t stp) : rows(r), cols(c), step(stp), data((uint8_t *)ext) {}
    void create(int r, int c, int /*type*/)
    {
        if (r == rows && c == cols && own_) return;
        own_.reset(new std::vector<uint8_t>((size_t)r * c));
        rows = r; cols = c; step = (size_t)c; data = own_->data();
    }
    void release() { own_.reset(); rows = cols = 0; step = 0; data = nullptr; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (size_t)cols; }
    template <typename T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    uint8_t *ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r = 0) const { return data + (size_t)r * step; }
    Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.step = step; m.data = data + (size_t)r * step; m.own_ = own_; return m; }
    Mat roi(int x, int y, int w, int h) const { Mat m; m.rows = h; m.cols = w; m.step = step; m.data = data + (size_t)y * step + x; m.own_ = own_; return m; }
    Mat getMat() const { return *this; }
private:
    std::shared_ptr<std::vector<uint8_t>> own_;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
