// cvlite.h -- the few OpenCV types the matcher's interface mentions (cv::Mat, cv::KeyPoint, cv::Point2f),
// as plain header-only stand-ins with OpenCV's member names.  OpenCV is not installed in this image; a
// build against the real library defines ORBMATCH_USE_OPENCV and gets <opencv2/core/core.hpp> instead
// (the host layer only uses rows/cols/ptr<T>/at<T>/row/clone/empty, which both provide).
#pragma once
#ifdef ORBMATCH_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

struct Point2f {
    float x = 0, y = 0;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; esz_ = type == CV_8U ? 1 : 4;
        buf_ = std::make_shared<std::vector<uint8_t>>((size_t)r * c * esz_, 0);
        off_ = 0;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type) {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f;
        return m;
    }
    int type() const { return type_; }
    bool empty() const { return !buf_ || rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(buf_->data() + off_ + (size_t)r * cols * esz_); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(buf_->data() + off_ + (size_t)r * cols * esz_); }
    template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T &at(int r, int c) const { return ptr<T>(r)[c]; }
    template <typename T> T &at(int i) { return ptr<T>(0)[i]; }
    template <typename T> const T &at(int i) const { return ptr<T>(0)[i]; }
    Mat row(int r) const {  // shares storage, like cv::Mat::row
        Mat m; m.rows = 1; m.cols = cols; m.type_ = type_; m.esz_ = esz_; m.buf_ = buf_;
        m.off_ = off_ + (size_t)r * cols * esz_;
        return m;
    }
    Mat clone() const {
        Mat m(rows, cols, type_);
        if (!empty()) std::memcpy(m.buf_->data(), buf_->data() + off_, (size_t)rows * cols * esz_);
        return m;
    }

private:
    int type_ = CV_8U, esz_ = 1;
    size_t off_ = 0;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

}  // namespace cv
#endif
