// host_gates.h -- the float 3-D arithmetic that precedes GetFeaturesInArea in the projection overloads
// (ORBmatcher.cc:411-415, 450-495, 1780-1845, 1967-2030, 876-884).  It stays on the host (O(queries), a few
// flops each) and is written the way OpenCV evaluates the reference's cv::Mat expressions:
//   A*x + b   -> one gemm: every output element is accumulated in double and rounded to float once;
//   cv::norm, Mat::dot -> accumulated in double.
// OpenCV is absent from the reference tree and from this image, so this boundary is "parity unpinned"
// (SURVEY.md 8c); the device side only ever sees the resulting (u, v, radius, level) values.
#pragma once
#include <cmath>

namespace orbm_host {

// out = alpha * op(R) * x + c   (R: row-major 3x3 with row stride `ld`; op = transpose if trans)
inline void gemm3(const float *R, int ld, bool trans, const float *x, double alpha, const float *c, float *out) {
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(trans ? R[k * ld + r] : R[r * ld + k]) * (double)x[k];
        s *= alpha;
        if (c) s += (double)c[r];
        out[r] = (float)s;
    }
}

inline float norm3(const float *v) {
    return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
}

inline double dot3(const float *a, const float *b) {
    return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2];
}

}  // namespace orbm_host
