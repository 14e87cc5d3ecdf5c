// TEST INFRASTRUCTURE -- never linked into the product.
// C-ABI driver around the reference's own grid_subsampling() so that the oracle's restatement
// (oracle/grid_subsample_oracle.py) and the golden fixtures can be checked against the real thing.
// The reference sources are compiled where they lie under /root/reference (see oracle/Makefile, target _ref):
//   cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp   (the algorithm, :9-110)
//   cpp_wrappers/cpp_utils/cloud/cloud.cpp                               (min_point / max_point)
// This file only marshals flat arrays in and out (what wrapper.cpp:200-290 does through numpy) and, because
// the reference emits voxels in unordered_map order, also returns each voxel's grid key so callers can
// put both sides in one canonical order.
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

#include "cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.h"

extern "C" int gridsub_ref(const float* points, const float* features, const int* classes, long long n, int fdim, int ldim,
                           float sampleDl, float* out_points, float* out_features, int* out_classes) {
    std::vector<PointXYZ> original(n), sub;
    for (long long i = 0; i < n; ++i) original[i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    std::vector<float> f_in, f_out;
    std::vector<int> c_in, c_out;
    if (features && fdim > 0) f_in.assign(features, features + (size_t)n * fdim);
    if (classes && ldim > 0) c_in.assign(classes, classes + (size_t)n * ldim);
    grid_subsampling(original, sub, f_in, f_out, c_in, c_out, sampleDl, 0);
    const size_t m = sub.size();
    for (size_t i = 0; i < m; ++i) { out_points[3 * i] = sub[i].x; out_points[3 * i + 1] = sub[i].y; out_points[3 * i + 2] = sub[i].z; }
    if (!f_out.empty()) std::memcpy(out_features, f_out.data(), f_out.size() * sizeof(float));
    if (!c_out.empty()) std::memcpy(out_classes, c_out.data(), c_out.size() * sizeof(int));
    return (int)m;
}
