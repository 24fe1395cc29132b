// oracle/ref_neighbors_shim.cpp -- TEST INFRASTRUCTURE.
// C-ABI shim around the REFERENCE's own vendored sources (compiled from /root/reference, never copied):
//   cpp_wrappers/cpp_utils/cloud/cloud.{h,cpp}       PointXYZ / PointCloud adaptor
//   cpp_wrappers/cpp_utils/nanoflann/nanoflann.hpp   KD-tree (v0x130)
// It performs the same call sequence as batch_nanoflann_neighbors
// (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-332) for ONE batch element: float KD-tree,
// max leaf 10, radiusSearch(q, r*r, sorted=true), pad value = supports.size().
// neighbors.cpp itself cannot be compiled here (tbb/tbb.h and Eigen headers are not in the image).
#include "cpp_utils/cloud/cloud.h"
#include "cpp_utils/nanoflann/nanoflann.hpp"
#include <vector>
#include <cstdint>
#include <cstring>

typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, PointCloud>, PointCloud, 3> kd_tree_t;

extern "C" int ref_radius_neighbors(const float* queries, int nq, const float* supports, int ns, float radius,
                                    int32_t* out, int max_out /* capacity per query */, int32_t* counts)
{
    PointCloud cloud;
    cloud.pts.resize(ns);
    for (int i = 0; i < ns; ++i) cloud.pts[i] = PointXYZ(supports[3 * i], supports[3 * i + 1], supports[3 * i + 2]);
    nanoflann::KDTreeSingleIndexAdaptorParams tree_params(10);
    kd_tree_t index(3, cloud, tree_params);
    index.buildIndex();
    nanoflann::SearchParams search_params;
    search_params.sorted = true;
    float r2 = radius * radius;
    int max_count = 0;
    for (int i = 0; i < nq; ++i) {
        std::vector<std::pair<size_t, float>> res;
        float q[3] = {queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]};
        size_t n = index.radiusSearch(q, r2, res, search_params);
        counts[i] = (int32_t)n;
        if ((int)n > max_count) max_count = (int)n;
        for (int j = 0; j < max_out; ++j) out[(size_t)i * max_out + j] = j < (int)n ? (int32_t)res[j].first : ns;
    }
    return max_count;
}

// The whole call sequence of batch_nanoflann_neighbors (neighbors.cpp:211-332) for one batch element, as the reference runs it:
// serial loop over the queries, all in-radius neighbours sorted by distance kept per query, then the dense
// [nq][max_count] index matrix padded with supports.size().  Returns max_count; *checksum = sum of the matrix (so that the
// work cannot be optimised away); the caller times the call -- this is the CPU baseline of the neighbour step (bench.py).
extern "C" int ref_batch_neighbors(const float* queries, int nq, const float* supports, int ns, float radius, long long* checksum)
{
    PointCloud cloud;
    cloud.pts.resize(ns);
    for (int i = 0; i < ns; ++i) cloud.pts[i] = PointXYZ(supports[3 * i], supports[3 * i + 1], supports[3 * i + 2]);
    nanoflann::KDTreeSingleIndexAdaptorParams tree_params(10);
    kd_tree_t index(3, cloud, tree_params);
    index.buildIndex();
    nanoflann::SearchParams search_params;
    search_params.sorted = true;
    const float r2 = radius * radius;
    size_t max_count = 0;
    std::vector<std::vector<std::pair<size_t, float>>> all((size_t)nq);
    for (int i = 0; i < nq; ++i) {
        all[i].reserve(max_count);
        float q[3] = {queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]};
        const size_t n = index.radiusSearch(q, r2, all[i], search_params);
        if (n > max_count) max_count = n;
    }
    std::vector<int> nb((size_t)nq * max_count);
    long long cs = 0;
    for (int i = 0; i < nq; ++i)
        for (size_t j = 0; j < max_count; ++j) {
            const int v = j < all[i].size() ? (int)all[i][j].first : ns;
            nb[(size_t)i * max_count + j] = v;
            cs += v;
        }
    if (checksum) *checksum = cs;
    return (int)max_count;
}
