// Replica of scikit-learn's KDTree (leaf_size 30) node order on the device.
//
// The reference snaps every voxel centroid to a real point with
//   NearestNeighbors(n_neighbors=1, algorithm='kd_tree').fit(points)
//     .kneighbors(centroids)                        (models/graph_gen.py:84-88)
// and a centroid is often exactly equidistant from two points (a 2-point voxel:
// centroid = exact midpoint).  Which of the tied points comes back is decided
// by the ORDER in which sklearn's depth-first query meets them
// (sklearn/neighbors/_binary_tree.pxi.tp `_query_single_depthfirst`: the heap
// keeps the first of equal distances; leaves are scanned in idx_array order;
// of two children the one with the smaller min_rdist is entered first, the
// left one on a tie).  That order is a function of
//   * idx_array  -- the permutation `std::nth_element` (libstdc++ introselect,
//     comparator (value, index), sklearn/neighbors/_partition_nodes.pyx) leaves
//     behind, node by node, split dimension = largest spread;
//   * node_bounds -- per-node min/max.
// kdtree.hip rebuilds both bit for bit (tests compare with
// KDTree.get_arrays()); this header holds the query-side rule.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace pgnn {

constexpr int kKdLeafSize = 30;  // NearestNeighbors default leaf_size

// sklearn: n_levels = int(log2(max(1, (n - 1) / leaf_size)) + 1),
// n_nodes = 2**n_levels - 1  (_binary_tree.pxi.tp:876-878)
inline void kd_shape(int64_t n, int *n_levels, int *n_nodes) {
  double x = n > 0 ? (double)(n - 1) / (double)kKdLeafSize : 1.0;
  if (x < 1.0) x = 1.0;
  int lv = 0;
  while ((double)((int64_t)1 << (lv + 1)) <= x) ++lv;  // floor(log2(x))
  *n_levels = lv + 1;
  *n_nodes = (1 << (lv + 1)) - 1;
}

struct KdView {
  const int32_t *pos;     // pos[point] = its slot in idx_array
  const double *bounds;   // [n_nodes][6]: lo x,y,z, hi x,y,z
  int32_t n;
  int32_t n_nodes;
};

#if defined(__HIPCC__)
// _kd_tree.pyx.tp min_rdist (p = 2): sum_j (0.5*((lo-c)+|lo-c| + (c-hi)+|c-hi|))^2
__device__ __forceinline__ double kd_min_rdist(const double *b, double cx,
                                               double cy, double cz) {
  const double c[3] = {cx, cy, cz};
  double r = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double d_lo = b[j] - c[j];
    const double d_hi = c[j] - b[3 + j];
    const double d = (d_lo + fabs(d_lo)) + (d_hi + fabs(d_hi));
    const double h = 0.5 * d;
    r += h * h;
  }
  return r;
}

// true when the point in idx_array slot `pa` is met before the one in slot
// `pb` by the depth-first 1-NN query for (cx, cy, cz)
__device__ __forceinline__ bool kd_met_before(const KdView &kd, int pa, int pb,
                                              double cx, double cy, double cz) {
  int node = 0, s = 0, e = kd.n;
  while (2 * node + 1 < kd.n_nodes) {
    const int m = s + (e - s) / 2;
    const bool a_left = pa < m, b_left = pb < m;
    if (a_left == b_left) {
      node = 2 * node + (a_left ? 1 : 2);
      if (a_left) e = m; else s = m;
      continue;
    }
    const double lb1 = kd_min_rdist(kd.bounds + 6 * (size_t)(2 * node + 1), cx, cy, cz);
    const double lb2 = kd_min_rdist(kd.bounds + 6 * (size_t)(2 * node + 2), cx, cy, cz);
    const bool left_first = lb1 <= lb2;
    return a_left == left_first;
  }
  return pa < pb;  // same leaf: idx_array order
}
#endif

// kdtree.hip
size_t kd_workspace_bytes(int64_t n);
struct KdBuild {
  int32_t *idx;     // idx_array [n]
  int32_t *pos;     // inverse permutation [n]
  double *bounds;   // [n_nodes][6]
  int32_t *status;  // != 0: introselect hit its depth limit (heap-select path
                    // of libstdc++ not replicated; order then falls back)
  int n_levels, n_nodes;
};

struct Arena;
int kd_build(const float *pts, int64_t n, Arena &a, KdBuild &kd,
             hipStream_t stream);

}  // namespace pgnn
