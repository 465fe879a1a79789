// Graph construction on the device: counting-sort spatial hash + fixed-radius
// search (models/graph_gen.py:197-220) and voxel keypoint selection
// (graph_gen.py:41-45, 84-88, 92-153).
//
// Points are bucketed by hashing their integer cell coordinates
// (cell edge = search radius, resp. voxel size) into 2^b buckets and stably
// sorted by bucket (sort.hip), so every bucket is a contiguous, index-ordered
// slice.  A query sweeps the cells that can contain a neighbour, visits each
// cell's bucket slice with the 64 lanes of a wave (coalesced 32-byte records),
// keeps candidates whose own cell equals the swept cell (hash collisions and
// double visits drop out here) and applies the reference's predicate in
// float64:  ((px-cx)^2 + (py-cy)^2) + (pz-cz)^2 <= r*r  -- three separately
// rounded products, inclusive, exactly scikit-learn's ball-tree test on
// float64 data.  This file is compiled with -ffp-contract=off.
//
// HBM traffic is a few MB per frame; these kernels are latency- and
// launch-bound, not bandwidth-bound (DESIGN.md "Graph kernels").
#include <math.h>

#include <mutex>
#include <vector>

#include "kdtree.h"
#include "sort.h"

namespace pgnn {
namespace {

struct __attribute__((aligned(32))) SortedPoint {
  double x, y, z;
  int32_t idx;
  int32_t pad;
};

__device__ __forceinline__ uint32_t cell_hash(int cx, int cy, int cz,
                                              uint32_t mask) {
  uint32_t h = (uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^
               (uint32_t)cz * 83492791u;
  h ^= h >> 15;  // fold the high bits into the bucket range
  return h & mask;
}

// Cell coordinate of a float64 position in a grid of edge `cell` anchored at
// `origin`.  The same function is used to build and to query, so membership
// tests are exact regardless of rounding.
__device__ __forceinline__ int cell_of(double v, double origin, double cell) {
  return (int)floor((v - origin) / cell);
}

// The radius graphs' own cell grid (anchored at 0, edge = r): the cell of a
// coordinate is floor(v * (1 / r)).  Which cell a point near a face lands in
// is immaterial -- candidates are accepted by the exact float64 distance --
// as long as the build and the queries use this one function (a candidate of
// another cell sharing the bucket is told apart by it) and it is monotone in
// v (every point of [c - rr, c + rr] then lies in the box the query scans);
// a multiplication is both, and costs a tenth of the float64 division.
__device__ __forceinline__ int rcell_of(double v, double inv_cell) {
  return (int)floor(v * inv_cell);
}

// NumPy's floor_divide for floating point (npy_divmod in
// numpy/core/src/npymath/npy_math_internal.h.src), divisor > 0: the exact
// floor of a / b for the given operands -- fmod is exact, so unlike
// floor(a / b) the result never jumps a cell when the rounded quotient lands
// on an integer.
__device__ __forceinline__ float npy_floor_divide_f32(float a, float b) {
  float mod = fmodf(a, b);
  float div = (a - mod) / b;
  if (mod != 0.0f && mod < 0.0f) div -= 1.0f;  // sign of b (> 0) != sign of mod
  if (div == 0.0f) return 0.0f;
  float fl = floorf(div);
  if (div - fl > 0.5f) fl += 1.0f;
  return fl;
}
__device__ __forceinline__ double npy_floor_divide_f64(double a, double b) {
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && mod < 0.0) div -= 1.0;
  if (div == 0.0) return 0.0;
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

// Voxel index of coordinate v along axis ax under the keypoint grid's rule
// (grid_origin_kernel writes the record):
//   rule[0..2] origin, rule[3..5] jitter, rule[6..8] cloud minimum, rule[9] mode
//   mode 0: floor((v - origin) / voxel) in float64       ('center': open3d 0.7)
//   float32 cloud (v is a float32 value held as double):
//   mode 1: graph_gen.py:123-124   (points - offset) // voxel, ALL float32:
//           float32 subtraction, NumPy floor_divide with float32(voxel)
//   mode 2: graph_gen.py:126-128   (points - offset + voxel * rnd) // voxel:
//           the float32 difference is promoted, the rest is float64
//   float64 cloud (what train.py:88-90 passes after the augmentations):
//   mode 3: graph_gen.py:123-124 in float64 throughout
//   mode 4: graph_gen.py:126-128 in float64 throughout, (v - min) + jitter
__device__ __forceinline__ int vox_cell(const double *__restrict__ rule, int ax,
                                        double v, double voxel) {
  const int mode = (int)rule[9];
  if (mode == 0) return cell_of(v, rule[ax], voxel);
  if (mode >= 3) {
    const double a = v - rule[6 + ax];
    return (int)npy_floor_divide_f64(mode == 3 ? a : a + rule[3 + ax], voxel);
  }
  const float a = (float)v - (float)rule[6 + ax];
  if (mode == 1) return (int)npy_floor_divide_f32(a, (float)voxel);
  return (int)npy_floor_divide_f64((double)a + rule[3 + ax], voxel);
}
constexpr int kVoxRuleDoubles = 10;

struct Scale3 {
  double x, y, z;
  int on;
};

// T = float (run.py:219-222 feeds the float32 cloud) or double (the training
// path: the augmented cloud stays float64 through graph generation,
// train.py:88-90 -- every predicate below is float64 either way, only the
// load differs)
template <typename T>
__device__ __forceinline__ void load_point(const T *p, int64_t i,
                                           const Scale3 &sc, double &x,
                                           double &y, double &z) {
  x = (double)p[3 * i];
  y = (double)p[3 * i + 1];
  z = (double)p[3 * i + 2];
  if (sc.on) {  // graph_gen.py:203-206: float32 array / float64 scale
    x = x / sc.x;
    y = y / sc.y;
    z = z / sc.z;
  }
}

// ---- build -------------------------------------------------------------------
template <typename T>
__global__ void cell_keys_kernel(const T *__restrict__ pts, int64_t n,
                                 Scale3 sc, double ox, double oy, double oz,
                                 const double *__restrict__ origin_dev,
                                 double cell, uint32_t mask,
                                 uint32_t *__restrict__ keys,
                                 uint32_t *__restrict__ vals,
                                 int32_t *__restrict__ vcell,
                                 const int32_t *__restrict__ n_dev) {
  graph_prio();
  // n_dev (nullable): the point count when only the device knows it (the
  // keypoints of this frame); n is then the capacity the grid is sized for
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  const double inv_cell = 1.0 / cell;  // radius grids (rcell_of)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double x, y, z;
    load_point(pts, i, sc, x, y, z);
    if (origin_dev) {  // keypoint grid: the rule record of grid_origin_kernel
      // the voxel index is evaluated ONCE per point (NumPy's floor_divide is
      // an fmod; the leader / pick kernels compare the stored integers)
      const int vx = vox_cell(origin_dev, 0, x, cell),
                vy = vox_cell(origin_dev, 1, y, cell),
                vz = vox_cell(origin_dev, 2, z, cell);
      keys[i] = cell_hash(vx, vy, vz, mask);
      vcell[3 * i] = vx;
      vcell[3 * i + 1] = vy;
      vcell[3 * i + 2] = vz;
    } else {
      keys[i] = cell_hash(rcell_of(x - ox, inv_cell), rcell_of(y - oy, inv_cell),
                          rcell_of(z - oz, inv_cell), mask);
    }
    vals[i] = (uint32_t)i;
  }
}

template <typename T>
__global__ void cell_bounds_kernel(const uint32_t *__restrict__ keys,
                                   const uint32_t *__restrict__ vals, int64_t n,
                                   const T *__restrict__ pts, Scale3 sc,
                                   int32_t *__restrict__ cell_start,
                                   int32_t *__restrict__ cell_end,
                                   SortedPoint *__restrict__ sorted,
                                   const int32_t *__restrict__ vcell,
                                   int32_t *__restrict__ vcell_sorted,
                                   const int32_t *__restrict__ n_dev) {
  graph_prio();
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) cell_start[k] = (int32_t)i;
    if (i == n - 1 || keys[i + 1] != k) cell_end[k] = (int32_t)(i + 1);
    SortedPoint sp;
    const uint32_t idx = vals[i];
    load_point(pts, idx, sc, sp.x, sp.y, sp.z);
    sp.idx = (int32_t)idx;
    sp.pad = 0;
    sorted[i] = sp;
    if (vcell) {
      vcell_sorted[3 * i] = vcell[3 * (int64_t)idx];
      vcell_sorted[3 * i + 1] = vcell[3 * (int64_t)idx + 1];
      vcell_sorted[3 * i + 2] = vcell[3 * (int64_t)idx + 2];
    }
  }
}

// ---- radius search -------------------------------------------------------------
// One wave per centre (a wave takes centres wave_id, + total waves, ...).
// FILL = false: write the neighbour count; FILL = true: write (point, centre)
// rows at offsets[centre].
//
// The cells that can hold a neighbour are a box of at most 4 x 4 x 4 cells of
// edge r around the centre (3 per axis, 4 when the centre sits within 1e-9 r of
// a cell face).  Lane c looks up cell c's bucket -- all (start, end) pairs in
// ONE round of loads -- and a wave scan turns the bucket sizes into offsets of
// one concatenated candidate list, which the wave then walks 64 candidates at
// a time (cell order z, y, x and slot order inside a bucket: the order of the
// nested loops this replaces, so rows come out in the same order).  The
// nested form made ~2 dependent memory round trips per cell, 54+ per centre;
// this one makes 2 plus one per 64 candidates -- the kernel is a chain of such
// round trips, and beside the message passing of another frame each of them
// takes several times longer.
template <bool FILL, typename T>
__global__ __launch_bounds__(1024) void radius_query_kernel(
    const T *__restrict__ centers, int64_t n_centers, Scale3 sc, double r,
    uint32_t mask, const int32_t *__restrict__ cell_start,
    const int32_t *__restrict__ cell_end,
    const SortedPoint *__restrict__ sorted, int32_t *__restrict__ counts,
    const int32_t *__restrict__ offsets, int32_t *__restrict__ edges,
    int64_t capacity, const int32_t *__restrict__ n_centers_dev) {
  graph_prio();
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int64_t wave0 = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * wpb;
  const int64_t cap = n_centers;
  if (n_centers_dev)
    // capacity form: n_centers is the capacity, the count is on the device
    n_centers = *n_centers_dev < cap ? *n_centers_dev : cap;
  const double r2 = r * r;
  const double inv_r = 1.0 / r;  // as in cell_keys_kernel
  // cells that can hold a point within r: widen by a relative 1e-9 so that
  // rounding in c -/+ r can never exclude a true neighbour
  const double rr = r * (1.0 + 1e-9) + 1e-300;
  for (int64_t q = wave0; q < cap; q += n_waves) {
    if (q >= n_centers) {
      // the slots behind the count get a zero so that the scan over the whole
      // capacity leaves the edge total in its last entry
      if (!FILL && lane == 0) counts[q] = 0;
      continue;
    }
    double cx, cy, cz;
    load_point(centers, q, sc, cx, cy, cz);
    const int x0 = rcell_of(cx - rr, inv_r), x1 = rcell_of(cx + rr, inv_r);
    const int y0 = rcell_of(cy - rr, inv_r), y1 = rcell_of(cy + rr, inv_r);
    const int z0 = rcell_of(cz - rr, inv_r), z1 = rcell_of(cz + rr, inv_r);
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int n_cells = nx * ny * nz;  // <= 64 (see above)
    // lane c: cell c of the box in (z, y, x) order
    int my_s = 0, my_cnt = 0, my_ix = 0, my_iy = 0, my_iz = 0;
    if (lane < n_cells) {
      my_ix = x0 + lane % nx;
      my_iy = y0 + (lane / nx) % ny;
      my_iz = z0 + lane / (nx * ny);
      const uint32_t b = cell_hash(my_ix, my_iy, my_iz, mask);
      my_s = cell_start[b];
      my_cnt = cell_end[b] - my_s;
    }
    int incl = my_cnt;  // inclusive scan of the bucket sizes over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    const int total_cand = __shfl(incl, 63);
    int64_t out_pos = FILL ? (int64_t)offsets[q] : 0;
    int total = 0;
    // Two chunks of 64 candidates per round: their searches and loads are
    // independent, so the second point load is in flight while the first is
    // being waited for (the round is one memory round trip either way, and a
    // centre next to the sensor has 20-30 chunks).  The loads carry no
    // predicate: a lane past the end re-reads the last candidate (always a
    // real one) and is masked at `hit` -- a load under a predicate is waited
    // for on the spot.  Rows leave in candidate order as before.
    for (int j0 = 0; j0 < total_cand; j0 += 128) {
      const bool two = j0 + 64 < total_cand;  // wave-uniform
      bool valid[2];
      int ix[2], iy[2], iz[2];
      SortedPoint sp[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        // every lane runs the search and the shuffles (a shuffle may only
        // read lanes that execute it)
        valid[u] = j0 + 64 * u + lane < total_cand;
        const int j = valid[u] ? j0 + 64 * u + lane : total_cand - 1;
        // the cell whose candidate range holds j: first lane with incl > j
        int lo = 0, hi = n_cells - 1;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
          const int mid = (lo + hi) >> 1;
          const int v = __shfl(incl, mid);
          if (v > j) hi = mid; else lo = mid + 1;
        }
        const int c = lo < n_cells ? lo : n_cells - 1;
        const int c_incl = __shfl(incl, c), c_cnt = __shfl(my_cnt, c);
        const int c_s = __shfl(my_s, c);
        ix[u] = __shfl(my_ix, c);
        iy[u] = __shfl(my_iy, c);
        iz[u] = __shfl(my_iz, c);
        sp[u] = sorted[c_s + (j - (c_incl - c_cnt))];
      }
      unsigned long long m[2] = {0ull, 0ull};
      bool hit[2] = {false, false};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        if (valid[u] && rcell_of(sp[u].x, inv_r) == ix[u] &&
            rcell_of(sp[u].y, inv_r) == iy[u] &&
            rcell_of(sp[u].z, inv_r) == iz[u]) {
          const double dx = sp[u].x - cx, dy = sp[u].y - cy,
                       dz = sp[u].z - cz;
          const double d2 = (dx * dx + dy * dy) + dz * dz;
          hit[u] = d2 <= r2;
        }
        m[u] = __ballot(hit[u]);
      }
      if (FILL) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (hit[u]) {
            const int64_t pos =
                out_pos + __popcll(m[u] & ((1ull << lane) - 1ull));
            if (pos < capacity) {
              edges[2 * pos] = sp[u].idx;
              edges[2 * pos + 1] = (int32_t)q;
            }
          }
          out_pos += __popcll(m[u]);
        }
      } else {
        total += __popcll(m[0]) + __popcll(m[1]);
      }
    }
    if (!FILL && lane == 0) counts[q] = total;
  }
}

// ---- neighbour cap (training) ---------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

__global__ void cap_counts_kernel(const int32_t *__restrict__ offsets,
                                  int64_t n, int32_t cap,
                                  int32_t *__restrict__ counts) {
  graph_prio();
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q > n) return;
  if (q == n) {  // the scan over n + 1 entries leaves the total here
    counts[n] = 0;
    return;
  }
  int c = offsets[q + 1] - offsets[q];
  counts[q] = c > cap ? cap : c;
}

// capacity form: the capped list's record {rows written, rows required}.  An
// uncapped list that did not fit ITS capacity makes the capped one unusable
// whatever its size: rows written = 0 < rows required flags the level.
__global__ void cap_total_kernel(const int32_t *__restrict__ new_total,
                                 const int32_t *__restrict__ n_in,
                                 int64_t capacity, int32_t *__restrict__ n_out) {
  graph_prio();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int32_t e = *new_total;
    const bool in_ok = n_in[1] <= n_in[0];
    n_out[0] = !in_ok ? 0 : ((int64_t)e < capacity ? e : (int32_t)capacity);
    n_out[1] = e;
  }
}

// One wave per centre.  Fan-in <= cap: copy.  Otherwise keep the `cap` edges
// with the smallest 64-bit hash(seed, centre, position): a uniformly random
// subset without replacement (np.random.choice(..., replace=False) semantics,
// graph_gen.py:210-214), selected by counting, emitted in original order.
__global__ __launch_bounds__(256) void cap_fill_kernel(
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ edges,
    int64_t n, int32_t cap, uint64_t seed,
    const int32_t *__restrict__ new_offsets, int32_t *__restrict__ new_edges,
    int64_t capacity, int64_t in_capacity) {
  graph_prio();
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n) return;
  const int s = offsets[q], e = offsets[q + 1], cnt = e - s;
  if ((int64_t)e > in_capacity) return;  // rows the uncapped list does not hold
  int64_t o = new_offsets[q];
  if (cnt <= cap) {
    for (int i = lane; i < cnt; i += 64) {
      if (o + i < capacity) {
        new_edges[2 * (o + i)] = edges[2 * (int64_t)(s + i)];
        new_edges[2 * (o + i) + 1] = edges[2 * (int64_t)(s + i) + 1];
      }
    }
    return;
  }
  const uint64_t base = mix64(seed ^ mix64((uint64_t)q + 0x9e3779b97f4a7c15ull));
  for (int i0 = 0; i0 < cnt; i0 += 64) {
    const int i = i0 + lane;
    bool keep = false;
    if (i < cnt) {
      const uint64_t hi = mix64(base + (uint64_t)i);
      int rank = 0;  // number of edges with a smaller (hash, position) key
      for (int j = 0; j < cnt; ++j) {
        const uint64_t hj = mix64(base + (uint64_t)j);
        rank += (hj < hi) || (hj == hi && j < i);
      }
      keep = rank < cap;
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int64_t pos = o + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < capacity) {
        new_edges[2 * pos] = edges[2 * (int64_t)(s + i)];
        new_edges[2 * pos + 1] = edges[2 * (int64_t)(s + i) + 1];
      }
    }
    o += __popcll(m);
  }
}

// ---- keypoints --------------------------------------------------------------------
// order-preserving map double -> uint64 (float32 inputs are widened first: the
// widening is exact and monotone, so the minimum is the float32 minimum)
__device__ __forceinline__ unsigned long long double_to_ordered(double f) {
  unsigned long long u = (unsigned long long)__double_as_longlong(f);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_to_double(unsigned long long u) {
  return __longlong_as_double((long long)(
      (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u));
}

template <typename T>
__global__ void min_bound_kernel(const T *__restrict__ pts, int64_t n,
                                 unsigned long long *__restrict__ ordered_min) {
  graph_prio();
  unsigned long long m0 = ~0ull, m1 = ~0ull, m2 = ~0ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long a = double_to_ordered((double)pts[3 * i]);
    const unsigned long long b = double_to_ordered((double)pts[3 * i + 1]);
    const unsigned long long c = double_to_ordered((double)pts[3 * i + 2]);
    m0 = a < m0 ? a : m0;
    m1 = b < m1 ? b : m1;
    m2 = c < m2 ? c : m2;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long a = (unsigned long long)__shfl_xor((long long)m0, d);
    const unsigned long long b = (unsigned long long)__shfl_xor((long long)m1, d);
    const unsigned long long c = (unsigned long long)__shfl_xor((long long)m2, d);
    m0 = a < m0 ? a : m0;
    m1 = b < m1 ? b : m1;
    m2 = c < m2 ? c : m2;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&ordered_min[0], m0);
    atomicMin(&ordered_min[1], m1);
    atomicMin(&ordered_min[2], m2);
  }
}

// origin[0..2] = grid origin, from the cloud minimum:
//   center mode (open3d 0.7): min - voxel/2
//   random mode (graph_gen.py:108-128): min - jitter
// mode: see vox_cell
__global__ void grid_origin_kernel(
    const unsigned long long *__restrict__ ordered_min, double sub_x,
    double sub_y, double sub_z, int mode, double *__restrict__ origin) {
  graph_prio();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double sub[3] = {sub_x, sub_y, sub_z};
    for (int ax = 0; ax < 3; ++ax) {
      const double mn = ordered_to_double(ordered_min[ax]);
      origin[ax] = mn - sub[ax];
      origin[3 + ax] = sub[ax];
      origin[6 + ax] = mn;
    }
    origin[9] = (double)mode;
  }
}

// Thread per sorted slot.  The first slot of a voxel inside its bucket (the
// "leader") sums the voxel's points in index order -- the sort is stable, so
// this is open3d's accumulation order -- and stores the float64 mean.
__global__ void voxel_leader_kernel(const SortedPoint *__restrict__ sorted,
                                    const uint32_t *__restrict__ keys, int64_t n,
                                    const int32_t *__restrict__ vc,
                                    const int32_t *__restrict__ cell_start,
                                    const int32_t *__restrict__ cell_end,
                                    int32_t *__restrict__ is_leader,
                                    double *__restrict__ centroid,
                                    int32_t *__restrict__ member_count) {
  graph_prio();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int vx = vc[3 * i], vy = vc[3 * i + 1], vz = vc[3 * i + 2];
    const uint32_t b = keys[i];
    const int s = cell_start[b], e = cell_end[b];
    // Leader = the first slot of this voxel in its bucket.  A bucket nearly
    // always holds ONE voxel (>= 2 buckets per point): a slot whose voxel is
    // the bucket's first voxel is a leader iff it is the first slot -- no
    // search; only the members of a colliding second voxel walk the slots
    // before them (the walk made this kernel O(members^2) per voxel: 300 us at
    // voxel 0.8 m)
    bool leader;
    if (vc[3 * s] == vx && vc[3 * s + 1] == vy && vc[3 * s + 2] == vz) {
      leader = (int)i == s;
    } else {
      leader = true;
      for (int j = s; j < (int)i; ++j) {
        if (vc[3 * j] == vx && vc[3 * j + 1] == vy && vc[3 * j + 2] == vz) {
          leader = false;
          break;
        }
      }
    }
    is_leader[i] = leader ? 1 : 0;
    if (!leader) continue;
    // The sum must run in slot order (open3d's accumulation order), but the
    // loads need not wait for each other: four slots' voxel indices and
    // records are requested together, without a predicate (every j < e is a
    // real slot), and a slot of another voxel is dropped by a select.  The
    // launch lasts as long as its fullest voxel (next to the sensor: 50-100
    // members), which used to be two dependent round trips per member.
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int cnt = 0;
    int j = (int)i;
    for (; j + 4 <= e; j += 4) {
      int ax[4], ay[4], az[4];
      SortedPoint o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ax[u] = vc[3 * (j + u)];
        ay[u] = vc[3 * (j + u) + 1];
        az[u] = vc[3 * (j + u) + 2];
        o[u] = sorted[j + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool m = ax[u] == vx && ay[u] == vy && az[u] == vz;
        sx = m ? sx + o[u].x : sx;
        sy = m ? sy + o[u].y : sy;
        sz = m ? sz + o[u].z : sz;
        cnt += m ? 1 : 0;
      }
    }
    for (; j < e; ++j) {
      const int ax = vc[3 * j], ay = vc[3 * j + 1], az = vc[3 * j + 2];
      const SortedPoint o = sorted[j];
      const bool m = ax == vx && ay == vy && az == vz;
      sx = m ? sx + o.x : sx;
      sy = m ? sy + o.y : sy;
      sz = m ? sz + o.z : sz;
      cnt += m ? 1 : 0;
    }
    centroid[3 * i] = sx / (double)cnt;
    centroid[3 * i + 1] = sy / (double)cnt;
    centroid[3 * i + 2] = sz / (double)cnt;
    member_count[i] = cnt;
  }
}

// Exact float64 1-NN of every voxel centroid among all points
// (graph_gen.py:84-88).  The nearest point is closer than 0.87 voxel edges (it
// is at most the RMS spread of the voxel's own points away), so the 27
// surrounding voxels suffice.  Exact distance ties (every 2-point voxel) go to
// the point scikit-learn's kd-tree query meets first (kdtree.h), which is the
// reference's pick.
// A wave takes `spw` sorted slots at a time (4 when the grid may be as large as
// it likes -- about one leader per wave --, 64 when the frame pipeline caps the
// grid) and serves the leaders among them one after the other: lane c looks up bucket c of the leader's 27 voxels in ONE
// round of loads, a scan concatenates the buckets, and the wave walks the
// candidates 64 at a time (see radius_query_kernel).  The winner does not
// depend on the order candidates are met in: the (distance, kd order) relation
// is a strict total order.
__global__ __launch_bounds__(1024) void voxel_nn_kernel(
    const SortedPoint *__restrict__ sorted, int64_t n,
    const double *__restrict__ origin, double voxel, uint32_t mask,
    const int32_t *__restrict__ cell_start, const int32_t *__restrict__ cell_end,
    const int32_t *__restrict__ is_leader, const int32_t *__restrict__ slot,
    const double *__restrict__ centroid, const float *__restrict__ pts,
    KdView kd, int32_t *__restrict__ kp_idx, float *__restrict__ kp_xyz,
    int spw /* sorted slots a wave takes at a time, <= 64 */) {
  graph_prio();
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * wpb;
  const double ox = origin[0], oy = origin[1], oz = origin[2];
  for (int64_t base = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * spw;
       base < n; base += n_waves * spw) {
    unsigned long long leaders = __ballot(
        lane < spw && base + lane < n && is_leader[base + lane] != 0);
    while (leaders) {
      const int64_t i = base + __builtin_ctzll(leaders);
      leaders &= leaders - 1;
      const double cx = centroid[3 * i], cy = centroid[3 * i + 1],
                   cz = centroid[3 * i + 2];
      const int vx = cell_of(cx, ox, voxel), vy = cell_of(cy, oy, voxel),
                vz = cell_of(cz, oz, voxel);
      int my_s = 0, my_cnt = 0, my_ix = 0, my_iy = 0, my_iz = 0;
      if (lane < 27) {  // (dz, dy, dx) order
        my_ix = vx - 1 + lane % 3;
        my_iy = vy - 1 + (lane / 3) % 3;
        my_iz = vz - 1 + lane / 9;
        const uint32_t b = cell_hash(my_ix, my_iy, my_iz, mask);
        my_s = cell_start[b];
        my_cnt = cell_end[b] - my_s;
      }
      int incl = my_cnt;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      const int total_cand = __shfl(incl, 26);
      double best = 1.0e300;
      int best_idx = 0x7fffffff;
      for (int j0 = 0; j0 < total_cand; j0 += 64) {
        const bool valid = j0 + lane < total_cand;
        const int j = valid ? j0 + lane : total_cand - 1;
        int lo = 0, hi = 26;
#pragma unroll
        for (int step = 0; step < 5; ++step) {
          const int mid = (lo + hi) >> 1;
          const int v = __shfl(incl, mid);
          if (v > j) hi = mid; else lo = mid + 1;
        }
        const int c = lo < 27 ? lo : 26;
        const int c_incl = __shfl(incl, c), c_cnt = __shfl(my_cnt, c);
        const int c_s = __shfl(my_s, c);
        const int ix = __shfl(my_ix, c), iy = __shfl(my_iy, c),
                  iz = __shfl(my_iz, c);
        if (valid) {
          const SortedPoint o = sorted[c_s + (j - (c_incl - c_cnt))];
          if (cell_of(o.x, ox, voxel) == ix && cell_of(o.y, oy, voxel) == iy &&
              cell_of(o.z, oz, voxel) == iz) {
            const double ex = o.x - cx, ey = o.y - cy, ez = o.z - cz;
            const double d2 = (ex * ex + ey * ey) + ez * ez;
            if (d2 < best ||
                (d2 == best && kd_met_before(kd, kd.pos[o.idx],
                                             kd.pos[best_idx], cx, cy, cz))) {
              best = d2;
              best_idx = o.idx;
            }
          }
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const double ob = __shfl_xor(best, d);
        const int oi = __shfl_xor(best_idx, d);
        if (ob < best ||
            (ob == best && oi != best_idx && oi != 0x7fffffff &&
             kd_met_before(kd, kd.pos[oi], kd.pos[best_idx], cx, cy, cz))) {
          best = ob;
          best_idx = oi;
        }
      }
      if (lane == 0) {
        const int k = slot[i];
        kp_idx[k] = best_idx;
        kp_xyz[3 * k] = pts[3 * (int64_t)best_idx];
        kp_xyz[3 * k + 1] = pts[3 * (int64_t)best_idx + 1];
        kp_xyz[3 * k + 2] = pts[3 * (int64_t)best_idx + 2];
      }
    }
  }
}

// voxel_nn_kernel for the SECOND and later pooling levels (graph_gen.py:78-88
// with i > 1): the centroids are still those of the ORIGINAL cloud's voxels,
// but the candidates are the points of another set -- `base`, the previous
// level's keypoints -- so the nearest one need not lie in the 27 surrounding
// voxels.  A wave searches the (2r+1)^3 voxels round the centroid's for r = 1,
// 2, 3 (cell grid built over `base` with the same origin and edge) and stops
// as soon as the best distance is below the distance to the cube's nearest
// face; a centroid still undecided then (a base set far away) is compared with
// every base point.  Ties go to the point scikit-learn's query on the kd-tree
// of `base` meets first (kd); the (distance, kd order) relation is a strict
// total order, so meeting a candidate twice (the cubes are nested, buckets may
// be shared) changes nothing.
__global__ __launch_bounds__(256) void voxel_nn_from_kernel(
    const SortedPoint *__restrict__ sorted_b, int64_t nb,
    const double *__restrict__ origin, double voxel, uint32_t mask_b,
    const int32_t *__restrict__ cell_start_b,
    const int32_t *__restrict__ cell_end_b, int64_t n,
    const int32_t *__restrict__ is_leader, const int32_t *__restrict__ slot,
    const double *__restrict__ centroid, const float *__restrict__ base,
    KdView kd, int32_t *__restrict__ kp_idx, float *__restrict__ kp_xyz) {
  graph_prio();
  constexpr int spw = 4;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * wpb;
  const double ox = origin[0], oy = origin[1], oz = origin[2];
  for (int64_t first = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * spw;
       first < n; first += n_waves * spw) {
    unsigned long long leaders = __ballot(
        lane < spw && first + lane < n && is_leader[first + lane] != 0);
    while (leaders) {
      const int64_t i = first + __builtin_ctzll(leaders);
      leaders &= leaders - 1;
      const double cx = centroid[3 * i], cy = centroid[3 * i + 1],
                   cz = centroid[3 * i + 2];
      const int vx = cell_of(cx, ox, voxel), vy = cell_of(cy, oy, voxel),
                vz = cell_of(cz, oz, voxel);
      double best = 1.0e300;
      int best_idx = 0x7fffffff;
      // candidate (x, y, z, idx): keep it if nearer, or as near and met first
      auto consider = [&](double px, double py, double pz, int idx) {
        const double ex = px - cx, ey = py - cy, ez = pz - cz;
        const double d2 = (ex * ex + ey * ey) + ez * ez;
        if (d2 < best ||
            (d2 == best && idx != best_idx &&
             kd_met_before(kd, kd.pos[idx], kd.pos[best_idx], cx, cy, cz))) {
          best = d2;
          best_idx = idx;
        }
      };
      // every lane ends up with the wave's best
      auto reduce = [&]() {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const double ob = __shfl_xor(best, d);
          const int oi = __shfl_xor(best_idx, d);
          if (ob < best ||
              (ob == best && oi != best_idx && oi != 0x7fffffff &&
               (best_idx == 0x7fffffff ||
                kd_met_before(kd, kd.pos[oi], kd.pos[best_idx], cx, cy, cz)))) {
            best = ob;
            best_idx = oi;
          }
        }
      };
      bool done = false;
      for (int r = 1; r <= 3 && !done; ++r) {
        const int side = 2 * r + 1, count = side * side * side;
        for (int c0 = 0; c0 < count; c0 += 64) {
          const int ci = c0 + lane;
          int my_s = 0, my_cnt = 0, my_ix = 0, my_iy = 0, my_iz = 0;
          if (ci < count) {  // (dz, dy, dx) order
            my_ix = vx - r + ci % side;
            my_iy = vy - r + (ci / side) % side;
            my_iz = vz - r + ci / (side * side);
            const uint32_t b = cell_hash(my_ix, my_iy, my_iz, mask_b);
            my_s = cell_start_b[b];
            my_cnt = cell_end_b[b] - my_s;
          }
          int incl = my_cnt;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
          }
          const int total_cand = __shfl(incl, 63);
          for (int j0 = 0; j0 < total_cand; j0 += 64) {
            const bool valid = j0 + lane < total_cand;
            const int j = valid ? j0 + lane : total_cand - 1;
            int lo = 0, hi = 63;  // first lane whose inclusive count exceeds j
#pragma unroll
            for (int step = 0; step < 6; ++step) {
              const int mid = (lo + hi) >> 1;
              const int v = __shfl(incl, mid);
              if (v > j) hi = mid; else lo = mid + 1;
            }
            const int c = lo < 64 ? lo : 63;
            const int c_incl = __shfl(incl, c), c_cnt = __shfl(my_cnt, c);
            const int c_s = __shfl(my_s, c);
            const int ix = __shfl(my_ix, c), iy = __shfl(my_iy, c),
                      iz = __shfl(my_iz, c);
            if (valid) {
              const SortedPoint o = sorted_b[c_s + (j - (c_incl - c_cnt))];
              if (cell_of(o.x, ox, voxel) == ix && cell_of(o.y, oy, voxel) == iy &&
                  cell_of(o.z, oz, voxel) == iz)
                consider(o.x, o.y, o.z, o.idx);
            }
          }
        }
        reduce();
        // every point outside the cube is at least `face` away
        const double fx0 = cx - (ox + (double)(vx - r) * voxel),
                     fx1 = (ox + (double)(vx + r + 1) * voxel) - cx,
                     fy0 = cy - (oy + (double)(vy - r) * voxel),
                     fy1 = (oy + (double)(vy + r + 1) * voxel) - cy,
                     fz0 = cz - (oz + (double)(vz - r) * voxel),
                     fz1 = (oz + (double)(vz + r + 1) * voxel) - cz;
        double face = fx0 < fx1 ? fx0 : fx1;
        face = fy0 < face ? fy0 : face;
        face = fy1 < face ? fy1 : face;
        face = fz0 < face ? fz0 : face;
        face = fz1 < face ? fz1 : face;
        face -= 1.0e-9 * voxel;  // (which cell a point on a face lands in)
        done = best_idx != 0x7fffffff && face > 0.0 && best < face * face;
      }
      if (!done) {  // undecided: every base point
        for (int64_t j = lane; j < nb; j += 64)
          consider((double)base[3 * j], (double)base[3 * j + 1],
                   (double)base[3 * j + 2], (int)j);
        reduce();
      }
      if (lane == 0) {
        const int k = slot[i];
        kp_idx[k] = best_idx;
        kp_xyz[3 * k] = base[3 * (int64_t)best_idx];
        kp_xyz[3 * k + 1] = base[3 * (int64_t)best_idx + 1];
        kp_xyz[3 * k + 2] = base[3 * (int64_t)best_idx + 2];
      }
    }
  }
}

// random mode: the leader picks member floor(u * count) of its voxel
template <typename T>
__global__ void voxel_random_pick_kernel(
    const SortedPoint *__restrict__ sorted, const uint32_t *__restrict__ keys,
    int64_t n, const int32_t *__restrict__ vc,
    const int32_t *__restrict__ cell_end, const int32_t *__restrict__ is_leader,
    const int32_t *__restrict__ slot, const int32_t *__restrict__ member_count,
    uint64_t seed, const T *__restrict__ pts, int32_t *__restrict__ kp_idx,
    T *__restrict__ kp_xyz) {
  graph_prio();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (!is_leader[i]) continue;
    const int my_idx = sorted[i].idx;
    const int vx = vc[3 * i], vy = vc[3 * i + 1], vz = vc[3 * i + 2];
    const int cnt = member_count[i];
    const uint64_t h =
        mix64(seed ^ mix64((uint64_t)my_idx + 0x632be59bd9b4e019ull));
    int target = (int)((h >> 11) * (1.0 / 9007199254740992.0) * (double)cnt);
    if (target >= cnt) target = cnt - 1;
    const int e = cell_end[keys[i]];
    // the target-th member of the voxel in slot order; four slots' voxel
    // indices per round trip (see voxel_leader_kernel)
    int chosen_slot = (int)i, seen = 0;
    bool found = false;
    int j = (int)i;
    for (; !found && j + 4 <= e; j += 4) {
      int ax[4], ay[4], az[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ax[u] = vc[3 * (j + u)];
        ay[u] = vc[3 * (j + u) + 1];
        az[u] = vc[3 * (j + u) + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool m = !found && ax[u] == vx && ay[u] == vy && az[u] == vz;
        if (m && seen == target) {
          chosen_slot = j + u;
          found = true;
        }
        seen += m ? 1 : 0;
      }
    }
    for (; !found && j < e; ++j) {
      if (vc[3 * j] == vx && vc[3 * j + 1] == vy && vc[3 * j + 2] == vz) {
        if (seen == target) {
          chosen_slot = j;
          found = true;
        }
        ++seen;
      }
    }
    const int chosen = found ? sorted[chosen_slot].idx : my_idx;
    const int k = slot[i];
    kp_idx[k] = chosen;
    kp_xyz[3 * k] = pts[3 * (int64_t)chosen];
    kp_xyz[3 * k + 1] = pts[3 * (int64_t)chosen + 1];
    kp_xyz[3 * k + 2] = pts[3 * (int64_t)chosen + 2];
  }
}

// num[0] = K; num[1] = tie-order status of the kd-tree replica (0 = the
// reference's order, 1 = libstdc++'s heap-select fallback would have run and
// is not replicated: exact ties then follow a different, still deterministic
// order); kd_status null (random mode / ablation): 0
__global__ void copy_total_kernel(const int32_t *src,
                                  const int32_t *kd_status, int32_t *num) {
  graph_prio();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    num[0] = *src;
    num[1] = kd_status ? *kd_status : 0;
  }
}

// hipMemsetAsync for the builder: the runtime's fill kernel picks its own grid,
// which the frame pipeline cannot cap (graph_max_wgs)
__global__ void graph_fill32_kernel(uint32_t *__restrict__ p, uint32_t v,
                                    int64_t count) {
  graph_prio();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
inline int graph_fill32(void *p, uint32_t v, int64_t count, hipStream_t stream) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(graph_fill32_kernel, dim3(graph_grid((count + 1023) / 1024)),
                     dim3(1024), graph_lds_pad(), stream, (uint32_t *)p, v,
                     count);
  return (int)hipGetLastError();
}

// ---- host-side helpers ---------------------------------------------------------------
int hash_bits(int64_t n) {
  int b = 10;
  while (b < 22 && ((int64_t)1 << b) < 2 * n) ++b;
  return b;  // >= 2 buckets per point on average, 2^10 .. 2^22 buckets
}

struct Grid {
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  uint32_t *keys, *vals;  // sorted result
  int32_t *cell_start, *cell_end;
  SortedPoint *sorted;
  void *sort_scratch;
  size_t sort_scratch_bytes;
  uint32_t mask;
  int bits;
};

size_t grid_bytes(int64_t n) {
  const int bits = hash_bits(n);
  size_t b = 0;
  b += 4 * align_up((size_t)n * 4, 256);                     // keys/vals x2
  b += 2 * align_up(((size_t)1 << bits) * 4, 256);           // start/end
  b += align_up((size_t)n * sizeof(SortedPoint), 256);       // sorted copy
  b += align_up(radix_sort_scratch_bytes(n), 256);
  return b + 2048;
}

int grid_carve(Arena &a, int64_t n, Grid &g) {
  g.bits = hash_bits(n);
  g.mask = ((uint32_t)1 << g.bits) - 1u;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  g.keys_a = a.take<uint32_t>(nn);
  g.vals_a = a.take<uint32_t>(nn);
  g.keys_b = a.take<uint32_t>(nn);
  g.vals_b = a.take<uint32_t>(nn);
  g.cell_start = a.take<int32_t>((size_t)1 << g.bits);
  g.cell_end = a.take<int32_t>((size_t)1 << g.bits);
  g.sorted = a.take<SortedPoint>(nn);
  g.sort_scratch_bytes = radix_sort_scratch_bytes(n);
  g.sort_scratch = a.take<char>(g.sort_scratch_bytes);
  if (!g.keys_a || !g.vals_a || !g.keys_b || !g.vals_b || !g.cell_start ||
      !g.cell_end || !g.sorted || !g.sort_scratch)
    return fail(PGNN_E_WORKSPACE, "graph: workspace too small");
  return 0;
}

template <typename T>
int grid_build(const T *pts, int64_t n, const Scale3 &sc, double ox,
               double oy, double oz, const double *origin_dev, double cell,
               Grid &g, hipStream_t stream, int32_t *vcell = nullptr,
               int32_t *vcell_sorted = nullptr,
               const int32_t *n_dev = nullptr) {
  const size_t nbuckets = (size_t)1 << g.bits;
  // cell_start and cell_end are carved back to back (grid_carve): one fill
  PGNN_REQUIRE(g.cell_end == g.cell_start + nbuckets, PGNN_E_INVALID,
               "graph: bucket tables are not contiguous");
  PGNN_HIP((hipError_t)graph_fill32(g.cell_start, 0u, 2 * (int64_t)nbuckets,
                                    stream));
  if (n <= 0) return 0;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(cell_keys_kernel<T>, dim3(graph_grid(blocks)), dim3(256), graph_lds_pad(), stream, pts,
                     n, sc, ox, oy, oz, origin_dev, cell, g.mask, g.keys_a,
                     g.vals_a, vcell, n_dev);
  int rc = radix_sort_pairs(g.keys_a, g.vals_a, g.keys_b, g.vals_b, n, g.bits,
                            g.sort_scratch, g.sort_scratch_bytes, &g.keys,
                            &g.vals, stream, n_dev);
  if (rc) return rc;
  hipLaunchKernelGGL(cell_bounds_kernel<T>, dim3(graph_grid(blocks)), dim3(256), graph_lds_pad(), stream,
                     g.keys, g.vals, n, pts, sc, g.cell_start, g.cell_end,
                     g.sorted, (const int32_t *)vcell, vcell_sorted, n_dev);
  PGNN_HIP(hipGetLastError());
  return 0;
}

Scale3 make_scale(const double *scale3_host) {
  Scale3 s;
  s.on = scale3_host != nullptr;
  s.x = s.on ? scale3_host[0] : 1.0;
  s.y = s.on ? scale3_host[1] : 1.0;
  s.z = s.on ? scale3_host[2] : 1.0;
  return s;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

// workspace layout (radius graph): [Grid | counts[n_centers] | scan scratch]
extern "C" size_t pgnn_radius_graph_workspace_bytes(int64_t n_points,
                                                    int64_t n_centers) {
  if (n_points < 0 || n_centers < 0) return 0;
  return grid_bytes(n_points) + align_up((size_t)(n_centers + 1) * 4, 256) +
         align_up(scan_scratch_bytes(n_centers), 256) + 1024;
}

namespace {
struct RadiusWs {
  Grid g;
  int32_t *counts;
  void *scan_scratch;
  size_t scan_bytes;
};
int radius_carve(void *ws, size_t ws_bytes, int64_t n_points, int64_t n_centers,
                 RadiusWs &r) {
  PGNN_REQUIRE(ws != nullptr &&
                   ws_bytes >= pgnn_radius_graph_workspace_bytes(n_points,
                                                                 n_centers),
               PGNN_E_WORKSPACE, "radius_graph: workspace too small");
  Arena a(ws, ws_bytes);
  int rc = grid_carve(a, n_points, r.g);
  if (rc) return rc;
  r.counts = a.take<int32_t>((size_t)n_centers + 1);
  r.scan_bytes = scan_scratch_bytes(n_centers);
  r.scan_scratch = a.take<char>(r.scan_bytes);
  PGNN_REQUIRE(r.counts && r.scan_scratch, PGNN_E_WORKSPACE,
               "radius_graph: workspace too small");
  return 0;
}
}  // namespace

namespace {
template <typename T>
int radius_count_impl(const T *points, int64_t n_points, const T *centers,
                      int64_t n_centers, double radius,
                      const double *scale3_host, void *workspace,
                      size_t workspace_bytes, int32_t *offsets,
                      hipStream_t stream) {
  PGNN_REQUIRE(n_points >= 0 && n_centers >= 0 && radius > 0.0 &&
                   offsets != nullptr,
               PGNN_E_INVALID, "radius_graph_count: bad argument");
  PGNN_REQUIRE((n_points == 0 || points) && (n_centers == 0 || centers),
               PGNN_E_INVALID, "radius_graph_count: null input");
  RadiusWs w;
  int rc = radius_carve(workspace, workspace_bytes, n_points, n_centers, w);
  if (rc) return rc;
  const Scale3 sc = make_scale(scale3_host);
  rc = grid_build(points, n_points, sc, 0.0, 0.0, 0.0, nullptr, radius, w.g,
                  stream);
  if (rc) return rc;
  if (n_centers > 0) {
    hipLaunchKernelGGL((radius_query_kernel<false, T>),
                       dim3(graph_grid((n_centers + (graph_wide_block() >> 6) - 1) /
                                       (graph_wide_block() >> 6))),
                       dim3(graph_wide_block()), graph_lds_pad(),
                       stream, centers, n_centers, sc, radius, w.g.mask,
                       w.g.cell_start, w.g.cell_end, w.g.sorted, w.counts,
                       (const int32_t *)nullptr, (int32_t *)nullptr,
                       (int64_t)0, (const int32_t *)nullptr);
    PGNN_HIP(hipGetLastError());
  }
  return exclusive_scan_i32(w.counts, offsets, n_centers, w.scan_scratch,
                            w.scan_bytes, stream);
}

template <typename T>
int radius_fill_impl(const T *points, int64_t n_points, const T *centers,
                     int64_t n_centers, double radius,
                     const double *scale3_host, void *workspace,
                     size_t workspace_bytes, const int32_t *offsets,
                     int32_t *edges, int64_t capacity, hipStream_t stream) {
  PGNN_REQUIRE(n_points >= 0 && n_centers >= 0 && radius > 0.0 && offsets &&
                   capacity >= 0,
               PGNN_E_INVALID, "radius_graph_fill: bad argument");
  if (n_centers == 0 || capacity == 0) return 0;
  PGNN_REQUIRE(edges && points && centers, PGNN_E_INVALID,
               "radius_graph_fill: null pointer");
  RadiusWs w;  // same carve as _count: the grid built there is reused
  int rc = radius_carve(workspace, workspace_bytes, n_points, n_centers, w);
  if (rc) return rc;
  const Scale3 sc = make_scale(scale3_host);
  hipLaunchKernelGGL((radius_query_kernel<true, T>),
                     dim3(graph_grid((n_centers + (graph_wide_block() >> 6) - 1) /
                                       (graph_wide_block() >> 6))),
                       dim3(graph_wide_block()), graph_lds_pad(), stream,
                     centers, n_centers, sc, radius, w.g.mask, w.g.cell_start,
                     w.g.cell_end, w.g.sorted, (int32_t *)nullptr, offsets,
                     edges, capacity, (const int32_t *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_radius_graph_count(const float *points, int64_t n_points,
                                       const float *centers, int64_t n_centers,
                                       double radius, const double *scale3_host,
                                       void *workspace, size_t workspace_bytes,
                                       int32_t *offsets, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_count_impl(points, n_points, centers, n_centers, radius,
                           scale3_host, workspace, workspace_bytes, offsets,
                           (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_fill(const float *points, int64_t n_points,
                                      const float *centers, int64_t n_centers,
                                      double radius, const double *scale3_host,
                                      void *workspace, size_t workspace_bytes,
                                      const int32_t *offsets, int32_t *edges,
                                      int64_t capacity, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_fill_impl(points, n_points, centers, n_centers, radius,
                          scale3_host, workspace, workspace_bytes, offsets,
                          edges, capacity, (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_count_f64(
    const double *points, int64_t n_points, const double *centers,
    int64_t n_centers, double radius, const double *scale3_host,
    void *workspace, size_t workspace_bytes, int32_t *offsets, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_count_impl(points, n_points, centers, n_centers, radius,
                           scale3_host, workspace, workspace_bytes, offsets,
                           (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_fill_f64(
    const double *points, int64_t n_points, const double *centers,
    int64_t n_centers, double radius, const double *scale3_host,
    void *workspace, size_t workspace_bytes, const int32_t *offsets,
    int32_t *edges, int64_t capacity, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_fill_impl(points, n_points, centers, n_centers, radius,
                          scale3_host, workspace, workspace_bytes, offsets,
                          edges, capacity, (hipStream_t)stream_);
  PGNN_GUARD_END
}

// ---- capacity form: no count leaves the device --------------------------------
// graph_gen.py:155-220 never waits for a size (NumPy arrays carry their own);
// here the keypoint count K and the edge counts live in device memory and the
// outputs are sized by a capacity.  One call = grid + count + scan + fill.
namespace {
__global__ void edge_total_kernel(const int32_t *__restrict__ total,
                                  int64_t capacity, int32_t *__restrict__ n_out) {
  graph_prio();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int32_t e = *total;
    n_out[0] = (int64_t)e < capacity ? e : (int32_t)capacity;  // rows written
    n_out[1] = e;                                              // rows required
  }
}

// The capacity-form builder in its two stages: the cell grid needs the POINTS
// only, the queries additionally the centres -- a caller whose centres are
// still being computed (the keypoints) can issue the grid stage beside that.
template <typename T>
int radius_dyn_carve(const T *points, int64_t points_cap, int64_t centers_cap,
                     double radius, void *workspace, size_t workspace_bytes,
                     RadiusWs &w, int32_t **offsets) {
  PGNN_REQUIRE(points_cap >= 0 && centers_cap >= 0 && radius > 0.0,
               PGNN_E_INVALID, "radius_graph_dyn: bad argument");
  PGNN_REQUIRE(points_cap == 0 || points, PGNN_E_INVALID,
               "radius_graph_dyn: null pointer");
  PGNN_REQUIRE(workspace &&
                   workspace_bytes >= pgnn_radius_graph_dyn_workspace_bytes(
                                          points_cap, centers_cap),
               PGNN_E_WORKSPACE, "radius_graph_dyn: workspace too small");
  int rc = radius_carve(workspace, workspace_bytes, points_cap, centers_cap, w);
  if (rc) return rc;
  *offsets = reinterpret_cast<int32_t *>(
      (char *)workspace +
      pgnn_radius_graph_workspace_bytes(points_cap, centers_cap));
  return 0;
}

template <typename T>
int radius_dyn_grid(const T *points, int64_t points_cap,
                    const int32_t *n_points_dev, int64_t centers_cap,
                    double radius, const double *scale3_host, void *workspace,
                    size_t workspace_bytes, hipStream_t stream) {
  RadiusWs w;
  int32_t *offsets;
  int rc = radius_dyn_carve(points, points_cap, centers_cap, radius, workspace,
                            workspace_bytes, w, &offsets);
  if (rc) return rc;
  return grid_build(points, points_cap, make_scale(scale3_host), 0.0, 0.0, 0.0,
                    nullptr, radius, w.g, stream, nullptr, nullptr,
                    n_points_dev);
}

template <typename T>
int radius_dyn_query(const T *points, int64_t points_cap, const T *centers,
                     int64_t centers_cap, const int32_t *n_centers_dev,
                     double radius, const double *scale3_host, void *workspace,
                     size_t workspace_bytes, int32_t *edges,
                     int64_t edge_capacity, int32_t *n_edges_dev,
                     hipStream_t stream) {
  PGNN_REQUIRE(edge_capacity >= 0 && edge_capacity <= 0x7fffffff && n_edges_dev,
               PGNN_E_INVALID, "radius_graph_dyn: bad argument");
  PGNN_REQUIRE((centers_cap == 0 || centers) && (edge_capacity == 0 || edges),
               PGNN_E_INVALID, "radius_graph_dyn: null pointer");
  RadiusWs w;
  int32_t *offsets;
  int rc = radius_dyn_carve(points, points_cap, centers_cap, radius, workspace,
                            workspace_bytes, w, &offsets);
  if (rc) return rc;
  // (the tables the grid stage filled sit where the carve puts them: their
  // layout is a function of the workspace and the two capacities alone)
  const Scale3 sc = make_scale(scale3_host);
  if (centers_cap > 0) {
    hipLaunchKernelGGL((radius_query_kernel<false, T>),
                       dim3(graph_grid((centers_cap + (graph_wide_block() >> 6) - 1) /
                                       (graph_wide_block() >> 6))),
                       dim3(graph_wide_block()), graph_lds_pad(),
                       stream, centers, centers_cap, sc, radius, w.g.mask,
                       w.g.cell_start, w.g.cell_end, w.g.sorted, w.counts,
                       (const int32_t *)nullptr, (int32_t *)nullptr,
                       (int64_t)0, n_centers_dev);
    PGNN_HIP(hipGetLastError());
  }
  rc = exclusive_scan_i32(w.counts, offsets, centers_cap, w.scan_scratch,
                          w.scan_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(edge_total_kernel, dim3(1), dim3(64), graph_lds_pad(), stream,
                     (const int32_t *)(offsets + centers_cap), edge_capacity,
                     n_edges_dev);
  if (centers_cap > 0 && edge_capacity > 0) {
    hipLaunchKernelGGL((radius_query_kernel<true, T>),
                       dim3(graph_grid((centers_cap + (graph_wide_block() >> 6) - 1) /
                                       (graph_wide_block() >> 6))),
                       dim3(graph_wide_block()), graph_lds_pad(),
                       stream, centers, centers_cap, sc, radius, w.g.mask,
                       w.g.cell_start, w.g.cell_end, w.g.sorted,
                       (int32_t *)nullptr, (const int32_t *)offsets, edges,
                       edge_capacity, n_centers_dev);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" size_t pgnn_radius_graph_dyn_workspace_bytes(int64_t points_cap,
                                                        int64_t centers_cap) {
  if (points_cap < 0 || centers_cap < 0) return 0;
  return pgnn_radius_graph_workspace_bytes(points_cap, centers_cap) +
         align_up((size_t)(centers_cap + 1) * 4, 256);
}

extern "C" int pgnn_radius_graph_dyn_grid(
    const float *points, int64_t points_cap, const int32_t *n_points_dev,
    int64_t centers_cap, double radius, const double *scale3_host,
    void *workspace, size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_grid(points, points_cap, n_points_dev, centers_cap, radius,
                         scale3_host, workspace, workspace_bytes,
                         (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_dyn_grid_f64(
    const double *points, int64_t points_cap, const int32_t *n_points_dev,
    int64_t centers_cap, double radius, const double *scale3_host,
    void *workspace, size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_grid(points, points_cap, n_points_dev, centers_cap, radius,
                         scale3_host, workspace, workspace_bytes,
                         (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_dyn_query(
    const float *points, int64_t points_cap, const float *centers,
    int64_t centers_cap, const int32_t *n_centers_dev, double radius,
    const double *scale3_host, void *workspace, size_t workspace_bytes,
    int32_t *edges, int64_t edge_capacity, int32_t *n_edges_dev,
    void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_query(points, points_cap, centers, centers_cap,
                          n_centers_dev, radius, scale3_host, workspace,
                          workspace_bytes, edges, edge_capacity, n_edges_dev,
                          (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_dyn_query_f64(
    const double *points, int64_t points_cap, const double *centers,
    int64_t centers_cap, const int32_t *n_centers_dev, double radius,
    const double *scale3_host, void *workspace, size_t workspace_bytes,
    int32_t *edges, int64_t edge_capacity, int32_t *n_edges_dev,
    void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_query(points, points_cap, centers, centers_cap,
                          n_centers_dev, radius, scale3_host, workspace,
                          workspace_bytes, edges, edge_capacity, n_edges_dev,
                          (hipStream_t)stream_);
  PGNN_GUARD_END
}

namespace {
// Both stages in order on one stream.  The arguments are checked for BOTH
// stages before anything is enqueued.
template <typename T>
int radius_dyn_both(const T *points, int64_t points_cap,
                    const int32_t *n_points_dev, const T *centers,
                    int64_t centers_cap, const int32_t *n_centers_dev,
                    double radius, const double *scale3_host, void *workspace,
                    size_t workspace_bytes, int32_t *edges,
                    int64_t edge_capacity, int32_t *n_edges_dev,
                    hipStream_t stream) {
  PGNN_REQUIRE(points_cap >= 0 && centers_cap >= 0 && radius > 0.0 &&
                   edge_capacity >= 0 && edge_capacity <= 0x7fffffff &&
                   n_edges_dev,
               PGNN_E_INVALID, "radius_graph_dyn: bad argument");
  PGNN_REQUIRE((points_cap == 0 || points) && (centers_cap == 0 || centers) &&
                   (edge_capacity == 0 || edges),
               PGNN_E_INVALID, "radius_graph_dyn: null pointer");
  int rc = radius_dyn_grid(points, points_cap, n_points_dev, centers_cap,
                           radius, scale3_host, workspace, workspace_bytes,
                           stream);
  if (rc) return rc;
  return radius_dyn_query(points, points_cap, centers, centers_cap,
                          n_centers_dev, radius, scale3_host, workspace,
                          workspace_bytes, edges, edge_capacity, n_edges_dev,
                          stream);
}
}  // namespace

extern "C" int pgnn_radius_graph_dyn(
    const float *points, int64_t points_cap, const int32_t *n_points_dev,
    const float *centers, int64_t centers_cap, const int32_t *n_centers_dev,
    double radius, const double *scale3_host, void *workspace,
    size_t workspace_bytes, int32_t *edges, int64_t edge_capacity,
    int32_t *n_edges_dev, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_both(points, points_cap, n_points_dev, centers, centers_cap,
                         n_centers_dev, radius, scale3_host, workspace,
                         workspace_bytes, edges, edge_capacity, n_edges_dev,
                         (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_dyn_f64(
    const double *points, int64_t points_cap, const int32_t *n_points_dev,
    const double *centers, int64_t centers_cap, const int32_t *n_centers_dev,
    double radius, const double *scale3_host, void *workspace,
    size_t workspace_bytes, int32_t *edges, int64_t edge_capacity,
    int32_t *n_edges_dev, void *stream_) {
  PGNN_GUARD_BEGIN
  return radius_dyn_both(points, points_cap, n_points_dev, centers, centers_cap,
                         n_centers_dev, radius, scale3_host, workspace,
                         workspace_bytes, edges, edge_capacity, n_edges_dev,
                         (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_cap_neighbors_count(const int32_t *offsets,
                                        int64_t n_centers,
                                        int32_t max_neighbors,
                                        int32_t *new_offsets, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(offsets && new_offsets && n_centers >= 0 && max_neighbors > 0,
               PGNN_E_INVALID, "cap_neighbors_count: bad argument");
  hipLaunchKernelGGL(cap_counts_kernel,
                     dim3((unsigned)((n_centers + 256) / 256)), dim3(256), graph_lds_pad(),
                     stream, offsets, n_centers, max_neighbors, new_offsets);
  PGNN_HIP(hipGetLastError());
  // exclusive scan over n_centers + 1 entries leaves the total in the last one
  return exclusive_scan_inplace_i32(new_offsets, n_centers + 1, stream);
  PGNN_GUARD_END
}

extern "C" int pgnn_cap_neighbors_fill(const int32_t *offsets,
                                       const int32_t *edges, int64_t n_centers,
                                       int32_t max_neighbors, uint64_t seed,
                                       const int32_t *new_offsets,
                                       int32_t *new_edges, int64_t new_capacity,
                                       void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(offsets && new_offsets && n_centers >= 0 && max_neighbors > 0 &&
                   new_capacity >= 0,
               PGNN_E_INVALID, "cap_neighbors_fill: bad argument");
  if (n_centers == 0 || new_capacity == 0) return 0;
  PGNN_REQUIRE(edges && new_edges, PGNN_E_INVALID,
               "cap_neighbors_fill: null edges");
  hipLaunchKernelGGL(cap_fill_kernel, dim3((unsigned)((n_centers + 3) / 4)),
                     dim3(256), graph_lds_pad(), stream, offsets, edges, n_centers,
                     max_neighbors, seed, new_offsets, new_edges, new_capacity,
                     (int64_t)0x7fffffff);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_radius_graph_dyn_cap(
    const void *workspace, size_t workspace_bytes, int64_t points_cap,
    int64_t centers_cap, const int32_t *edges, int64_t edge_capacity,
    const int32_t *n_edges_dev, int32_t max_neighbors, uint64_t seed,
    int32_t *new_offsets, int32_t *new_edges, int64_t new_capacity,
    int32_t *n_new_dev, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(points_cap >= 0 && centers_cap >= 0 && edge_capacity >= 0 &&
                   new_capacity >= 0 && new_capacity <= 0x7fffffff &&
                   max_neighbors > 0,
               PGNN_E_INVALID, "radius_graph_dyn_cap: bad argument");
  PGNN_REQUIRE(workspace && n_edges_dev && new_offsets && n_new_dev &&
                   (edge_capacity == 0 || edges) &&
                   (new_capacity == 0 || new_edges),
               PGNN_E_INVALID, "radius_graph_dyn_cap: null pointer");
  PGNN_REQUIRE(workspace_bytes >= pgnn_radius_graph_dyn_workspace_bytes(
                                      points_cap, centers_cap),
               PGNN_E_WORKSPACE, "radius_graph_dyn_cap: workspace too small");
  // the CSR offsets pgnn_radius_graph_dyn(_query) left behind its tables
  const int32_t *offsets = reinterpret_cast<const int32_t *>(
      (const char *)workspace +
      pgnn_radius_graph_workspace_bytes(points_cap, centers_cap));
  hipLaunchKernelGGL(cap_counts_kernel,
                     dim3((unsigned)((centers_cap + 256) / 256)), dim3(256),
                     graph_lds_pad(), stream, offsets, centers_cap,
                     max_neighbors, new_offsets);
  PGNN_HIP(hipGetLastError());
  int rc = exclusive_scan_inplace_i32(new_offsets, centers_cap + 1, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(cap_total_kernel, dim3(1), dim3(64), graph_lds_pad(), stream,
                     (const int32_t *)(new_offsets + centers_cap), n_edges_dev,
                     new_capacity, n_new_dev);
  if (centers_cap > 0 && new_capacity > 0 && edge_capacity > 0) {
    hipLaunchKernelGGL(cap_fill_kernel, dim3((unsigned)((centers_cap + 3) / 4)),
                       dim3(256), graph_lds_pad(), stream, offsets, edges,
                       centers_cap, max_neighbors, seed,
                       (const int32_t *)new_offsets, new_edges, new_capacity,
                       edge_capacity);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

// workspace layout (keypoints): [Grid | ordered_min[4] | voxel rule[10] |
//   is_leader[n] | slot[n+1] | member_count[n] | centroid[3n] | vcell[3n] x2 |
//   scan scratch]
extern "C" size_t pgnn_keypoints_workspace_bytes(int64_t n_points) {
  if (n_points < 0) return 0;
  const size_t n = (size_t)(n_points > 0 ? n_points : 1);
  return grid_bytes(n_points) + 256 + 256 + 3 * align_up((n + 1) * 4, 256) +
         align_up(3 * n * 8, 256) + 2 * align_up(3 * n * 4, 256) +
         align_up(scan_scratch_bytes(n_points), 256) +
         kd_workspace_bytes(n_points) + 2048;
}

namespace pgnn {
int g_graph_debug = 0;  // benchmarks only: 1 = skip the kd-tree replica (exact
                        // ties then go to idx_array-free slot order)
}
namespace {
// The (fork, join) event pair of a (stream, aux stream) couple, created once
// and re-recorded every call.  Creating and destroying two events per frame
// is not free in the HIP runtime: destroying an event whose record is still
// pending makes the host wait for it, i.e. for the whole kd-tree build, on
// every frame.  (An internal cache like device_cu_count()'s; entries live as
// long as the process.)
int fork_join_events(hipStream_t stream, hipStream_t aux, hipEvent_t *fork,
                     hipEvent_t *join) {
  struct Pair {
    hipStream_t s, a;
    hipEvent_t f, j;
  };
  static std::mutex mu;
  static std::vector<Pair> cache;
  std::lock_guard<std::mutex> lock(mu);
  for (const Pair &p : cache)
    if (p.s == stream && p.a == aux) {
      *fork = p.f;
      *join = p.j;
      return 0;
    }
  Pair p = {stream, aux, nullptr, nullptr};
  PGNN_HIP(hipEventCreateWithFlags(&p.f, hipEventDisableTiming));
  PGNN_HIP(hipEventCreateWithFlags(&p.j, hipEventDisableTiming));
  cache.push_back(p);
  *fork = p.f;
  *join = p.j;
  return 0;
}

inline int kd_build_any(const float *pts, int64_t n, Arena &a, KdBuild &kb,
                        hipStream_t s) {
  return kd_build(pts, n, a, kb, s);
}
inline int kd_build_any(const double *, int64_t, Arena &, KdBuild &,
                        hipStream_t) {
  return fail(PGNN_E_UNSUPPORTED, "keypoints: 'center' mode on float64 points");
}
inline void launch_voxel_nn(const Grid &g, int64_t n, const double *origin,
                            double voxel, const int32_t *is_leader,
                            const int32_t *slot, const double *centroid,
                            const float *points, const KdView &kd,
                            int32_t *kp_idx, float *kp_xyz, hipStream_t stream);
inline void launch_voxel_nn(const Grid &, int64_t, const double *, double,
                            const int32_t *, const int32_t *, const double *,
                            const double *, const KdView &, int32_t *, double *,
                            hipStream_t) {}

template <typename T>
int keypoints_impl(const T *points, int64_t n, double voxel, bool center,
                   const double *jitter3, uint64_t seed, void *workspace,
                   size_t workspace_bytes, int32_t *kp_idx, T *kp_xyz,
                   int32_t *num_kp, hipStream_t stream,
                   hipStream_t aux = nullptr,
                   const T *origin_pts = nullptr /* the cloud whose minimum
                       anchors the voxel grid (graph_gen.py:108-110: the
                       ORIGINAL cloud's, also for the second and later pooling
                       levels); null: `points` */,
                   int64_t n_origin = 0) {
  constexpr bool kF64 = sizeof(T) == 8;
  PGNN_REQUIRE(n >= 0 && voxel > 0.0 && kp_idx && kp_xyz && num_kp,
               PGNN_E_INVALID, "keypoints: bad argument");
  // 'center' on a float64 cloud: the kd-tree replica keys are float32 and no
  // shipped config asks for it (training uses 'random', run.py feeds float32)
  PGNN_REQUIRE(!(center && kF64), PGNN_E_UNSUPPORTED,
               "keypoints: 'center' mode takes float32 points");
  if (n == 0) {
    PGNN_HIP(hipMemsetAsync(num_kp, 0, 8, stream));
    return 0;
  }
  PGNN_REQUIRE(points != nullptr, PGNN_E_INVALID, "keypoints: null points");
  PGNN_REQUIRE(workspace && workspace_bytes >= pgnn_keypoints_workspace_bytes(n),
               PGNN_E_WORKSPACE, "keypoints: workspace too small");
  Arena a(workspace, workspace_bytes);
  Grid g;
  int rc = grid_carve(a, n, g);
  if (rc) return rc;
  unsigned long long *omin = a.take<unsigned long long>(4);
  double *origin = a.take<double>(kVoxRuleDoubles);
  int32_t *is_leader = a.take<int32_t>((size_t)n + 1);
  int32_t *slot = a.take<int32_t>((size_t)n + 1);
  int32_t *members = a.take<int32_t>((size_t)n + 1);
  double *centroid = a.take<double>(3 * (size_t)n);
  int32_t *vcell = a.take<int32_t>(3 * (size_t)n);
  int32_t *vcell_sorted = a.take<int32_t>(3 * (size_t)n);
  const size_t scan_bytes = scan_scratch_bytes(n);
  void *scan_scratch = a.take<char>(scan_bytes);
  PGNN_REQUIRE(omin && origin && is_leader && slot && members && centroid &&
                   vcell && vcell_sorted && scan_scratch,
               PGNN_E_WORKSPACE, "keypoints: workspace too small");
  // scikit-learn's kd-tree node order over the points decides exact ties
  // ('center' only).  It depends on nothing but the points: with an aux
  // stream it runs beside the voxel hashing (fork here, join before the
  // nearest-neighbour kernel).
  KdBuild kb;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // The join is enqueued in front of the first kernel that reads the tree
  // (copy_total_kernel: its status word), i.e. BEHIND the voxel hashing --
  // and on every other way out of this function: nothing may be left running
  // on aux that `stream`'s later work does not wait for.
  struct Joiner {
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    void arm(hipStream_t s_, hipEvent_t e_) { s = s_; e = e_; }
    void join() {
      if (e) (void)hipStreamWaitEvent(s, e, 0);
      e = nullptr;
    }
    ~Joiner() { join(); }
  } joiner;
  kb.pos = nullptr;
  kb.bounds = nullptr;
  kb.status = nullptr;
  kb.n_nodes = 0;
#ifdef PGNN_DIAG  // ablation (non-reference tie order): diagnostic builds only
  const bool use_kd = center && !(g_graph_debug & 1);
#else
  const bool use_kd = center;
#endif
  if (!use_kd && center) {  // ablation: a valid (all-zero) slot table
    kb.pos = a.take<int32_t>((size_t)n);
    PGNN_REQUIRE(kb.pos, PGNN_E_WORKSPACE, "keypoints: workspace too small");
    PGNN_HIP(hipMemsetAsync(kb.pos, 0, (size_t)n * 4, stream));
  }
  if (use_kd) {
    hipStream_t kd_stream = stream;
    if (aux && aux != stream) {
      int erc = fork_join_events(stream, aux, &ev_fork, &ev_join);
      if (erc) return erc;
      PGNN_HIP(hipEventRecord(ev_fork, stream));
      PGNN_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
      kd_stream = aux;
    }
    rc = kd_build_any(points, n, a, kb, kd_stream);
    if (ev_join) (void)hipEventRecord(ev_join, aux);
    joiner.arm(stream, ev_join);
    if (rc) return rc;
  }
  PGNN_HIP((hipError_t)graph_fill32(omin, 0xffffffffu, 8, stream));
  const T *min_pts = origin_pts && n_origin > 0 ? origin_pts : points;
  const int64_t n_min = origin_pts && n_origin > 0 ? n_origin : n;
  int mb = (int)((n_min + 255) / 256);
  if (mb > 1024) mb = 1024;
  hipLaunchKernelGGL(min_bound_kernel<T>, dim3(graph_grid(mb)), dim3(256), graph_lds_pad(), stream, min_pts,
                     n_min, omin);
  double sx, sy, sz;
  if (center) {
    sx = sy = sz = voxel * 0.5;  // open3d 0.7: origin = min_bound - voxel/2
  } else {
    sx = jitter3 ? jitter3[0] : 0.0;  // graph_gen.py:126-128: + jitter
    sy = jitter3 ? jitter3[1] : 0.0;
    sz = jitter3 ? jitter3[2] : 0.0;
  }
  const int mode = center ? 0 : (kF64 ? (jitter3 ? 4 : 3) : (jitter3 ? 2 : 1));
  hipLaunchKernelGGL(grid_origin_kernel, dim3(1), dim3(64), graph_lds_pad(), stream, omin, sx,
                     sy, sz, mode, origin);
  Scale3 sc = make_scale(nullptr);
  rc = grid_build(points, n, sc, 0.0, 0.0, 0.0, origin, voxel, g, stream, vcell,
                  vcell_sorted);
  if (rc) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(voxel_leader_kernel, dim3(graph_grid(blocks)), dim3(256), graph_lds_pad(), stream,
                     g.sorted, g.keys, n, (const int32_t *)vcell_sorted,
                     g.cell_start, g.cell_end, is_leader, centroid, members);
  rc = exclusive_scan_i32(is_leader, slot, n, scan_scratch, scan_bytes, stream);
  if (rc) return rc;
  joiner.join();
  hipLaunchKernelGGL(copy_total_kernel, dim3(1), dim3(64), graph_lds_pad(), stream, slot + n,
                     (const int32_t *)kb.status, num_kp);
  if (center) {
    KdView kd;
    kd.pos = kb.pos;
    kd.bounds = kb.bounds;
    kd.n = (int32_t)n;
    kd.n_nodes = kb.n_nodes;
    launch_voxel_nn(g, n, origin, voxel, is_leader, slot, centroid, points, kd,
                    kp_idx, kp_xyz, stream);
  } else {
    hipLaunchKernelGGL(voxel_random_pick_kernel<T>, dim3(graph_grid(blocks)), dim3(256), graph_lds_pad(),
                       stream, g.sorted, g.keys, n,
                       (const int32_t *)vcell_sorted, g.cell_end, is_leader,
                       slot, members, seed, points, kp_idx, kp_xyz);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
}

inline void launch_voxel_nn(const Grid &g, int64_t n, const double *origin,
                            double voxel, const int32_t *is_leader,
                            const int32_t *slot, const double *centroid,
                            const float *points, const KdView &kd,
                            int32_t *kp_idx, float *kp_xyz,
                            hipStream_t stream) {
  const int64_t wb = graph_wide_block();
  const int spw = g_graph_max_wgs > 0 ? 64 : 4;
  const int64_t per_wg = (wb >> 6) * spw;
  hipLaunchKernelGGL(voxel_nn_kernel, dim3(graph_grid((n + per_wg - 1) / per_wg)),
                     dim3((unsigned)wb), graph_lds_pad(), stream, g.sorted, n, origin, voxel, g.mask,
                     g.cell_start, g.cell_end, is_leader, slot, centroid, points,
                     kd, kp_idx, kp_xyz, spw);
}
}  // namespace

// workspace layout (keypoints, second and later pooling levels, 'center'):
//   [Grid(points) | ordered_min | voxel rule | is_leader | slot | members |
//    centroid | vcell x2 | scan scratch | Grid(base) | vcell(base) x2 | kd(base)]
extern "C" size_t pgnn_keypoints_from_workspace_bytes(int64_t n_points,
                                                      int64_t n_base) {
  if (n_points < 0 || n_base < 0) return 0;
  const size_t nb = (size_t)(n_base > 0 ? n_base : 1);
  return pgnn_keypoints_workspace_bytes(n_points) + grid_bytes(n_base) +
         2 * align_up(3 * nb * 4, 256) + kd_workspace_bytes(n_base) + 2048;
}

namespace {
// graph_gen.py:41-45 + :78-88 for a pooling level whose search set is not the
// voxelised cloud: centroids of `points`' voxels, 1-NN among `base`.
int keypoints_center_from_impl(const float *points, int64_t n, const float *base,
                               int64_t nb, double voxel, void *workspace,
                               size_t workspace_bytes, int32_t *kp_idx,
                               float *kp_xyz, int32_t *num_kp,
                               hipStream_t stream) {
  PGNN_REQUIRE(n >= 0 && nb >= 0 && voxel > 0.0 && kp_idx && kp_xyz && num_kp,
               PGNN_E_INVALID, "keypoints_from: bad argument");
  if (n == 0) {
    PGNN_HIP(hipMemsetAsync(num_kp, 0, 8, stream));
    return 0;
  }
  PGNN_REQUIRE(points && base && nb > 0, PGNN_E_INVALID,
               "keypoints_from: null points / empty search set");
  PGNN_REQUIRE(workspace &&
                   workspace_bytes >= pgnn_keypoints_from_workspace_bytes(n, nb),
               PGNN_E_WORKSPACE, "keypoints_from: workspace too small");
  Arena a(workspace, workspace_bytes);
  Grid g, gb;
  int rc = grid_carve(a, n, g);
  if (rc) return rc;
  unsigned long long *omin = a.take<unsigned long long>(4);
  double *origin = a.take<double>(kVoxRuleDoubles);
  int32_t *is_leader = a.take<int32_t>((size_t)n + 1);
  int32_t *slot = a.take<int32_t>((size_t)n + 1);
  int32_t *members = a.take<int32_t>((size_t)n + 1);
  double *centroid = a.take<double>(3 * (size_t)n);
  int32_t *vcell = a.take<int32_t>(3 * (size_t)n);
  int32_t *vcell_sorted = a.take<int32_t>(3 * (size_t)n);
  const size_t scan_bytes = scan_scratch_bytes(n);
  void *scan_scratch = a.take<char>(scan_bytes);
  rc = grid_carve(a, nb, gb);
  if (rc) return rc;
  int32_t *vcell_b = a.take<int32_t>(3 * (size_t)nb);
  int32_t *vcell_b_sorted = a.take<int32_t>(3 * (size_t)nb);
  PGNN_REQUIRE(omin && origin && is_leader && slot && members && centroid &&
                   vcell && vcell_sorted && scan_scratch && vcell_b &&
                   vcell_b_sorted,
               PGNN_E_WORKSPACE, "keypoints_from: workspace too small");
  // scikit-learn's kd-tree over the SEARCH set decides exact ties
  KdBuild kb;
  rc = kd_build(base, nb, a, kb, stream);
  if (rc) return rc;
  PGNN_HIP((hipError_t)graph_fill32(omin, 0xffffffffu, 8, stream));
  int mb = (int)((n + 255) / 256);
  if (mb > 1024) mb = 1024;
  hipLaunchKernelGGL(min_bound_kernel<float>, dim3(graph_grid(mb)), dim3(256),
                     graph_lds_pad(), stream, points, n, omin);
  // open3d 0.7: origin = min_bound(points) - voxel / 2
  hipLaunchKernelGGL(grid_origin_kernel, dim3(1), dim3(64), graph_lds_pad(), stream,
                     omin, voxel * 0.5, voxel * 0.5, voxel * 0.5, 0, origin);
  Scale3 sc = make_scale(nullptr);
  rc = grid_build(points, n, sc, 0.0, 0.0, 0.0, origin, voxel, g, stream, vcell,
                  vcell_sorted);
  if (rc) return rc;
  rc = grid_build(base, nb, sc, 0.0, 0.0, 0.0, origin, voxel, gb, stream, vcell_b,
                  vcell_b_sorted);
  if (rc) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(voxel_leader_kernel, dim3(graph_grid(blocks)), dim3(256),
                     graph_lds_pad(), stream, g.sorted, g.keys, n,
                     (const int32_t *)vcell_sorted, g.cell_start, g.cell_end,
                     is_leader, centroid, members);
  rc = exclusive_scan_i32(is_leader, slot, n, scan_scratch, scan_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(copy_total_kernel, dim3(1), dim3(64), graph_lds_pad(), stream,
                     slot + n, (const int32_t *)kb.status, num_kp);
  KdView kd;
  kd.pos = kb.pos;
  kd.bounds = kb.bounds;
  kd.n = (int32_t)nb;
  kd.n_nodes = kb.n_nodes;
  const int64_t per_wg = 4 * 4;  // 4 waves x 4 sorted slots
  hipLaunchKernelGGL(voxel_nn_from_kernel,
                     dim3(graph_grid((n + per_wg - 1) / per_wg)), dim3(256),
                     graph_lds_pad(), stream, gb.sorted, nb, origin, voxel, gb.mask,
                     gb.cell_start, gb.cell_end, n, is_leader, slot, centroid, base,
                     kd, kp_idx, kp_xyz);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_voxel_keypoints_center_from(
    const float *points, int64_t n_points, const float *search_points,
    int64_t n_search, double voxel_size, void *workspace, size_t workspace_bytes,
    int32_t *keypoint_indices, float *keypoint_xyz, int32_t *num_keypoints,
    void *stream) {
  PGNN_GUARD_BEGIN
  return keypoints_center_from_impl(points, n_points, search_points, n_search,
                                    voxel_size, workspace, workspace_bytes,
                                    keypoint_indices, keypoint_xyz, num_keypoints,
                                    (hipStream_t)stream);
  PGNN_GUARD_END
}

extern "C" int pgnn_voxel_keypoints_random_from(
    const float *points, int64_t n_points, const float *origin_points,
    int64_t n_origin, double voxel_size, const double *jitter3_host,
    uint64_t seed, void *workspace, size_t workspace_bytes,
    int32_t *keypoint_indices, float *keypoint_xyz, int32_t *num_keypoints,
    void *stream) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_origin >= 0 && (n_origin == 0 || origin_points), PGNN_E_INVALID,
               "keypoints_random_from: bad origin set");
  return keypoints_impl(points, n_points, voxel_size, false, jitter3_host, seed,
                        workspace, workspace_bytes, keypoint_indices,
                        keypoint_xyz, num_keypoints, (hipStream_t)stream, nullptr,
                        origin_points, n_origin);
  PGNN_GUARD_END
}

extern "C" int pgnn_voxel_keypoints_random_from_f64(
    const double *points, int64_t n_points, const double *origin_points,
    int64_t n_origin, double voxel_size, const double *jitter3_host,
    uint64_t seed, void *workspace, size_t workspace_bytes,
    int32_t *keypoint_indices, double *keypoint_xyz, int32_t *num_keypoints,
    void *stream) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_origin >= 0 && (n_origin == 0 || origin_points), PGNN_E_INVALID,
               "keypoints_random_from: bad origin set");
  return keypoints_impl(points, n_points, voxel_size, false, jitter3_host, seed,
                        workspace, workspace_bytes, keypoint_indices,
                        keypoint_xyz, num_keypoints, (hipStream_t)stream, nullptr,
                        origin_points, n_origin);
  PGNN_GUARD_END
}

extern "C" int pgnn_voxel_keypoints_center(const float *points, int64_t n_points,
                                           double voxel_size, void *workspace,
                                           size_t workspace_bytes,
                                           int32_t *keypoint_indices,
                                           float *keypoint_xyz,
                                           int32_t *num_keypoints,
                                           void *stream, void *aux_stream) {
  PGNN_GUARD_BEGIN
  return keypoints_impl(points, n_points, voxel_size, true, nullptr, 0,
                        workspace, workspace_bytes, keypoint_indices,
                        keypoint_xyz, num_keypoints, (hipStream_t)stream,
                        (hipStream_t)aux_stream);
  PGNN_GUARD_END
}

extern "C" int pgnn_voxel_keypoints_random(
    const float *points, int64_t n_points, double voxel_size,
    const double *jitter3_host, uint64_t seed, void *workspace,
    size_t workspace_bytes, int32_t *keypoint_indices, float *keypoint_xyz,
    int32_t *num_keypoints, void *stream) {
  PGNN_GUARD_BEGIN
  return keypoints_impl(points, n_points, voxel_size, false, jitter3_host, seed,
                        workspace, workspace_bytes, keypoint_indices,
                        keypoint_xyz, num_keypoints, (hipStream_t)stream);
  PGNN_GUARD_END
}

extern "C" int pgnn_voxel_keypoints_random_f64(
    const double *points, int64_t n_points, double voxel_size,
    const double *jitter3_host, uint64_t seed, void *workspace,
    size_t workspace_bytes, int32_t *keypoint_indices, double *keypoint_xyz,
    int32_t *num_keypoints, void *stream) {
  PGNN_GUARD_BEGIN
  return keypoints_impl(points, n_points, voxel_size, false, jitter3_host, seed,
                        workspace, workspace_bytes, keypoint_indices,
                        keypoint_xyz, num_keypoints, (hipStream_t)stream);
  PGNN_GUARD_END
}
