// Stable LSD radix sort (8-bit digits) and exclusive scan for the graph
// builders.  Sizes here are tens of thousands of points: these kernels are
// launch-latency-bound, so the design goal is determinism (a *stable* sort
// makes every downstream result independent of atomics ordering) and a small
// number of launches, not peak bandwidth.
#include "sort.h"

namespace pgnn {
namespace {

constexpr int kRadix = 256;
constexpr int kSortThreads = 256;
constexpr int kSortItems = kSortTile / kSortThreads;  // 8 rounds per tile

// ---- block-wide exclusive scan of one int per thread (1024 threads) --------
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// returns the exclusive prefix of v within the block; *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *total,
                                                    int *lds /* >= 17 ints */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  int inc = wave_inclusive_scan(v, lane);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  if (w == 0) {
    int s = (lane < nw) ? lds[lane] : 0;
    int si = wave_inclusive_scan(s, lane);
    if (lane < nw) lds[lane] = si - s;  // exclusive wave offsets
    if (lane == nw - 1) lds[16] = si;
  }
  __syncthreads();
  int res = inc - v + lds[w];
  *total = lds[16];
  __syncthreads();
  return res;
}

// single block, any n: in-place exclusive scan of data[0..n), returns nothing;
// optionally writes the grand total to *total_out.
__global__ __launch_bounds__(1024) void scan_single_block_kernel(
    int32_t *data, int64_t n, int32_t *total_out) {
  graph_prio();
  __shared__ int lds[17];
  int carry = 0;
  for (int64_t base = 0; base < n; base += 1024) {
    int64_t i = base + threadIdx.x;
    int v = (i < n) ? data[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot, lds);
    if (i < n) data[i] = ex + carry;
    carry += tot;
  }
  if (total_out && threadIdx.x == 0) *total_out = carry;
}

// ---- 3-kernel scan for larger n --------------------------------------------
constexpr int kScanTile = 4096;  // 1024 threads x 4

__global__ __launch_bounds__(1024) void scan_tile_sums_kernel(
    const int32_t *in, int64_t n, int32_t *tile_sums, int num_tiles) {
  graph_prio();
  __shared__ int lds[17];
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    int64_t base = (int64_t)tile * kScanTile + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + k < n) s += in[base + k];
    int tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}

__global__ __launch_bounds__(1024) void scan_tiles_kernel(
    const int32_t *in, int32_t *out, int64_t n, const int32_t *tile_offsets,
    int num_tiles) {
  graph_prio();
  __shared__ int lds[17];
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    int64_t base = (int64_t)tile * kScanTile + threadIdx.x * 4;
    int v[4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (base + k < n) ? in[base + k] : 0;
      s += v[k];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, lds) + tile_offsets[tile];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + k < n) out[base + k] = ex;
      ex += v[k];
    }
    // the entry one past the end holds the grand total
    if (base <= n - 1 && n - 1 < base + 4) out[n] = ex;
  }
}

// ---- radix sort --------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(
    const uint32_t *__restrict__ keys, int64_t n, int shift,
    int32_t *__restrict__ block_hist /* [kRadix][num_tiles] */, int num_tiles,
    const int32_t *__restrict__ n_dev /* nullable: the pair count lives on the
                                         device, n is its upper bound */) {
  graph_prio();
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  __shared__ int hist[kRadix];
  // a workgroup takes tiles blockIdx.x, + gridDim.x, ...: the frame pipeline
  // caps the builder's grids (graph_max_wgs)
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * kSortTile;
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      int64_t idx = base + i * kSortThreads + threadIdx.x;
      if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & (kRadix - 1)], 1);
    }
    __syncthreads();
    block_hist[(int64_t)threadIdx.x * num_tiles + tile] = hist[threadIdx.x];
  }
}

__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
    int shift, const int32_t *__restrict__ block_offs /* scanned */,
    int num_tiles, const int32_t *__restrict__ n_dev) {
  graph_prio();
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  __shared__ int base[kRadix];     // next output slot per digit for this tile
  __shared__ int wcnt[4][kRadix];  // per-wave digit counts of the current round
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    base[t] = block_offs[(int64_t)t * num_tiles + tile];
#pragma unroll
    for (int r = 0; r < 4; ++r) wcnt[r][t] = 0;
    __syncthreads();
    const int64_t tile_base = (int64_t)tile * kSortTile;
    for (int i = 0; i < kSortItems; ++i) {
      const int64_t idx = tile_base + i * kSortThreads + t;
      const bool valid = idx < n;
      const uint32_t key = valid ? keys_in[idx] : 0u;
      const uint32_t val = valid ? vals_in[idx] : 0u;
      const int digit = (int)((key >> shift) & (kRadix - 1));
      // lanes of this wave holding the same digit (valid lanes only)
      unsigned long long mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1;
        const unsigned long long bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
      }
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (valid && rank == 0) wcnt[w][digit] = __popcll(mask);
      __syncthreads();
      if (valid) {
        int off = base[digit] + rank;
        for (int w2 = 0; w2 < w; ++w2) off += wcnt[w2][digit];
        keys_out[off] = key;
        vals_out[off] = val;
      }
      __syncthreads();
      {
        int tot = wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
        base[t] += tot;
#pragma unroll
        for (int r = 0; r < 4; ++r) wcnt[r][t] = 0;
      }
      __syncthreads();
    }
  }
}

}  // namespace

size_t radix_sort_scratch_bytes(int64_t n) {
  int64_t tiles = (n + kSortTile - 1) / kSortTile;
  if (tiles < 1) tiles = 1;
  return align_up((size_t)tiles * kRadix * sizeof(int32_t), 256) + 256;
}

int radix_sort_pairs(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b,
                     uint32_t *vals_b, int64_t n, int nbits, void *scratch,
                     size_t scratch_bytes, uint32_t **keys_out,
                     uint32_t **vals_out, hipStream_t stream,
                     const int32_t *n_dev) {
  *keys_out = keys_a;
  *vals_out = vals_a;
  if (n <= 0) return 0;
  PGNN_REQUIRE(scratch_bytes >= radix_sort_scratch_bytes(n), PGNN_E_WORKSPACE,
               "radix sort: scratch too small");
  const int num_tiles = (int)((n + kSortTile - 1) / kSortTile);
  int32_t *hist = (int32_t *)scratch;
  uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
  for (int shift = 0; shift < nbits; shift += 8) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3(graph_grid(num_tiles)), dim3(kSortThreads), graph_lds_pad(),
                       stream, kin, n, shift, hist, num_tiles, n_dev);
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), graph_lds_pad(), stream,
                       hist, (int64_t)kRadix * num_tiles, (int32_t *)nullptr);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(graph_grid(num_tiles)),
                       dim3(kSortThreads), graph_lds_pad(), stream, kin, vin, kout, vout, n,
                       shift, hist, num_tiles, n_dev);
    uint32_t *tk = kin, *tv = vin;
    kin = kout;
    vin = vout;
    kout = tk;
    vout = tv;
  }
  PGNN_HIP(hipGetLastError());
  *keys_out = kin;
  *vals_out = vin;
  return 0;
}

size_t scan_scratch_bytes(int64_t n) {
  int64_t tiles = (n + kScanTile - 1) / kScanTile;
  if (tiles < 1) tiles = 1;
  return align_up((size_t)(tiles + 1) * sizeof(int32_t), 256);
}

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n,
                       void *scratch, size_t scratch_bytes,
                       hipStream_t stream) {
  if (n <= 0) {
    PGNN_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), stream));
    return 0;
  }
  PGNN_REQUIRE(scratch_bytes >= scan_scratch_bytes(n), PGNN_E_WORKSPACE,
               "scan: scratch too small");
  const int num_tiles = (int)((n + kScanTile - 1) / kScanTile);
  int32_t *tile_sums = (int32_t *)scratch;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(graph_grid(num_tiles)), dim3(1024), graph_lds_pad(),
                     stream, in, n, tile_sums, num_tiles);
  hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), graph_lds_pad(), stream,
                     tile_sums, (int64_t)num_tiles, (int32_t *)nullptr);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(graph_grid(num_tiles)), dim3(1024), graph_lds_pad(), stream,
                     in, out, n, tile_sums, num_tiles);
  PGNN_HIP(hipGetLastError());
  return 0;
}

int exclusive_scan_inplace_i32(int32_t *data, int64_t n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(1024), graph_lds_pad(), stream,
                     data, n, (int32_t *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace pgnn
