// Standalone scatter-max (tf.math.unsorted_segment_max, models/gnn.py:106-109).
//
// HBM-bound: every input byte is read exactly once, in full rows, by
// consecutive lanes (16 B per lane when the layout allows).  A wave owns a
// contiguous range of rows, keeps a running column-wise max in registers while
// the segment id stays the same and flushes on every id change.  With sorted
// ids a run that starts and ends inside the wave's range is a complete segment
// and is written with plain stores; only the (at most two) runs that touch the
// range boundary go through float atomic-max.  Unsorted ids degrade to
// "every run is flushed atomically", which is still correct for any order.
// The output is pre-filled with float lowest (TF's value for empty segments and
// the identity of the atomic max) by a D32 memset on the same stream.
#include "pgnn_common.h"

namespace pgnn {
int g_scatter_rows_per_wave = 0;  // 0 = auto
int g_scatter_nt = 1;            // non-temporal row loads
}

namespace {

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  typedef float4 type;
  static __device__ __forceinline__ type load(const float *p) {
    // streamed once: non-temporal keeps the rows out of L2's way
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ type load_plain(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
  }
  static __device__ __forceinline__ type vmax(type a, type b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z),
                       fmaxf(a.w, b.w));
  }
  static __device__ __forceinline__ void store(float *p, type v) {
    *reinterpret_cast<float4 *>(p) = v;
  }
  static __device__ __forceinline__ void amax(float *p, type v) {
    pgnn::atomic_max_f32(p + 0, v.x + 0.0f);
    pgnn::atomic_max_f32(p + 1, v.y + 0.0f);
    pgnn::atomic_max_f32(p + 2, v.z + 0.0f);
    pgnn::atomic_max_f32(p + 3, v.w + 0.0f);
  }
};
template <>
struct Vec<1> {
  typedef float type;
  static __device__ __forceinline__ type load(const float *p) {
    return __builtin_nontemporal_load(p);
  }
  static __device__ __forceinline__ type load_plain(const float *p) { return *p; }
  static __device__ __forceinline__ type vmax(type a, type b) {
    return fmaxf(a, b);
  }
  static __device__ __forceinline__ void store(float *p, type v) { *p = v; }
  static __device__ __forceinline__ void amax(float *p, type v) {
    pgnn::atomic_max_f32(p, v + 0.0f);
  }
};

// VEC floats per lane per column group, NJ column groups per lane, BATCH rows
// loaded before any is consumed (memory-level parallelism).
template <int VEC, int NJ, int BATCH, bool NTL>
__global__ __launch_bounds__(256) void scatter_max_kernel(
    const float *__restrict__ data, int64_t ld, const int32_t *__restrict__ seg,
    int64_t n_rows, int32_t n_colv /* columns in VEC units */,
    int32_t num_segments, float *__restrict__ out, int64_t ldo, int32_t sorted,
    int32_t rows_per_wave) {
  typedef Vec<VEC> V;
  typedef typename V::type vec_t;
  static_assert(BATCH <= 64, "batch ids are held one per lane");
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t n_chunks = (n_rows + rows_per_wave - 1) / rows_per_wave;

  int colv[NJ];
  bool col_ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int c = lane + 64 * j;
    col_ok[j] = c < n_colv;
    colv[j] = (c < n_colv ? c : n_colv - 1) * VEC;  // clamp: always a valid load
  }

  for (int64_t chunk = wave0; chunk < n_chunks; chunk += n_waves) {
    const int64_t r0 = chunk * rows_per_wave;
    const int64_t r1 = (r0 + rows_per_wave < n_rows) ? r0 + rows_per_wave : n_rows;
    int cur = -1;
    int64_t run_start = r0;
    vec_t acc[NJ];

    auto flush = [&](int64_t run_end) {
      if (cur < 0 || cur >= num_segments) return;
      bool whole = sorted != 0;
      if (whole && run_start == r0 && r0 > 0)
        whole = __builtin_amdgcn_readfirstlane(seg[r0 - 1]) != cur;
      if (whole && run_end == r1 && r1 < n_rows)
        whole = __builtin_amdgcn_readfirstlane(seg[r1]) != cur;
      float *orow = out + (int64_t)cur * ldo;
      if (whole) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (col_ok[j]) V::store(orow + colv[j], acc[j]);
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (col_ok[j]) V::amax(orow + colv[j], acc[j]);
      }
    };

    for (int64_t rb = r0; rb < r1; rb += BATCH) {
      // one id per lane for this batch; rows past the range are clamped to the
      // last row (max is idempotent, so duplicates are harmless)
      int64_t my_row = rb + (lane < BATCH ? lane : BATCH - 1);
      if (my_row > r1 - 1) my_row = r1 - 1;
      const int my_seg = seg[my_row];
      vec_t v[BATCH][NJ];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        int64_t row = rb + k;
        if (row > r1 - 1) row = r1 - 1;
        const float *p = data + row * ld;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          v[k][j] = NTL ? V::load(p + colv[j]) : V::load_plain(p + colv[j]);
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int s = __builtin_amdgcn_readlane(my_seg, k);
        int64_t row = rb + k;
        if (row > r1 - 1) row = r1 - 1;
        if (s != cur) {
          flush(row);
          cur = s;
          run_start = row;
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] = v[k][j];
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] = V::vmax(acc[j], v[k][j]);
        }
      }
    }
    flush(r1);
  }
}

__global__ void fill_rows_kernel(float *out, int64_t ldo, int32_t n_cols,
                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    out[(i / n_cols) * ldo + (i % n_cols)] = pgnn::kFloatLowest;
}

template <int VEC, int NJ, int BATCH>
int launch(const float *data, int64_t ld, const int32_t *seg, int64_t n_rows,
           int32_t n_colv, int32_t num_segments, float *out, int64_t ldo,
           int32_t sorted, hipStream_t stream) {
  int rpw = pgnn::g_scatter_rows_per_wave;
  if (rpw <= 0) {
    // ~8 waves per CU, each owning one contiguous row range: long ranges mean
    // few boundary runs (atomics); measured optimum 256-512 rows at E~500k
    const int64_t waves = (int64_t)pgnn::device_cu_count() * 8;
    int64_t r = (n_rows + waves - 1) / waves;
    r = (r + BATCH - 1) / BATCH * BATCH;
    rpw = (int)(r < 32 ? 32 : (r > 4096 ? 4096 : r));
  }
  if (rpw < BATCH) rpw = BATCH;
  int64_t n_chunks = (n_rows + rpw - 1) / rpw;
  int64_t blocks = (n_chunks + 3) / 4;
  int64_t cap = (int64_t)pgnn::device_cu_count() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (pgnn::g_scatter_nt)
    hipLaunchKernelGGL((scatter_max_kernel<VEC, NJ, BATCH, true>),
                       dim3((unsigned)blocks), dim3(256), 0, stream, data, ld, seg,
                       n_rows, n_colv, num_segments, out, ldo, sorted, rpw);
  else
    hipLaunchKernelGGL((scatter_max_kernel<VEC, NJ, BATCH, false>),
                       dim3((unsigned)blocks), dim3(256), 0, stream, data, ld, seg,
                       n_rows, n_colv, num_segments, out, ldo, sorted, rpw);
  PGNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int pgnn_scatter_max_f32(const float *data, int64_t ld_data,
                                    const int32_t *seg_ids, int64_t n_rows,
                                    int32_t n_cols, int32_t num_segments,
                                    float *out, int64_t ld_out,
                                    int32_t ids_sorted, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && n_cols > 0 && num_segments >= 0, PGNN_E_INVALID,
               "scatter_max: negative size");
  PGNN_REQUIRE(ld_data >= n_cols && ld_out >= n_cols, PGNN_E_INVALID,
               "scatter_max: row stride smaller than n_cols");
  if (num_segments == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "scatter_max: out is null");
  // empty-segment value + atomic identity
  if (ld_out == n_cols) {
    PGNN_HIP(hipMemsetD32Async((hipDeviceptr_t)out, (int)pgnn::kFloatLowestBits,
                               (size_t)num_segments * n_cols, stream));
  } else {
    const int64_t total = (int64_t)num_segments * n_cols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       stream, out, ld_out, n_cols, total);
    PGNN_HIP(hipGetLastError());
  }
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(data != nullptr && seg_ids != nullptr, PGNN_E_INVALID,
               "scatter_max: null input");
  const bool vec4 = (n_cols % 4 == 0) && (ld_data % 4 == 0) &&
                    (ld_out % 4 == 0) && ((uintptr_t)data % 16 == 0) &&
                    ((uintptr_t)out % 16 == 0);
  if (vec4) {
    const int32_t nv = n_cols / 4;
    if (nv <= 64)
      return launch<4, 1, 16>(data, ld_data, seg_ids, n_rows, nv, num_segments,
                              out, ld_out, ids_sorted, stream);
    if (nv <= 128)
      return launch<4, 2, 8>(data, ld_data, seg_ids, n_rows, nv, num_segments,
                             out, ld_out, ids_sorted, stream);
    if (nv <= 256)
      return launch<4, 4, 4>(data, ld_data, seg_ids, n_rows, nv, num_segments,
                             out, ld_out, ids_sorted, stream);
    if (nv <= 512)
      return launch<4, 8, 2>(data, ld_data, seg_ids, n_rows, nv, num_segments,
                             out, ld_out, ids_sorted, stream);
    return pgnn::fail(PGNN_E_UNSUPPORTED, "scatter_max: n_cols > 2048");
  }
  if (n_cols <= 64)
    return launch<1, 1, 16>(data, ld_data, seg_ids, n_rows, n_cols,
                            num_segments, out, ld_out, ids_sorted, stream);
  if (n_cols <= 256)
    return launch<1, 4, 8>(data, ld_data, seg_ids, n_rows, n_cols, num_segments,
                           out, ld_out, ids_sorted, stream);
  if (n_cols <= 1024)
    return launch<1, 16, 2>(data, ld_data, seg_ids, n_rows, n_cols,
                            num_segments, out, ld_out, ids_sorted, stream);
  return pgnn::fail(PGNN_E_UNSUPPORTED,
                    "scatter_max: unaligned n_cols > 1024");
  PGNN_GUARD_END
}

// ---- unsorted_segment_sum / unsorted_segment_mean (models/gnn.py:111-119) ----
// Registered by the reference next to scatter-max but used by no shipped config:
// a plain float atomic-add per element (TensorFlow's GPU kernel does the same,
// so the summation order is unspecified there too); mean = sum / max(count, 1).
namespace {

__global__ void scatter_add_kernel(const float *__restrict__ data, int64_t ld,
                                   const int32_t *__restrict__ seg, int64_t n_rows,
                                   int32_t n_cols, int32_t num_segments,
                                   float *__restrict__ out, int64_t ldo,
                                   int32_t *__restrict__ counts) {
  const int64_t total = n_rows * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n_cols;
    const int c = (int)(i - r * n_cols);
    const int s = seg[r];
    if (s < 0 || s >= num_segments) continue;
    atomicAdd(out + (int64_t)s * ldo + c, data[r * ld + c]);
    if (counts && c == 0) atomicAdd(counts + s, 1);
  }
}

__global__ void scatter_mean_finish_kernel(float *__restrict__ out, int64_t ldo,
                                           int32_t n_cols, int32_t num_segments,
                                           const int32_t *__restrict__ counts) {
  const int64_t total = (int64_t)num_segments * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i / n_cols;
    const int c = (int)(i - s * n_cols);
    const int n = counts[s];
    out[s * ldo + c] = out[s * ldo + c] / (float)(n > 1 ? n : 1);
  }
}

}  // namespace

extern "C" int pgnn_scatter_sum_f32(const float *data, int64_t ld_data,
                                    const int32_t *seg_ids, int64_t n_rows,
                                    int32_t n_cols, int32_t num_segments,
                                    float *out, int64_t ld_out, int32_t mean,
                                    int32_t *counts_ws, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && n_cols > 0 && num_segments >= 0, PGNN_E_INVALID,
               "scatter_sum: negative size");
  PGNN_REQUIRE(ld_data >= n_cols && ld_out >= n_cols, PGNN_E_INVALID,
               "scatter_sum: row stride smaller than n_cols");
  PGNN_REQUIRE(!mean || counts_ws, PGNN_E_INVALID,
               "scatter_sum: mean needs the int32[num_segments] workspace");
  if (num_segments == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "scatter_sum: out is null");
  PGNN_HIP(hipMemset2DAsync(out, (size_t)ld_out * 4, 0, (size_t)n_cols * 4,
                            (size_t)num_segments, stream));
  if (mean)
    PGNN_HIP(hipMemsetAsync(counts_ws, 0, (size_t)num_segments * 4, stream));
  if (n_rows > 0) {
    PGNN_REQUIRE(data != nullptr && seg_ids != nullptr, PGNN_E_INVALID,
                 "scatter_sum: null input");
    const int64_t total = n_rows * n_cols;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)pgnn::device_cu_count() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(scatter_add_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       stream, data, ld_data, seg_ids, n_rows, n_cols,
                       num_segments, out, ld_out, mean ? counts_ws : nullptr);
    PGNN_HIP(hipGetLastError());
  }
  if (mean) {
    const int64_t total = (int64_t)num_segments * n_cols;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_mean_finish_kernel, dim3((unsigned)blocks),
                       dim3(256), 0, stream, out, ld_out, n_cols, num_segments,
                       counts_ws);
    PGNN_HIP(hipGetLastError());
  }
  return 0;
  PGNN_GUARD_END
}
