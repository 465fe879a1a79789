#include <cmath>
// Library plumbing: error state, device queries, host-side weight packing.
#include <string.h>

#include <mutex>
#include <vector>

#include "pgnn_common.h"

namespace pgnn {

std::string &last_error() {
  static thread_local std::string err;
  return err;
}

namespace {
__global__ void sched_zero_kernel(int32_t *sched) {
  if (threadIdx.x < PGNN_SCHED_WS_INTS) sched[threadIdx.x] = 0;
}
}  // namespace

int arm_sched(int32_t *sched, hipStream_t stream) {
  if (!sched) return 0;
  static_assert(PGNN_SCHED_WS_INTS <= 64, "one wave zeroes the counters");
  hipLaunchKernelGGL(sched_zero_kernel, dim3(1), dim3(64), 0, stream, sched);
  return (int)hipGetLastError();
}

int g_graph_lds_pad = 0;
int g_graph_max_wgs = 0;

int device_cu_count() {
  // per-device cache; the value never changes for a given ordinal
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) !=
            hipSuccess ||
        n <= 0)
      n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// CUs the kernels of `stream` may run on: the popcount of its CU mask
// (pgnn_stream_create_cu_mask), the whole device for ordinary streams.  The
// persistent kernels size their grids with it: a grid of 2 workgroups per
// DEVICE CU on a stream that owns fewer CUs would run in two waves.
namespace {
struct StreamCuEntry {
  hipStream_t s;
  int n;
};
std::mutex g_stream_cu_mu;
std::vector<StreamCuEntry> g_stream_cu_cache;
}  // namespace

// (pgnn_stream_destroy: a destroyed handle's address can be handed out again,
// with another mask or none)
void stream_cu_forget(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_stream_cu_mu);
  for (size_t i = 0; i < g_stream_cu_cache.size();) {
    if (g_stream_cu_cache[i].s == stream) {
      g_stream_cu_cache[i] = g_stream_cu_cache.back();
      g_stream_cu_cache.pop_back();
    } else {
      ++i;
    }
  }
}

int stream_cu_count(hipStream_t stream) {
  const int total = device_cu_count();
  if (stream == nullptr) return total;
  // the mask of a stream never changes: ask the runtime once per handle (this
  // sits on the launch path of every fused kernel)
  typedef StreamCuEntry Entry;
  std::mutex &mu = g_stream_cu_mu;
  std::vector<Entry> &cache = g_stream_cu_cache;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry &e : cache)
      if (e.s == stream) return e.n;
  }
  uint32_t mask[32] = {0};
  const int words = (total + 31) / 32;
  int n = 0;
  if (words > 32 ||
      hipExtStreamGetCUMask(stream, (uint32_t)words, mask) != hipSuccess) {
    (void)hipGetLastError();
    n = total;
  } else {
    for (int i = 0; i < total; ++i) n += (mask[i >> 5] >> (i & 31)) & 1u;
    if (n <= 0) n = total;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() < 4096) cache.push_back({stream, n});
  return n;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size
// instead of once per launch; fails (error code) when the device cannot give
// a workgroup that much LDS
int ensure_dynamic_lds(const void *kernel, size_t bytes) {
  struct Entry {
    const void *k;
    size_t b;
  };
  static std::mutex mu;
  static std::vector<Entry> done;
  std::lock_guard<std::mutex> lock(mu);
  for (const Entry &e : done)
    if (e.k == kernel && e.b >= bytes) return 0;
  PGNN_HIP(hipFuncSetAttribute(kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes));
  done.push_back({kernel, bytes});
  return 0;
}

// LDS a workgroup may allocate on the current device (160 KiB on gfx950)
size_t device_max_lds() {
  static size_t cached = 0;
  if (cached) return cached;
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock,
                            dev) != hipSuccess || v <= 0) {
    (void)hipGetLastError();
    return 64 * 1024;
  }
  cached = (size_t)v;
  return cached;
}

extern int g_scatter_rows_per_wave;
extern int g_mlp_blocks_per_cu;
extern int g_edge_msub;
extern int g_pool_msub;
extern int g_graph_debug;
extern int g_mlp_pool_pct;
extern int g_ws_xcds;
extern int g_ws_prio;
extern int g_b16_force;
extern int g_f16_pool;
extern int g_ws_pool_pct;
extern int g_ws_chunk;
extern int g_ws_balance;
extern int g_ws_reserve;
extern int g_mlp_debug;
extern void *g_mlp_ts;
extern int g_scatter_nt;
extern int g_wgrad_wg_target;
extern int g_train_h1;

}  // namespace pgnn

extern "C" int pgnn_version(void) { return 100; }

extern "C" const char *pgnn_last_error(void) {
  return pgnn::last_error().c_str();
}

extern "C" int pgnn_check_device_pointer(const void *p) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(p != nullptr, PGNN_E_INVALID, "null pointer");
  hipPointerAttribute_t attr;
  PGNN_HIP(hipPointerGetAttributes(&attr, p));
  PGNN_REQUIRE(attr.type == hipMemoryTypeDevice ||
                   attr.type == hipMemoryTypeManaged,
               PGNN_E_INVALID, "not a device pointer");
  return 0;
  PGNN_GUARD_END
}

// Streams restricted to a set of CUs (hipExtStreamCreateWithCUMask).  The
// frame pipeline uses two disjoint sets: a handful of CUs for the latency-bound
// graph construction of the next frame, the rest for the message passing of the
// current one.  Why: the fused MFMA kernels occupy every CU completely (two
// workgroups = all 512 VGPRs per SIMD lane and 159.4 of 160 KB LDS), so a
// kernel of another stream can only start at their kernel boundaries, ~1 ms
// apart -- a 40-launch dependent chain then takes several milliseconds -- and
// every workgroup it does get delays one persistent workgroup of the next
// fused kernel by its whole duration.  Mask bits are striped over the XCDs
// first (bit i -> XCD i % 8), so a contiguous bit range takes the same number
// of CUs from every XCD and the per-XCD round-robin of workgroups stays even.
extern "C" int pgnn_stream_create_cu_mask(int32_t cu_first, int32_t cu_count,
                                          int32_t complement,
                                          void **stream_out) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(stream_out, PGNN_E_INVALID, "stream_create_cu_mask: null output");
  *stream_out = nullptr;
  const int total = pgnn::device_cu_count();
  PGNN_REQUIRE(cu_first >= 0 && cu_count > 0 && cu_first + cu_count <= total,
               PGNN_E_INVALID, "stream_create_cu_mask: CU range outside the device");
  PGNN_REQUIRE(!complement || cu_count < total, PGNN_E_INVALID,
               "stream_create_cu_mask: empty complement");
  uint32_t mask[32] = {0};
  const int words = (total + 31) / 32;
  PGNN_REQUIRE(words <= 32, PGNN_E_UNSUPPORTED,
               "stream_create_cu_mask: more than 1024 CUs");
  for (int i = 0; i < total; ++i) {
    const bool in_range = i >= cu_first && i < cu_first + cu_count;
    if (in_range != (complement != 0)) mask[i >> 5] |= 1u << (i & 31);
  }
  hipStream_t s = nullptr;
  PGNN_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream_out = (void *)s;
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_stream_destroy(void *stream) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(stream, PGNN_E_INVALID, "stream_destroy: null stream");
  pgnn::stream_cu_forget((hipStream_t)stream);
  PGNN_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
  PGNN_GUARD_END
}

// Tuning hook for benchmarks (not part of the reference-facing surface).
extern "C" int pgnn_set_tunable(const char *key, int value) {
  if (!key) return PGNN_E_INVALID;
  if (!strcmp(key, "scatter_rows_per_wave")) {
    if (value < 0 || value > 65536) return PGNN_E_INVALID;
    pgnn::g_scatter_rows_per_wave = value;
    return 0;
  }
  if (!strcmp(key, "mlp_blocks_per_cu")) {
    if (value < 1 || value > 8) return PGNN_E_INVALID;
    pgnn::g_mlp_blocks_per_cu = value;
    return 0;
  }
  if (!strcmp(key, "edge_msub")) {
    pgnn::g_edge_msub = value;
    return 0;
  }
  if (!strcmp(key, "scatter_nt")) {
    pgnn::g_scatter_nt = value;
    return 0;
  }
  if (!strcmp(key, "mlp_debug")) {
#ifndef PGNN_DIAG
    if (value & 7)
      return pgnn::fail(PGNN_E_INVALID,
                        "mlp_debug bits 1/2/4 (timing ablations with wrong "
                        "results) need a -DPGNN_DIAG build");
#endif
    pgnn::g_mlp_debug = value;
    return 0;
  }
  if (!strcmp(key, "mlp_pool_pct")) {
    if (value < 0 || value > 90) return PGNN_E_INVALID;
    pgnn::g_mlp_pool_pct = value;
    return 0;
  }
  if (!strcmp(key, "ws_xcds")) {
    if (value < 1 || value > 64) return PGNN_E_INVALID;
    pgnn::g_ws_xcds = value;
    return 0;
  }
  if (!strcmp(key, "ws_pool_pct")) {
    if (value < 0 || value > 90) return PGNN_E_INVALID;
    pgnn::g_ws_pool_pct = value;
    return 0;
  }
  if (!strcmp(key, "ws_chunk")) {
    if (value < 1 || value > 64) return PGNN_E_INVALID;
    pgnn::g_ws_chunk = value;
    return 0;
  }
  if (!strcmp(key, "ws_balance")) {
    if (value < 0 || value > 2) return PGNN_E_INVALID;
    pgnn::g_ws_balance = value;
    return 0;
  }
  if (!strcmp(key, "train_h1")) {
    if (value < 0 || value > 1) return PGNN_E_INVALID;
    pgnn::g_train_h1 = value;
    return 0;
  }
  if (!strcmp(key, "ws_reserve")) {
    if (value < 0 || value > 128 || value % 8) return PGNN_E_INVALID;
    pgnn::g_ws_reserve = value;
    return 0;
  }
  if (!strcmp(key, "graph_lds_pad")) {
    if (value < 0 || value > 60 * 1024) return PGNN_E_INVALID;
    pgnn::g_graph_lds_pad = value;
    return 0;
  }
  if (!strcmp(key, "graph_max_wgs")) {
    if (value < 0 || value > 65535) return PGNN_E_INVALID;
    pgnn::g_graph_max_wgs = value;
    return 0;
  }
  if (!strcmp(key, "ws_prio")) {
    pgnn::g_ws_prio = value;
    return 0;
  }
  if (!strcmp(key, "b16_force")) {
    pgnn::g_b16_force = value != 0;
    return 0;
  }
  if (!strcmp(key, "f16_pool")) {
    pgnn::g_f16_pool = value;
    return 0;
  }
  if (!strcmp(key, "graph_debug")) {
#ifndef PGNN_DIAG
    if (value & 1)
      return pgnn::fail(PGNN_E_INVALID,
                        "graph_debug bit 1 (skip the kd-tree replica: "
                        "non-reference tie order) needs a -DPGNN_DIAG build");
#endif
    pgnn::g_graph_debug = value;
    return 0;
  }
  if (!strcmp(key, "pool_msub")) {
    pgnn::g_pool_msub = value;
    return 0;
  }
  if (!strcmp(key, "wgrad_wg_target")) {
    if (value < 1) return pgnn::fail(PGNN_E_INVALID, "wgrad_wg_target < 1");
    pgnn::g_wgrad_wg_target = value;
    return 0;
  }
  return pgnn::fail(PGNN_E_INVALID, "unknown tunable");
}

// Profiling hook: per-tile cycle stamps of the fused kernels are written to
// this device buffer (int64 [grid * 32 * 4]); NULL disables.
extern "C" int pgnn_set_debug_buffer(void *device_ptr) {
  pgnn::g_mlp_ts = device_ptr;
  return 0;
}

extern "C" size_t pgnn_packed_fc_floats(int32_t k_in, int32_t n_out) {
  if (k_in <= 0 || n_out <= 0) return 0;
  size_t kq = ((size_t)k_in + 15) / 16, nt = ((size_t)n_out + 15) / 16;
  return kq * nt * 256 + nt * 16;
}

// ---- split-bf16 weight image (edge_ws_bf16.h) ----------------------------------
namespace {
inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);  // finite inputs
}
inline float bf16_as_float(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
}  // namespace

extern "C" size_t pgnn_packed_fc_bf16x3_bytes(int32_t k_in, int32_t n_out) {
  if (k_in <= 0 || n_out <= 0) return 0;
  const size_t kb = (k_in + 31) / 32, nt = (n_out + 15) / 16;
  return kb * nt * 3 * 1024 + nt * 16 * sizeof(float);
}

extern "C" int pgnn_pack_fc_bf16x3(const float *w, const float *b, int32_t k_in,
                                   int32_t n_out, void *image) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(w && image && k_in > 0 && n_out > 0, PGNN_E_INVALID,
               "pack_fc_bf16x3: bad argument");
  const int kb_n = (k_in + 31) / 32, nt = (n_out + 15) / 16;
  uint16_t *img = reinterpret_cast<uint16_t *>(image);
  for (int kb = 0; kb < kb_n; ++kb)
    for (int t = 0; t < nt; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * kb + 8 * (lane >> 4) + j;
          const int n = 16 * t + (lane & 15);
          const float x = (k < k_in && n < n_out) ? w[(size_t)k * n_out + n] : 0.0f;
          // x = x0 + x1 + x2 exactly: both residuals are exact in fp32
          const uint16_t p0 = bf16_rne(x);
          const float r1 = x - bf16_as_float(p0);
          const uint16_t p1 = bf16_rne(r1);
          const float r2 = r1 - bf16_as_float(p1);
          const uint16_t p2 = bf16_rne(r2);
          const size_t frag = ((size_t)kb * nt + t) * 3;
          img[((frag + 0) * 64 + lane) * 8 + j] = p0;
          img[((frag + 1) * 64 + lane) * 8 + j] = p1;
          img[((frag + 2) * 64 + lane) * 8 + j] = p2;
        }
  float *bias = reinterpret_cast<float *>(reinterpret_cast<char *>(image) +
                                          (size_t)kb_n * nt * 3 * 1024);
  for (int n = 0; n < nt * 16; ++n) bias[n] = (b && n < n_out) ? b[n] : 0.0f;
  return 0;
  PGNN_GUARD_END
}

extern "C" size_t pgnn_packed_fc_f16x2_bytes(int32_t k_in, int32_t n_out) {
  if (k_in <= 0 || n_out <= 0) return 0;
  const size_t kb = (k_in + 31) / 32, nt = (n_out + 15) / 16;
  return kb * nt * 2 * 1024 + nt * 16 * sizeof(float);
}

namespace {
// acc_order: slot (g, j) of block kb holds feature 32 kb + 4 g + j (j < 4) or
// 32 kb + 16 + 4 g + j - 4 -- the order in which a lane of the fp32 MFMA
// accumulators of the layer below holds a row's features (pool_ws_f16.h)
int pack_fc_f16x2(const float *w, const float *b, int32_t k_in, int32_t n_out,
                  void *image, bool acc_order) {
  PGNN_REQUIRE(w && image && k_in > 0 && n_out > 0, PGNN_E_INVALID,
               "pack_fc_f16x2: bad argument");
  const int kb_n = (k_in + 31) / 32, nt = (n_out + 15) / 16;
  for (size_t i = 0; i < (size_t)k_in * n_out; ++i)
    PGNN_REQUIRE(std::fabs(w[i]) < 32768.0f, PGNN_E_UNSUPPORTED,
                 "pack_fc_f16x2: a weight outside fp16's range");
  _Float16 *img = reinterpret_cast<_Float16 *>(image);
  for (int kb = 0; kb < kb_n; ++kb)
    for (int t = 0; t < nt; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int g = lane >> 4;
          const int k = 32 * kb + (!acc_order ? 8 * g + j
                                   : j < 4    ? 4 * g + j
                                              : 16 + 4 * g + (j - 4));
          const int n = 16 * t + (lane & 15);
          const float x = (k < k_in && n < n_out) ? w[(size_t)k * n_out + n] : 0.0f;
          // x ~ w0 + w1' / 2^11: the residual is exact in fp32, the scaling
          // (a power of two) keeps it out of fp16's subnormals
          const _Float16 w0 = (_Float16)x;
          const _Float16 w1 = (_Float16)((x - (float)w0) * 2048.0f);
          const size_t frag = ((size_t)kb * nt + t) * 2;
          img[((frag + 0) * 64 + lane) * 8 + j] = w0;
          img[((frag + 1) * 64 + lane) * 8 + j] = w1;
        }
  float *bias = reinterpret_cast<float *>(reinterpret_cast<char *>(image) +
                                          (size_t)kb_n * nt * 2 * 1024);
  for (int n = 0; n < nt * 16; ++n) bias[n] = (b && n < n_out) ? b[n] : 0.0f;
  return 0;
}
}  // namespace

extern "C" int pgnn_pack_fc_f16x2(const float *w, const float *b, int32_t k_in,
                                  int32_t n_out, void *image) {
  PGNN_GUARD_BEGIN
  return pack_fc_f16x2(w, b, k_in, n_out, image, false);
  PGNN_GUARD_END
}

extern "C" int pgnn_pack_fc_f16x2_acc(const float *w, const float *b,
                                      int32_t k_in, int32_t n_out, void *image) {
  PGNN_GUARD_BEGIN
  return pack_fc_f16x2(w, b, k_in, n_out, image, true);
  PGNN_GUARD_END
}

extern "C" int pgnn_pack_fc(const float *w, const float *b, int32_t k_in,
                            int32_t n_out, float *packed) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(w && packed && k_in > 0 && n_out > 0, PGNN_E_INVALID,
               "pack_fc: bad argument");
  const int kq = (k_in + 15) / 16, nt = (n_out + 15) / 16;
  for (int q = 0; q < kq; ++q)
    for (int t = 0; t < nt; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) {
          const int k = 16 * q + 4 * (lane >> 4) + s;
          const int n = 16 * t + (lane & 15);
          packed[(((size_t)q * nt + t) * 64 + lane) * 4 + s] =
              (k < k_in && n < n_out) ? w[(size_t)k * n_out + n] : 0.0f;
        }
  float *bias = packed + (size_t)kq * nt * 256;
  for (int n = 0; n < nt * 16; ++n) bias[n] = (b && n < n_out) ? b[n] : 0.0f;
  return 0;
  PGNN_GUARD_END
}
