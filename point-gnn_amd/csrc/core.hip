// Library plumbing: error state, device queries, host-side weight packing.
#include <string.h>

#include <mutex>

#include "pgnn_common.h"

namespace pgnn {

std::string &last_error() {
  static thread_local std::string err;
  return err;
}

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

extern int g_scatter_rows_per_wave;
extern int g_mlp_blocks_per_cu;
extern int g_edge_msub;
extern int g_pool_msub;
extern int g_mlp_chunks_per_wg;
extern int g_graph_debug;
extern int g_mlp_debug;
extern void *g_mlp_ts;
extern int g_scatter_nt;
extern int g_wgrad_wg_target;

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
    pgnn::g_mlp_debug = value;
    return 0;
  }
  if (!strcmp(key, "mlp_chunks_per_wg")) {
    if (value < 0 || value > 64) return PGNN_E_INVALID;
    pgnn::g_mlp_chunks_per_wg = value;
    return 0;
  }
  if (!strcmp(key, "graph_debug")) {
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
