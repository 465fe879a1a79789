// Internal helpers shared by the HIP translation units of libpointgnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <exception>
#include <string>

#include "pointgnn_hip.h"

namespace pgnn {

std::string &last_error();

inline int fail(int code, const char *msg) {
  last_error() = msg;
  return code;
}

#define PGNN_HIP(expr)                                                        \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      char buf_[256];                                                         \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr,             \
               hipGetErrorString(e_), __FILE__, __LINE__);                    \
      pgnn::last_error() = buf_;                                              \
      return (int)e_;                                                         \
    }                                                                         \
  } while (0)

#define PGNN_REQUIRE(cond, code, msg)                                         \
  do {                                                                        \
    if (!(cond)) return pgnn::fail(code, msg);                                \
  } while (0)

// every extern "C" entry wraps its body: no exception crosses the ABI
#define PGNN_GUARD_BEGIN try {
#define PGNN_GUARD_END                                                        \
  }                                                                           \
  catch (const std::exception &ex) {                                          \
    return pgnn::fail(PGNN_E_INVALID, ex.what());                             \
  }                                                                           \
  catch (...) {                                                               \
    return pgnn::fail(PGNN_E_INVALID, "unknown C++ exception");               \
  }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
  char *base;
  size_t size, used;
  Arena(void *p, size_t n) : base((char *)p), size(n), used(0) {}
  template <typename T>
  T *take(size_t count) {
    size_t off = align_up(used, 256);
    size_t end = off + count * sizeof(T);
    used = end;
    if (end > size || base == nullptr) return nullptr;
    return (T *)(base + off);
  }
};

constexpr float kFloatLowest = -3.402823466e+38f;  // numeric_limits<float>::lowest()
constexpr uint32_t kFloatLowestBits = 0xFF7FFFFFu;

// float atomic max on global memory that is correct for any sign mix, given
// the target was initialised to a float value (lowest()): non-negative values
// order like signed ints, negative values order inversely as unsigned ints.
__device__ __forceinline__ void atomic_max_f32(float *addr, float v) {
  if (v >= 0.0f) {
    __hip_atomic_fetch_max((int *)addr, __float_as_int(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_fetch_min((unsigned int *)addr, __float_as_uint(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Wave priority of the graph-build kernels (s_setprio), compile-time, default 0.
// These kernels are chains of short dependent steps (a load, a ballot, a
// barrier ...); in the frame pipeline they share CUs with the persistent MFMA
// kernels of another frame's message passing and every step then takes 5-9x
// longer (tools/corun_probe.py: VALU chain 14 -> 69 ns per step, L2 pointer
// chase 117 -> 1018 ns, barrier 66 -> 511 ns).  Raising their priority does
// NOT change that (same table, priority 3: 67 / 1018 / 444 ns) -- the waves
// are not losing an arbitration they could win, the SIMD and the memory path
// are simply busy -- and it costs the MFMA kernels ~0.5 % once the pipeline
// is bound by them (-DPGNN_GRAPH_PRIO=3: 298.8 vs 300.3 frames/s).
#ifndef PGNN_GRAPH_PRIO
#define PGNN_GRAPH_PRIO 0
#endif
__device__ __forceinline__ void graph_prio() {
#if PGNN_GRAPH_PRIO > 0
  __builtin_amdgcn_s_setprio(PGNN_GRAPH_PRIO);
#endif
}

// Dynamic LDS every graph-build launch asks for on top of what it uses
// (tunable `graph_lds_pad`, bytes, default 0).  The frame pipeline sets it to
// 32 KiB together with `ws_reserve`: a workgroup of the weights-stationary
// kernels holds 133-153 KiB of a CU's 160 KiB, so a padded builder workgroup
// cannot be placed beside one -- it lands on the CUs those kernels leave free
// (ws_reserve) and runs there at full speed, instead of on a CU whose SIMDs
// are saturated with MFMAs, where every dependent step of the builder's
// chains takes 5-9x longer (tools/corun_probe.py).
extern int g_graph_lds_pad;
inline size_t graph_lds_pad() { return (size_t)g_graph_lds_pad; }
// ... and the cap on the workgroups of one builder launch (tunable
// `graph_max_wgs`, default 0 = none): every builder kernel strides over its
// work, so a grid of 8 covers any size.  With 8 CUs left free (one per XCD),
// launches of at most 8 workgroups are placed on them at once; the
// dispatcher does not route the later workgroups of a larger grid around
// CUs that lack the LDS -- they wait for the persistent kernel to end
// (profiles/r03_pipeline_probes.txt) -- and a persistent kernel of (CUs - 8) workgroups
// starts at once beside up to 8 resident builder workgroups, not beside more
// (same file).
extern int g_graph_max_wgs;
inline unsigned graph_grid(int64_t natural) {
  if (natural < 1) natural = 1;
  if (g_graph_max_wgs > 0 && natural > g_graph_max_wgs) natural = g_graph_max_wgs;
  return (unsigned)natural;
}
// threads per workgroup of the wave-per-item kernels: 4 waves normally, 16
// when the grid is capped (the few workgroups then carry all the parallelism)
inline unsigned graph_wide_block() { return g_graph_max_wgs > 0 ? 1024u : 256u; }

int device_cu_count();
int stream_cu_count(hipStream_t stream);
int ensure_dynamic_lds(const void *kernel, size_t bytes);
size_t device_max_lds();
// The tile-pool counters of the fused kernels ({next tile, finished
// workgroups}) must be zero when a launch starts.  The kernels hand them back
// zeroed, but an aborted launch or a caller sharing one buffer between
// streams would leave them poisoned and later launches would silently skip
// pool tiles: re-arm them in stream order before every launch that uses them.
// (a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node
// did not take effect on replays after the first -- the pooling kernel of a
// replayed ped_cyl frame then skipped pool tiles, tools/ped_check.py)
int arm_sched(int32_t *sched, hipStream_t stream);

// gnn.hip, for trainer.hip: y = gate > 0 ? x W : 0 (one layer, rows to HBM) --
// a backward dX pass with the ReluGrad of the layer below in its epilogue.
int mlp_rows_gated(const float *x, int64_t ld_x, int32_t nx, int64_t n_rows,
                   const pgnn_fc_layer *layer, const float *gate, int64_t ld_gate,
                   float *y, int64_t ld_y, hipStream_t stream);
// gnn.hip, for trainer.hip: a chain of layers on few rows in one launch, every
// intermediate output gated (backward chains) and written out (see gnn.hip)
struct RowsTap {
  float *mid;         // nullable: where layer li's output rows go
  const float *gate;  // nullable: out = gate > 0 ? out : 0 first
};
int mlp_rows_chain(const float *x, int64_t ld_x, int32_t nx, int64_t n_rows,
                   const pgnn_fc_layer *layers, int32_t n_layers,
                   const RowsTap *taps, const float *residual, int64_t ld_res,
                   int res_gate, float *y, int64_t ld_y, hipStream_t stream);

}  // namespace pgnn
