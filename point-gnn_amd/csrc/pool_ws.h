// Weights-stationary form of the fused PointSetPooling kernel (gnn.py:256-277)
// for the car point MLP 4 -> 32 -> 64 -> 128 -> 300:
//     out[d] = max over edges (s -> d) of MLP([f(s), xyz(s) - xyz(kp(d))]).
//
// Same design as edge_ws.h.  The last layer (128 -> 300: 79 % of the kernel's
// MFMAs, 8 x 19 fragments = 152 KiB) stays in LDS for the life of the kernel;
// an 8-wave workgroup owns a CU; every wave is autonomous: it gathers the 4
// input features of its 16 rows straight into the B-operand layout, runs the
// three hidden layers in registers (mlp_engine.h reg_layer: weights are the A
// operand, a layer's accumulators ARE the next layer's B operands; those
// 42 KiB of weights come from L2 / L1 per tile as in the LDS-tile kernel's
// PRO_POOL_R3 form), multiplies the [16 x 128] result with the LDS-resident
// fragments in three column blocks of 7 / 6 / 6 tiles, and keeps the running
// max of the open segment per lane (edge_ws.h, ws_epilogue).  No workgroup
// barrier after the weights are in LDS, no activation tile in LDS, no weight
// stream for the wide layer.  Every output element sees the same sequence of
// MFMA updates as in the LDS-tile kernel: bit-identical (tested; `mlp_debug`
// bit 8192 selects the LDS-tile kernel).
#pragma once
#include "edge_ws.h"

namespace pgnn {

struct PoolWsArgs {
  const float *feat;  // [n_points, nfeat]
  int nfeat;
  const float *xyz;       // [n_points, 3]
  const int32_t *kp;      // [num_segments] point index of each keypoint
  const int32_t *edges;   // [n_edges, 2] rows (point, keypoint)
  int64_t n_edges;
  const int32_t *n_dev;   // capacity form (nullable), see EdgeWsArgs
  LayerDev l0, l1, l2;    // hidden layers (packed weights in global memory)
  const float *wp;        // last layer: packed weights, bias follows
  int kq, nt;             // ... its K groups (8) and column tiles (19)
  int relu_from;
  float *out;
  int64_t ldo;
  int num_segments;
  int sorted;
  int prio;
  long long *ts;
  int32_t *sched;
  int pool_pct;
  int chunk;
  int slices;  // pool_split.h only: row slices (workgroup b -> slice b % slices)
  // training forward (EMIT kernel only): the activations of the four layers
  // are ALSO written -- a1 [E,32], a2 [E,64], a3 [E,128], a4 [E, ld4] -- for
  // the backward (a4 against `out` gives the arg-max rows; rows and maxima
  // come from the same accumulators)
  float *a1_out, *a2_out, *a3_out, *a4_out;
  int64_t ld4;
  // pool_ws_f16.h only: the 64 -> 128 layer's two-part fp16 image (nullable:
  // that layer in fp32 through l2)
  const void *l2_f16;
};

// acc[t] = sum_q W[q][tb + t]^T h[q]  for the NTB column tiles from tb on
template <int KQ, int NT, int NTB>
__device__ __forceinline__ void pool_ws_block(const v4f *const (&wfrag)[3],
                                              int tb, const v4f (&h)[KQ],
                                              v4f (&acc)[NTB]) {
  v4f w[2][NTB];
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    w[0][t] = wfrag[(tb + t) >> 6][((tb + t) & 63) * 64];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): see edge_ws.h
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < KQ) {
#pragma unroll
      for (int t = 0; t < NTB; ++t)
        w[(q + 1) & 1][t] = wfrag[((q + 1) * NT + tb + t) >> 6]
                                 [(((q + 1) * NT + tb + t) & 63) * 64];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < NTB; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q & 1][t][s], h[q][s],
                                                      acc[t], 0, 0, 0);
#if PGNN_WS_SCHED == 2
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 3, 0);
      if (q + 1 < KQ)
        __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, 1, 0);
    }
    // (three MFMAs per request: the last fragment is then requested a
    // quarter of a group -- ~200 cycles -- before the group boundary's wait)
    __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, NTB, 0);
#else
    if (q + 1 < KQ)
      __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, NTB, 0);
    __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 4 * NTB, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tiles [tile_first, tile_last) of 16 edge rows; last layer = 8 K groups x 19
// column tiles in LDS
template <bool EMIT>
__device__ __forceinline__ void pool_ws_body(const PoolWsArgs &a,
                                             const v4f *__restrict__ wl,
                                             const float *bias_lds,
                                             int64_t tile_first,
                                             int64_t tile_last, int lane,
                                             long long *tsw, int &stamped,
                                             const int64_t E) {
  constexpr int KQ = 8, NT = 19;
  if (tile_first >= tile_last) return;
  const int n = lane & 15;
  const int64_t e_first = tile_first * 16;
  const int64_t e_end = tile_last * 16 < E ? tile_last * 16 : E;
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);
  int cur_d = e_first > 0 ? a.edges[2 * (e_first - 1) + 1] : -1;
  int d_after = e_end < E ? a.edges[2 * e_end + 1] : -1;
  cur_d = __builtin_amdgcn_readfirstlane(cur_d);
  d_after = __builtin_amdgcn_readfirstlane(d_after);
  WsRun run = {cur_d, false, false};
  v4f carry0[7], carry1[6], carry2[6];  // one per column block
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    carry0[t] = (v4f){kFloatLowest, kFloatLowest, kFloatLowest, kFloatLowest};
    if (t < 6) {
      carry1[t] = (v4f){kFloatLowest, kFloatLowest, kFloatLowest, kFloatLowest};
      carry2[t] = (v4f){kFloatLowest, kFloatLowest, kFloatLowest, kFloatLowest};
    }
  }
  const float inf = opaque_inf();
  // (point, keypoint) of a tile's rows are requested one tile ahead, the
  // keypoint's point index (second level of the chain) half a tile ahead;
  // unconditional clamped requests, validity selects at the use (edge_ws.h)
  bool nxt_ok = e_first + n < E;
  int2 nxt = e2[nxt_ok ? e_first + n : 0];
  int nxt_k;
  {
    const int d0 = nxt_ok ? nxt.y : 0;
    nxt_k = a.kp[((unsigned)d0 < (unsigned)a.num_segments) ? d0 : 0];
  }
  for (int64_t tile = tile_first;; ++tile) {
    const bool fin = tile >= tile_last;
    const int64_t e0 = tile * 16;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int lz;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    const v4f *const wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    long long *tst = nullptr;
    if (tsw && !fin && stamped < kWsStampTiles) tst = tsw + 8 + 4 * stamped++;
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[0] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned starts = 1u;  // virtual tile: "row 0 opens a run"
    int my_d = -1;
    v4f h3[KQ];
    if (fin) {
#pragma unroll
      for (int q = 0; q < KQ; ++q) h3[q] = (v4f){0.f, 0.f, 0.f, 0.f};
    } else {
      const bool ok = nxt_ok;
      const int my_s = ok ? nxt.x : 0;
      my_d = ok ? nxt.y : -1;
      const int my_k = nxt_k;
      nxt_ok = tile + 1 < tile_last && e0 + 16 + n < E;
      nxt = e2[nxt_ok ? e0 + 16 + n : 0];
      // lane (g, n) holds input features 4g .. 4g+3 of row n: the B operand
      // of the first layer (K group 0); [f(src), xyz(src) - xyz(kp(dst))]
      v4f x[1];
      x[0] = (v4f){0.f, 0.f, 0.f, 0.f};
      if (ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 4 * g + i;  // input column
          float v = 0.0f;
          if (c < a.nfeat) {
            v = a.feat[(int64_t)my_s * a.nfeat + c];
          } else if (c < a.nfeat + 3) {
            // points within a set use coordinates relative to its keypoint
            const int ax = c - a.nfeat;
            v = a.xyz[3 * (int64_t)my_s + ax] - a.xyz[3 * (int64_t)my_k + ax];
          }
          x[0][i] = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tst) {
        const long long c = __builtin_readcyclecounter();
        if (lane == 0) tst[1] = c;
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      // hidden layers in registers
      v4f h1[2], h2[4];
#ifdef PGNN_POOL_ABL_NO_HIDDEN  // timing ablation (wrong results): no hidden layers
#pragma unroll
      for (int q = 0; q < KQ; ++q) h3[q] = x[0];
      h1[0] = h1[1] = x[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) h2[q] = x[0];
#else
      reg_layer<1, 2>(a.l0, lane, x, h1);
      reg_layer<2, 4>(a.l1, lane, h1, h2);
      reg_layer<4, 8>(a.l2, lane, h2, h3);
#endif
      if constexpr (EMIT) {
        // hidden activations: lane (g, n) holds features 16 q + 4 g .. + 3 of
        // row n -- one 16-byte store per K group
        if (e0 + n < E) {
          const int64_t row = e0 + n;
#pragma unroll
          for (int q = 0; q < 2; ++q)
            *reinterpret_cast<v4f *>(a.a1_out + row * 32 + 16 * q + 4 * g) = h1[q];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<v4f *>(a.a2_out + row * 64 + 16 * q + 4 * g) = h2[q];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            *reinterpret_cast<v4f *>(a.a3_out + row * 128 + 16 * q + 4 * g) = h3[q];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // second level of the next tile's index chain (its pair has landed)
      {
        const int dn = nxt_ok ? nxt.y : 0;
        nxt_k = a.kp[((unsigned)dn < (unsigned)a.num_segments) ? dn : 0];
      }
      const int up = __shfl_up(my_d, 1);
      const int prev = n == 0 ? run.cur_d : up;
      starts = (unsigned)(__ballot(my_d != prev) & 0xFFFFull);
    }
    // last layer in three column blocks of 7 / 6 / 6 tiles (register budget:
    // 76 carry + 32 h3 + 28 acc + 56 fragment stages); every block replays the tile's run structure on its own columns from the same
    // incoming state
    const WsRun run0 = run;
#define PGNN_POOL_WS_BLOCK(NTB, TB, CARRY)                                     \
    {                                                                          \
      v4f acc[NTB];                                                            \
      if (!fin) {                                                              \
        pool_ws_block<KQ, NT, NTB>(wfrag, TB, h3, acc);                        \
        if constexpr (EMIT) { /* the rows: same accumulators, bias, ReLU */    \
          if (e0 + n < E) {                                                    \
            float *dstp = a.a4_out + (e0 + n) * a.ld4 + 16 * (TB) + 4 * g;     \
            _Pragma("unroll")                                                  \
            for (int t = 0; t < NTB; ++t) {                                    \
              const v4f bb = *reinterpret_cast<const v4f *>(                   \
                  bias_lds + 16 * ((TB) + t) + 4 * g);                         \
              v4f y;                                                           \
              _Pragma("unroll")                                                \
              for (int r = 0; r < 4; ++r) {                                    \
                float xx = acc[t][r] + bb[r];                                  \
                if (16 * ((TB) + t) + 4 * g + r >= a.relu_from)                \
                  xx = xx > 0.0f ? xx : 0.0f;                                  \
                y[r] = xx;                                                     \
              }                                                                \
              *reinterpret_cast<v4f *>(dstp + 16 * t) = y;                     \
            }                                                                  \
          }                                                                    \
        }                                                                      \
      } else { /* defined on both paths: an undef phi would keep every */      \
        _Pragma("unroll") /* block's accumulators live round the loop */       \
        for (int t = 0; t < NTB; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};      \
      }                                                                        \
      if (a.prio) __builtin_amdgcn_s_setprio(3);                               \
      run = run0;                                                              \
      ws_epilogue<NTB>(a, bias_lds + 16 * (TB), TB, lane, acc, CARRY, starts,  \
                       my_d, run, fin, d_after, inf);                          \
      __builtin_amdgcn_s_setprio(0);                                           \
    }
    PGNN_POOL_WS_BLOCK(7, 0, carry0)
    PGNN_POOL_WS_BLOCK(6, 7, carry1)
    PGNN_POOL_WS_BLOCK(6, 13, carry2)
#undef PGNN_POOL_WS_BLOCK
    if (fin) break;
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) {
        tst[2] = c;
        tst[3] = c;
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

template <bool EMIT = false>
__global__ __launch_bounds__(64 * kWsWaves) void pool_ws_kernel(PoolWsArgs a) {
  constexpr int KQ = 8, NT = 19;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4f *wl = reinterpret_cast<v4f *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KQ * NT * 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(a.wp);
    // all 19 fragment requests of a wave in flight before the first LDS
    // write (see edge_ws_kernel: the plain copy loop is 19 dependent round
    // trips)
    constexpr int PER = (KQ * NT + kWsWaves - 1) / kWsWaves;
    v4f tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      tmp[i] = src[(size_t)(f < KQ * NT ? f : 0) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      if (f < KQ * NT) wl[(size_t)f * 64 + lane] = tmp[i];
    }
    if ((int)threadIdx.x < 16 * NT)
      bias_lds[threadIdx.x] = a.wp[(size_t)KQ * NT * 256 + threadIdx.x];
  }
  __syncthreads();
  // static ranges for (100 - pool_pct) % of the 16-row tiles, the rest in
  // chunks from a pool (edge_ws.h); ~10 tiles per wave at E0 = 350k, so the
  // pool works in single tiles
  int64_t n_edges = a.n_edges;
  if (a.n_dev) {
    const int64_t nd = *a.n_dev;
    n_edges = nd < n_edges ? nd : n_edges;
  }
  const int64_t n_wt = (n_edges + 15) / 16;
  const int64_t nw = (int64_t)gridDim.x * kWsWaves;
  const int64_t wi = (int64_t)blockIdx.x * kWsWaves + wave;
  int64_t span = n_wt;
  int64_t pool = a.sched ? span * a.pool_pct / 100 : 0;
  if (span - pool < 2 * nw) pool = 0;
  span -= pool;
  const int64_t pool_first = span;
  int64_t tile_first = span * wi / nw;
  int64_t tile_last = span * (wi + 1) / nw;
  long long *tsw = nullptr;
  if (a.ts) {
    tsw = a.ts + wi * kWsStampStride;
    if (lane == 0) {
      tsw[0] = __builtin_readcyclecounter();
      tsw[2] = __builtin_amdgcn_s_memrealtime();
      tsw[4] = tile_last - tile_first;
      tsw[5] = NT;
      tsw[6] = 0;
    }
  }
  int stamped = 0;
  for (;;) {
    pool_ws_body<EMIT>(a, wl, bias_lds, tile_first, tile_last, lane, tsw,
                       stamped, n_edges);
    if (pool == 0) break;
    int c = 0;
    if (lane == 0)
      c = __hip_atomic_fetch_add(&a.sched[2], a.chunk, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    c = __builtin_amdgcn_readfirstlane(c);
    if (c >= pool) break;
    tile_first = pool_first + c;
    tile_last = tile_first + a.chunk < n_wt ? tile_first + a.chunk : n_wt;
  }
  if (tsw && lane == 0) {
    tsw[1] = __builtin_readcyclecounter();
    tsw[3] = __builtin_amdgcn_s_memrealtime();
    tsw[7] = stamped;
  }
  if (a.sched && lane == 0) {
    const int total = (int)gridDim.x * kWsWaves;
    const int done = __hip_atomic_fetch_add(&a.sched[1], 1, __ATOMIC_ACQ_REL,
                                            __HIP_MEMORY_SCOPE_AGENT);
    if (done == total - 1) {
      __hip_atomic_store(&a.sched[2], 0, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.sched[1], 0, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace pgnn
