// fp16 two-part form of the weights-stationary pooling kernel (pool_ws.h): the
// car point MLP 4 -> 32 -> 64 -> 128 -> 300 of PointSetPooling (gnn.py:256-277),
//     out[d] = max over edges (s -> d) of MLP([f(s), xyz(s) - xyz(kp(d))]),
// with the LAST layer (128 -> 300: 79 % of the fp32 kernel's matrix time) in
// the arithmetic of edge_ws_f16.h -- both operands as x0 + x1' / 2^11 in fp16,
// three fp16 MFMAs per 32-wide block, two fp32 accumulators; optionally the
// 64 -> 128 layer below it as well (reg_layer_f16: 76 % of what is left); the
// two narrowest layers stay fp32 MFMA in registers (reg_layer).  It belongs to
// the SECONDARY arithmetic `edge_arith = 'f16x2'`: with the edge stage at
// ~390 us the fp32 pooling kernel (~300 us) had become a fifth of the frame.
//
// No layout change between the layers: the fp32 accumulators of reg_layer
// hold, in lane (g, n), features 16 q + 4 g .. + 3 of row n -- eight values per
// 32-wide block (q = 2 kb, 2 kb + 1), which is what one lane contributes to the
// B operand of v_mfma_f32_16x16x32_f16; only WHICH eight differs from the
// instruction's natural order (8 g .. 8 g + 7), and the order of the summation
// index is free: the image pgnn_pack_fc_f16x2_acc writes has the weights' rows
// in that order (slot (g, j) of block kb <-> feature 32 kb + 4 g + j for j < 4,
// 32 kb + 16 + 4 g + j - 4 otherwise).  4 bytes per weight: all 19 column tiles
// of a workgroup stay in LDS (152 KiB, the fp32 kernel's footprint), so --
// unlike the edge kernel's three column groups -- every row is gathered, run
// through the hidden layers and split ONCE.
//
// Range: the hidden activations are clamped at 65504 and flagged from 32768 on
// like the edge kernel's (`status` bit 0).
#pragma once
#include "edge_ws_f16.h"
#include "pool_ws.h"

namespace pgnn {

// the [16 x 32 KB] hidden rows of a tile (fp32 accumulators of the layer
// below, >= 0) -> two packed fp16 parts per 32-wide block
template <int KB>
__device__ __forceinline__ void pool_split_f16(const v4f (&h)[2 * KB],
                                               v4u (&X0)[KB], v4u (&X1)[KB],
                                               u32 &gmax) {
  const v4f zero[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const v4f p[2] = {h[2 * kb], h[2 * kb + 1]};
    split_block_f16(p, zero, X0[kb], X1[kb], gmax);
  }
}

// reg_layer (mlp_engine.h) in the two-part fp16 arithmetic, for the 64 -> 128
// layer (76 % of the hidden layers' matrix time): `img` is the layer's
// pgnn_pack_fc_f16x2_acc image in global memory ([kb][t][part][lane], bias
// behind), read per tile through L1 like reg_layer's fp32 fragments (same
// bytes); column tiles in blocks of 4, one stage of fragments in flight.
template <int KB, int NT>
__device__ __forceinline__ void reg_layer_f16(const void *img, int relu_from,
                                              int lane, const v4f (&in)[2 * KB],
                                              v4f (&out)[NT], u32 &gmax) {
  constexpr int TB = 4;
  static_assert(NT % TB == 0, "column tiles come in blocks");
  int zero;  // opaque: see reg_layer
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  const v4u *__restrict__ wp = reinterpret_cast<const v4u *>(img) + lane + zero;
  const float *bias = reinterpret_cast<const float *>(
                          reinterpret_cast<const v4u *>(img) + KB * NT * 2 * 64) +
                      zero;
  const int g4 = 4 * (lane >> 4) + zero;
  auto frag = [&](int s, int j, int part) -> v4u {  // stage s = (t0 / TB, kb)
    const int kb = s % KB, t = (s / KB) * TB + j;
    return wp[(size_t)((kb * NT + t) * 2 + part) * 64];
  };
  constexpr int NS = (NT / TB) * KB;
  v4u w[2][2][TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) w[0][0][j] = frag(0, j, 0), w[0][1][j] = frag(0, j, 1);
  v4u X0[KB], X1[KB];
  pool_split_f16<KB>(in, X0, X1, gmax);
  v4f alo[TB];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kb = s % KB, t0 = (s / KB) * TB;
    if (s + 1 < NS) {
#pragma unroll
      for (int j = 0; j < TB; ++j)
        w[(s + 1) & 1][0][j] = frag(s + 1, j, 0), w[(s + 1) & 1][1][j] = frag(s + 1, j, 1);
    }
    if (kb == 0) {
#pragma unroll
      for (int j = 0; j < TB; ++j)
        out[t0 + j] = alo[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < TB; ++j)
      out[t0 + j] = mfma_f16(w[s & 1][0][j], X0[kb], out[t0 + j]);
#pragma unroll
    for (int j = 0; j < TB; ++j) alo[j] = mfma_f16(w[s & 1][0][j], X1[kb], alo[j]);
#pragma unroll
    for (int j = 0; j < TB; ++j) alo[j] = mfma_f16(w[s & 1][1][j], X0[kb], alo[j]);
    __builtin_amdgcn_sched_barrier(0);  // one stage of prefetch, no more
    if (kb == KB - 1) {
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int t = t0 + j;
        const v4f b = *reinterpret_cast<const v4f *>(bias + 16 * t + g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(alo[j][r], 1.0f / kF16Scale, out[t][r]) + b[r];
          if (16 * t + g4 + r >= relu_from) v = v > 0.0f ? v : 0.0f;
          out[t][r] = v;
        }
      }
    }
  }
}

// acc[t] = (sum over the 4 blocks of w0 x0) + (w0 x1' + w1' x0) / 2^11 for the
// NTB column tiles from tb on; fragments [kb][t][part][lane] in LDS
template <int NT, int NTB>
__device__ __forceinline__ void pool_ws2_block(const v4u *const (&wfrag)[3],
                                               int tb, const v4u (&X0)[4],
                                               const v4u (&X1)[4],
                                               v4f (&acc)[NTB]) {
  auto frag = [&](int kb, int t, int part) -> v4u {
    const int f = (kb * NT + tb + t) * 2 + part;
    return wfrag[f >> 6][(f & 63) * 64];
  };
  v4u w0[NTB], w1[NTB];
  v4f alo[NTB];
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    acc[t] = alo[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    w0[t] = frag(0, t, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    // term-major as in edge_ws2_body: w1' of this block is requested under
    // the first term, w0 of the next block under the last
#pragma unroll
    for (int t = 0; t < NTB; ++t) w1[t] = frag(kb, t, 1);
#pragma unroll
    for (int t = 0; t < NTB; ++t) acc[t] = mfma_f16(w0[t], X0[kb], acc[t]);
#pragma unroll
    for (int t = 0; t < NTB; ++t) alo[t] = mfma_f16(w0[t], X1[kb], alo[t]);
    if (kb + 1 < 4) {
#pragma unroll
      for (int t = 0; t < NTB; ++t) w0[t] = frag(kb + 1, t, 0);
    }
#pragma unroll
    for (int t = 0; t < NTB; ++t) alo[t] = mfma_f16(w1[t], X0[kb], alo[t]);
#pragma unroll
    for (int m = 0; m < 3 * NTB; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 1, 0);
      if (m < NTB || (m >= 2 * NTB && kb + 1 < 4))
        __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[t][r] = __builtin_fmaf(alo[t][r], 1.0f / kF16Scale, acc[t][r]);
}

// tiles [tile_first, tile_last) of 16 edge rows; structure of pool_ws_body.
// L2F16: the 64 -> 128 layer in the two-part arithmetic as well (a.l2_f16)
template <bool L2F16>
__device__ __forceinline__ void pool_ws2_body(const PoolWsArgs &a,
                                              const v4u *__restrict__ wl,
                                              const float *bias_lds,
                                              int64_t tile_first,
                                              int64_t tile_last, int lane,
                                              const int64_t E, u32 &gmax) {
  constexpr int NT = 19;
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
  // index chain one tile ahead (pool_ws_body)
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
    int lz;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    const v4u *const wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    unsigned starts = 1u;  // virtual tile: "row 0 opens a run"
    int my_d = -1;
    v4u X0[4], X1[4];
    if (fin) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) X0[kb] = X1[kb] = (v4u){0u, 0u, 0u, 0u};
    } else {
      const bool ok = nxt_ok;
      const int my_s = ok ? nxt.x : 0;
      my_d = ok ? nxt.y : -1;
      const int my_k = nxt_k;
      nxt_ok = tile + 1 < tile_last && e0 + 16 + n < E;
      nxt = e2[nxt_ok ? e0 + 16 + n : 0];
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
            const int ax = c - a.nfeat;
            v = a.xyz[3 * (int64_t)my_s + ax] - a.xyz[3 * (int64_t)my_k + ax];
          }
          x[0][i] = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      v4f h1[2], h2[4], h3[8];
      reg_layer<1, 2>(a.l0, lane, x, h1);
      reg_layer<2, 4>(a.l1, lane, h1, h2);
      if constexpr (L2F16)
        reg_layer_f16<2, 8>(a.l2_f16, a.l2.relu_from, lane, h2, h3, gmax);
      else
        reg_layer<4, 8>(a.l2, lane, h2, h3);
      __builtin_amdgcn_sched_barrier(0);
      pool_split_f16<4>(h3, X0, X1, gmax);
      {
        const int dn = nxt_ok ? nxt.y : 0;
        nxt_k = a.kp[((unsigned)dn < (unsigned)a.num_segments) ? dn : 0];
      }
      const int prev = __builtin_amdgcn_update_dpp(run.cur_d, my_d,
                                                   0x111 /*row_shr:1*/, 0xF, 0xF,
                                                   false);
      starts = (unsigned)(__ballot(my_d != prev) & 0xFFFFull);
    }
    const WsRun run0 = run;
#define PGNN_POOL_WS2_BLOCK(NTB, TB, CARRY)                                    \
    {                                                                          \
      v4f acc[NTB];                                                            \
      if (!fin) {                                                              \
        pool_ws2_block<NT, NTB>(wfrag, TB, X0, X1, acc);                       \
      } else {                                                                 \
        _Pragma("unroll")                                                      \
        for (int t = 0; t < NTB; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};      \
      }                                                                        \
      run = run0;                                                              \
      ws_epilogue<NTB>(a, bias_lds + 16 * (TB), TB, lane, acc, CARRY, starts,  \
                       my_d, run, fin, d_after, inf);                          \
    }
    PGNN_POOL_WS2_BLOCK(7, 0, carry0)
    PGNN_POOL_WS2_BLOCK(6, 7, carry1)
    PGNN_POOL_WS2_BLOCK(6, 13, carry2)
#undef PGNN_POOL_WS2_BLOCK
    if (fin) break;
  }
}

// a.wp: the image pgnn_pack_fc_f16x2_acc wrote for the last layer (4 blocks x
// 19 tiles x 2 parts, bias behind); static ranges + tile pool as pool_ws_kernel
template <bool L2F16>
__global__ __launch_bounds__(64 * kWsWaves) void pool_ws_f16x2_kernel(
    PoolWsArgs a, int32_t *status) {
  constexpr int KB = 4, NT = 19;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4u *wl = reinterpret_cast<v4u *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KB * NT * 2 * 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const v4u *__restrict__ src = reinterpret_cast<const v4u *>(a.wp);
    constexpr int PER = (KB * NT * 2 + kWsWaves - 1) / kWsWaves;
    v4u tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      tmp[i] = src[(size_t)(f < KB * NT * 2 ? f : 0) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      if (f < KB * NT * 2) wl[(size_t)f * 64 + lane] = tmp[i];
    }
    if ((int)threadIdx.x < 16 * NT)
      bias_lds[threadIdx.x] = a.wp[(size_t)KB * NT * 2 * 256 + threadIdx.x];
  }
  __syncthreads();
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
  u32 gmax = 0;
  for (;;) {
    pool_ws2_body<L2F16>(a, wl, bias_lds, tile_first, tile_last, lane, n_edges,
                         gmax);
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
  if (status && ((gmax & 0xffffu) >= 0x7800u || (gmax >> 16) >= 0x7800u))
    atomicOr(status, 1);
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
