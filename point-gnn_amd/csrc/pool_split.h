// Two-launch form of the fused PointSetPooling stage (gnn.py:256-277) for
// point MLPs whose last layer is too wide to stay in one CU's LDS -- ped_cyl's
// 4 -> 32 -> 64 -> 128 -> 256 -> 512:
//     out[d] = max over edges (s -> d) of MLP([f(s), xyz(s) - xyz(kp(d))]).
//
// The LDS-tile kernel (gnn.hip, fused_mlp_kernel<2, PRO_POOL>) streams every
// weight fragment of every layer from L2 once per 32-row tile and synchronises
// the workgroup twice per layer pass: 0.67 of the fp32-MFMA peak is its
// ceiling (profiles/r05_ped_pool_sweep.txt).  The last layer is 75 % of the
// chain's MFMAs and has the shape of the GNN's edge stage -- rows in, a wide
// product, a segmented max --, so it takes that stage's kernel:
//
//   1. `pool_hidden_kernel` (here): pool_ws.h's wave-autonomous form (12 waves
//      per workgroup) up to the 256-wide hidden layer.  The three narrow layers run in registers
//      (reg_layer), the 128 -> 256 layer's 8 x 16 fragments (128 KiB) stay in
//      LDS for the life of the kernel, and the activated rows go to a
//      workspace in HBM, [n_edges, 256] (one 16-byte store per column tile and
//      lane: 1 KiB per edge, 0.87 GB for ped_dense -- 0.11 ms of HBM write
//      under 0.6 ms of MFMA work);
//   2. `edge_ws_kernel<16, 8, false, ROWS>` (edge_ws.h): the 256 -> 512 layer's
//      32 column tiles in four groups of 8 (16 x 8 fragments = 128 KiB of LDS
//      each), every wave reads its rows as MFMA B operands (one dwordx4 per K
//      group and lane), keeps the running max of the open segment per lane.
//      The four column groups of an XCD walk the same slice of the rows at the
//      same pace, so a row comes from HBM once and from that XCD's L2 three
//      times.
//
// Every output element sees the same sequence of MFMA updates (K groups
// ascending, k-steps ascending) as in the LDS-tile kernel and the hidden rows
// pass through memory unchanged: bit-identical maxima (tested; `mlp_debug` bit
// 8192 selects the LDS-tile kernel).
#pragma once
#include "pool_ws.h"

namespace pgnn {

// tiles [tile_first, tile_last) of 16 edge rows; PoolWsArgs as for
// pool_ws_kernel with `wp` = the 128 -> 256 layer and `a4_out` / `ld4` = the
// hidden rows' workspace (`out`, the run state and a1..a3 are not used)
__device__ __forceinline__ void pool_hidden_body(const PoolWsArgs &a,
                                                 const v4f *__restrict__ wl,
                                                 const float *bias_lds,
                                                 int64_t tile_first,
                                                 int64_t tile_last, int lane,
                                                 const int64_t E) {
  constexpr int KQ = 8, NT = 16, NTB = 8;
  if (tile_first >= tile_last) return;
  const int n = lane & 15;
  const int64_t e_first = tile_first * 16;
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);
  // (point, keypoint) of a tile's rows one tile ahead, the keypoint's point
  // index half a tile ahead (pool_ws.h)
  bool nxt_ok = e_first + n < E;
  int2 nxt = e2[nxt_ok ? e_first + n : 0];
  int nxt_k;
  {
    const int d0 = nxt_ok ? nxt.y : 0;
    nxt_k = a.kp[((unsigned)d0 < (unsigned)a.num_segments) ? d0 : 0];
  }
  for (int64_t tile = tile_first; tile < tile_last; ++tile) {
    const int64_t e0 = tile * 16;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int lz;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    // (128 fragments: the third base is never indexed)
    const v4f *const wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    const bool ok = nxt_ok;
    const int my_s = ok ? nxt.x : 0;
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
    __builtin_amdgcn_s_setprio(0);
    v4f h1[2], h2[4], h3[KQ];
#ifdef PGNN_POOL_ABL_NO_HIDDEN  // timing ablation (wrong results): no hidden layers
#pragma unroll
    for (int q = 0; q < KQ; ++q) h3[q] = x[0];
#else
    reg_layer<1, 2>(a.l0, lane, x, h1);
    reg_layer<2, 4>(a.l1, lane, h1, h2);
    reg_layer<4, 8>(a.l2, lane, h2, h3);
#endif
    __builtin_amdgcn_sched_barrier(0);
    {
      const int dn = nxt_ok ? nxt.y : 0;
      nxt_k = a.kp[((unsigned)dn < (unsigned)a.num_segments) ? dn : 0];
    }
    // 128 -> 256 in two column blocks of 8 tiles (32 h3 + 32 acc + 64 fragment
    // stages), the activated rows straight to the workspace: lane (g, n) holds
    // features 16 t + 4 g .. + 3 of row n
#pragma unroll
    for (int tb = 0; tb < NT; tb += NTB) {
      v4f acc[NTB];
      pool_ws_block<KQ, NT, NTB>(wfrag, tb, h3, acc);
      if (e0 + n < E) {
        float *dstp = a.a4_out + (e0 + n) * a.ld4 + 16 * tb + 4 * g;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
          const v4f bb =
              *reinterpret_cast<const v4f *>(bias_lds + 16 * (tb + t) + 4 * g);
          v4f y;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float xx = acc[t][r] + bb[r];
            if (16 * (tb + t) + 4 * g + r >= a.relu_from)
              xx = xx > 0.0f ? xx : 0.0f;
            y[r] = xx;
          }
          *reinterpret_cast<v4f *>(dstp + 16 * t) = y;
        }
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

// Waves per workgroup: the body needs ~160 VGPRs (no carried segment state),
// so THREE waves per SIMD fit where pool_ws.h / edge_ws.h run two -- a third
// wave covers more of a tile's dependent gather chain (edge -> point ->
// keypoint -> coordinates) with the others' MFMAs: 2 254 -> 2 215 us for the
// ped_dense stage; 16 waves (128 VGPRs) spill: 2 315 us.
#ifndef PGNN_POOLH_WAVES
#define PGNN_POOLH_WAVES 12
#endif
constexpr int kPoolHWaves = PGNN_POOLH_WAVES;
__global__ __launch_bounds__(64 * kPoolHWaves) void pool_hidden_kernel(PoolWsArgs a) {
  constexpr int KQ = 8, NT = 16;
  constexpr int kWsWaves = kPoolHWaves;  // (shadows the 8-wave constant below)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4f *wl = reinterpret_cast<v4f *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KQ * NT * 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    // all 16 fragment requests of a wave in flight before the first LDS write
    // (edge_ws_kernel)
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(a.wp);
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
  int64_t n_edges = a.n_edges;
  if (a.n_dev) {
    const int64_t nd = *a.n_dev;
    n_edges = nd < n_edges ? nd : n_edges;
  }
  // contiguous ranges of 16-row tiles, in the order of the second launch's
  // slices (workgroup b -> XCD b % 8 takes the b % 8-th eighth of the rows, as
  // edge_ws_kernel does)
  const int64_t n_wt = (n_edges + 15) / 16;
  const int xcds = a.slices;
  const int slice = blockIdx.x % xcds, local = blockIdx.x / xcds;
  const int64_t s_first = n_wt * slice / xcds;
  const int64_t s_last = n_wt * (slice + 1) / xcds;
  const int64_t nw = (int64_t)(gridDim.x / xcds) * kWsWaves;
  const int64_t wi = (int64_t)local * kWsWaves + wave;
  const int64_t span = s_last - s_first;
  pool_hidden_body(a, wl, bias_lds, s_first + span * wi / nw,
                   s_first + span * (wi + 1) / nw, lane, n_edges);
}

}  // namespace pgnn
