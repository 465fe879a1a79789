// Weights-stationary form of the fused GraphNetAutoCenter edge stage
// (gnn.py:338-365 with the first edge layer factored per vertex):
//     out[d] = max over edges (s -> d) of ReLU( ReLU(P[s] - Q[d]) W + b ).
//
// The LDS-tile kernel (gnn.hip, fused_mlp_kernel<., PRO_EDGE>) keeps a 64-row
// activation tile in LDS and streams every weight fragment from L2 once per
// tile: 9-12 operand loads per 76 MFMAs, a drained vmcnt at every K-loop
// header, two workgroup barriers per tile.  On CDNA4 the 160 KB LDS is large
// enough to turn the roles round:
//
//   * a workgroup (8 waves = 2 per SIMD, the whole CU) owns a GROUP of <= 8 of
//     the layer's column tiles and keeps their weight fragments resident in LDS
//     for the life of the kernel (C = 300: 19 column tiles in groups of 7/6/6,
//     19 x 7 KiB = 133 KiB; C = 256: 16 tiles in groups of 8/8, 128 KiB);
//   * every wave is autonomous: it owns a contiguous range of 16-row tiles of
//     the edge list, gathers ReLU(P[src] - Q[dst]) of its 16 rows straight into
//     registers in the MFMA B-operand layout (lane (g, n): row n, features
//     16q + 4g .. 16q + 4g + 3 -- one dwordx4 per K-group q per operand), and
//     computes the TRANSPOSED product out^T = W^T h^T: the weight fragment
//     (one ds_read_b128 per lane, in the host-packed order of pgnn_pack_fc) is
//     the A operand, the activations the B operand (mlp_engine.h, reg_layer);
//   * the accumulators hold out^T: lane (g, n), register r <-> output feature
//     16t + 4g + r of row n.  The running max of an open segment is kept PER
//     LANE (v_max over the accumulators, no cross-lane step) and only reduced
//     over the 16 rows when the segment closes -- with a mean fan-in of ~200
//     most 16-row tiles lie inside one segment, so the scatter-max of a tile
//     is 4 * tiles VALU instructions;
//   * no barrier after the weights are in LDS, no LDS traffic for activations,
//     no weight traffic from L2 at all; the price is that the rows are gathered
//     once per column group (3x / 2x the gather bytes, all L2 / MALL hits:
//     P and Q are K x 304 floats).
//
// Every output element sees the same sequence of MFMA updates (K-groups
// ascending, k-steps ascending, the same four products per step) as in the
// LDS-tile kernel, so the two kernels agree bit for bit (tested; `mlp_debug`
// bit 2048 selects the LDS-tile kernel).
//
// Work distribution.  Workgroup b lands on XCD b % 8 (round-robin dispatch),
// so XCD x takes the x-th eighth of the 16-row tiles and its workgroups divide
// that slice among the column groups in proportion to the groups' tile
// counts: the gathers of one slice (all column groups) meet in one L2.
#pragma once
#include "mlp_engine.h"

namespace pgnn {

constexpr int kWsMaxGroups = 4;
constexpr int kWsWaves = 8;
constexpr int kWsStampTiles = 38;
constexpr int kWsStampStride = 8 + 4 * kWsStampTiles;  // int64 per wave

struct EdgeWsArgs {
  const float *P, *Q;   // [num_segments, 4 * ldv4]
  int ldv4;
  const int32_t *edges;  // [n_edges, 2] rows (src, dst)
  int64_t n_edges;
  const float *wp;  // packed weights (pgnn_pack_fc), bias follows
  int nt;           // column tiles of the layer (= its K groups)
  int relu_from;
  float *out;
  int64_t ldo;
  int num_segments;
  int sorted;
  int xcds;                          // row slices (workgroup b -> slice b % xcds)
  int prio;                          // static s_setprio 1 for waves 4..7
  long long *ts;                     // profiling stamps (tools/ws_timeline.py) or null
  int groups;                        // column groups
  int tile0[kWsMaxGroups + 1];       // group g owns column tiles [tile0[g], tile0[g+1])
  int wg0[kWsMaxGroups + 1];         // ... and local workgroups [wg0[g], wg0[g+1]) of a slice
};

template <int NTG>
__device__ __forceinline__ void ws_flush(const EdgeWsArgs &a, int t0, int lane,
                                         int d, const v4f (&v)[NTG],
                                         bool whole) {
  if (d < 0 || d >= a.num_segments) return;  // wave-uniform
  // An opaque zero keeps the column indices, bias addresses and ReLU
  // predicates of this (rare) path from being hoisted out of the caller's tile
  // loop, where they would sit in ~60 registers across the MFMA phase.
  int zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  const int g = (lane >> 4) + zero;
  const float *bias = a.wp + (size_t)a.nt * a.nt * 256;
  float *orow = a.out + (int64_t)d * a.ldo;
#pragma unroll
  for (int t = 0; t < NTG; ++t) {
    const int col = 16 * (t0 + t) + 4 * g;
    const v4f b = *reinterpret_cast<const v4f *>(bias + col);
    v4f x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = v[t][r];
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 2));
      m = fmaxf(m, __shfl_xor(m, 4));
      m = fmaxf(m, __shfl_xor(m, 8));
      // max_r act(a_r + b) == act(max_r a_r + b): +b and ReLU are monotone
      m += b[r];
      if (col + r >= a.relu_from) m = m > 0.0f ? m : 0.0f;
      x[r] = m;
    }
    if ((lane & 15) == 0) {
      if (whole) {
        *reinterpret_cast<v4f *>(orow + col) = x;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomic_max_f32(orow + col + r, x[r] + 0.0f);
      }
    }
  }
}

// tiles [tile_first, tile_last) of 16 edge rows, column tiles t0 .. t0+NTG-1
// whose fragments sit in `wl` ([KQ][NTG][64] float4)
template <int KQ, int NTG>
__device__ __forceinline__ void edge_ws_body(const EdgeWsArgs &a,
                                             const v4f *__restrict__ wl, int t0,
                                             int64_t tile_first,
                                             int64_t tile_last, int lane,
                                             long long *tsw) {
  if (tile_first >= tile_last) return;
  const int g = lane >> 4, n = lane & 15;
  const int64_t E = a.n_edges;
  const int64_t e_first = tile_first * 16;
  const int64_t e_end = tile_last * 16 < E ? tile_last * 16 : E;
  const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(a.P);
  const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(a.Q);
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);
  const v4f *__restrict__ wlane = wl + lane;

  // the open run: the run of equal dst that contains the previous edge.  At
  // the start of the range that is the run of the edge BEFORE the range
  // (nothing accumulated yet, not left-closed: it began in another wave's
  // range).
  int cur_d = e_first > 0 ? a.edges[2 * (e_first - 1) + 1] : -1;
  int d_after = e_end < E ? a.edges[2 * e_end + 1] : -1;
  cur_d = __builtin_amdgcn_readfirstlane(cur_d);
  d_after = __builtin_amdgcn_readfirstlane(d_after);
  bool cur_left_closed = false, cur_has = false;
  v4f carry[NTG];
#pragma unroll
  for (int t = 0; t < NTG; ++t)
    carry[t] = (v4f){kFloatLowest, kFloatLowest, kFloatLowest, kFloatLowest};

  int nxt_s = 0, nxt_d = -1;
  if (e_first + n < E) {
    const int2 sd = e2[e_first + n];
    nxt_s = sd.x;
    nxt_d = sd.y;
  }
  for (int64_t tile = tile_first; tile < tile_last; ++tile) {
    const int64_t e0 = tile * 16;
    // profiling builds only: s_memtime at the four phase boundaries of the
    // first 38 tiles of every wave
    long long *tst = nullptr;
    if (tsw && tile - tile_first < kWsStampTiles)
      tst = tsw + 8 + 4 * (tile - tile_first);
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[0] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    const int my_s = nxt_s, my_d = nxt_d;
    nxt_s = 0;
    nxt_d = -1;
    if (tile + 1 < tile_last && e0 + 16 + n < E) {
      const int2 sd = e2[e0 + 16 + n];
      nxt_s = sd.x;
      nxt_d = sd.y;
    }
    // ---- gather: B operands of all K groups ------------------------------
    // rows past the end / foreign ids gather row 0 (finite values in rows the
    // epilogue never reads)
    v4f in[KQ];
    {
      const int dq = ((unsigned)my_d < (unsigned)a.num_segments) ? my_d : 0;
      const v4f *__restrict__ pr = P4 + (int64_t)my_s * a.ldv4 + g;
      const v4f *__restrict__ qr = Q4 + (int64_t)dq * a.ldv4 + g;
      v4f pq[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        in[q] = pr[4 * q];
        pq[q] = qr[4 * q];
      }
      // all 2*KQ loads in flight before the first use
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) in[q][i] = fmaxf(in[q][i] - pq[q][i], 0.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tst) {
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[1] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- out^T = W^T h^T --------------------------------------------------
    v4f acc[NTG];
    {
      v4f w[2][NTG];
#pragma unroll
      for (int t = 0; t < NTG; ++t) {
        acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
        w[0][t] = wlane[t * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        if (q + 1 < KQ) {
#pragma unroll
          for (int t = 0; t < NTG; ++t)
            w[(q + 1) & 1][t] = wlane[((q + 1) * NTG + t) * 64];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NTG; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                w[q & 1][t][s], in[q][s], acc[t], 0, 0, 0);
        // the next K-group's fragments are requested BEFORE this group's
        // MFMAs (left alone, hipcc sinks the ds_reads below them and the next
        // group starts on a cold lgkmcnt); one K-group of prefetch, no more
        if (q + 1 < KQ)
          __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, NTG, 0);
        __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 4 * NTG, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (tst) {
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[2] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- segmented max over the 16 rows ------------------------------------
    // bit r of `starts`: row r does not continue the run of the edge before it
    const int up = __shfl_up(my_d, 1);
    const int prev = n == 0 ? cur_d : up;
    const unsigned starts = (unsigned)(__ballot(my_d != prev) & 0xFFFFull);
    int pos = 0;
    if (!(starts & 1u)) {
      // rows [0, f) continue the open run
      const int f = starts ? __builtin_ctz(starts) : 16;
      if (f == 16) {
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            carry[t][r] = fmaxf(carry[t][r], acc[t][r]);
      } else {
        const bool in_run = n < f;
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            carry[t][r] =
                fmaxf(carry[t][r], in_run ? acc[t][r] : kFloatLowest);
      }
      cur_has = true;
      pos = f;
    }
    while (pos < 16) {  // wave-uniform; `pos` opens a run
      // the open run ends in front of row `pos`: the next edge has another dst
      if (cur_has)
        ws_flush<NTG>(a, t0, lane, cur_d, carry, a.sorted && cur_left_closed);
      const unsigned rest = starts & ~((2u << pos) - 1u);
      const int nextpos = rest ? __builtin_ctz(rest) : 16;
      const bool in_run = n >= pos && n < nextpos;
#pragma unroll
      for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          carry[t][r] = in_run ? acc[t][r] : kFloatLowest;
      cur_d = __builtin_amdgcn_readlane(my_d, pos);
      cur_left_closed = true;
      cur_has = true;
      pos = nextpos;
    }
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[3] = c;
    }
  }
  // the run left open at the end of the range
  if (cur_has)
    ws_flush<NTG>(a, t0, lane, cur_d, carry,
                  a.sorted && cur_left_closed && d_after != cur_d);
}

template <int KQ, int NTMAX>
__global__ __launch_bounds__(64 * kWsWaves) void edge_ws_kernel(EdgeWsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4f *wl = reinterpret_cast<v4f *>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = blockIdx.x % a.xcds;
  const int local = blockIdx.x / a.xcds;
  int grp = 0;
  while (grp + 1 < a.groups && local >= a.wg0[grp + 1]) ++grp;
  const int t0 = a.tile0[grp];
  const int ntg = a.tile0[grp + 1] - t0;
  // this group's weight fragments -> LDS, [q][t][lane] float4 (1 KiB each)
  {
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(a.wp);
    for (int f = wave; f < KQ * ntg; f += kWsWaves) {
      const int q = f / ntg, t = f - q * ntg;
      wl[(size_t)f * 64 + lane] = src[((size_t)q * a.nt + t0 + t) * 64 + lane];
    }
  }
  __syncthreads();
  // the second-dispatched half of an 8-wave workgroup loses every arbitration
  // against the older half (MI355X_MICROARCH.md, two waves per SIMD)
  if (a.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // 16-row tiles of this slice, divided among the group's waves
  const int64_t n_wt = (a.n_edges + 15) / 16;
  const int64_t s_first = n_wt * slice / a.xcds;
  const int64_t s_last = n_wt * (slice + 1) / a.xcds;
  const int64_t nw = (int64_t)(a.wg0[grp + 1] - a.wg0[grp]) * kWsWaves;
  const int64_t wi = (int64_t)(local - a.wg0[grp]) * kWsWaves + wave;
  const int64_t span = s_last - s_first;
  const int64_t tile_first = s_first + span * wi / nw;
  const int64_t tile_last = s_first + span * (wi + 1) / nw;
  long long *tsw = nullptr;
  if (a.ts) {
    tsw = a.ts + ((int64_t)blockIdx.x * kWsWaves + wave) * kWsStampStride;
    if (lane == 0) {
      tsw[0] = __builtin_readcyclecounter();
      tsw[2] = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
      tsw[4] = tile_last - tile_first;
      tsw[5] = ntg;
      tsw[6] = slice;
    }
  }
  if (ntg == NTMAX)
    edge_ws_body<KQ, NTMAX>(a, wl, t0, tile_first, tile_last, lane, tsw);
  else if (ntg == NTMAX - 1)
    edge_ws_body<KQ, NTMAX - 1>(a, wl, t0, tile_first, tile_last, lane, tsw);
  if (tsw && lane == 0) {
    tsw[1] = __builtin_readcyclecounter();
    tsw[3] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace pgnn
