// Split-bf16 form of the weights-stationary edge kernel (edge_ws.h): the same
// fused stage
//     out[d] = max over edges (s -> d) of ReLU( ReLU(P[s] - Q[d]) W + b ),
// with the 300x300 product evaluated on the bf16 matrix pipe instead of the
// fp32 one.  SECONDARY path (bench.py `secondary_bf16x3`, gnn.EDGE_ARITH): the
// fp32-MFMA kernel stays the default and the parity reference.
//
// Why.  v_mfma_f32_16x16x4_f32 runs at the VALU's rate (157 TFLOP/s, and
// nothing issues beside it: edge_ws.h); the bf16 matrix core is 16x faster
// and separate from the VALU.  An fp32 value is EXACTLY the sum of three
// bf16 values (8 + 8 + 8 significand bits): x = x0 + x1 + x2 with
// x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), both differences
// exact in fp32.  With both operands split,
//     x w = sum_{i+j<=2} x_i w_j  +  (x1 w2 + x2 w1 + x2 w2),
// and the three dropped terms are below 2^-26 |x w| -- a quarter of the
// rounding error one fp32 FMA commits (2^-24).  Every kept product of two
// bf16 values is exact in fp32, so six bf16 MFMAs accumulating in fp32
// reproduce the fp32 product to BELOW fp32 rounding; what remains is the
// accumulation's own fp32 rounding, of which this form has fewer steps (one
// per 32-wide block and term instead of one per element).  Measured on the
// bench frames: max |logit - float64 oracle| no larger than the fp32-MFMA
// kernel's (tests/test_gpu_bf16x3.py prints both).
//
// Cost.  6 x v_mfma_f32_16x16x32_bf16 (16 cycles each) per 32 x 16 x 16 block
// = 96 cycles against 8 x 32 = 256 for fp32: 2.67x fewer matrix cycles.  The
// price is VALU work per gathered element -- 3 v_cvt_pk_bf16_f32, 2 unpack
// pairs, 2 packed subtractions per float pair (~4.5 instructions per element
// on top of the subtraction and ReLU) -- which, unlike next to fp32 MFMAs,
// can issue beside the partner wave's matrix instructions; and 1.5x the LDS
// for the weights (three bf16 images = 6 bytes per weight): 19 column tiles
// in FOUR groups of 5/5/5/4 (150 KiB) instead of three, i.e. the rows are
// gathered and split four times.
//
// Layouts.  v_mfma_f32_16x16x32_bf16, transposed product out^T = W^T h^T:
//   A (weights)      lane (g, i): W[32 kb + 8 g + j][16 t + i], j = 0..7
//   B (activations)  lane (g, n): h[row n][32 kb + 8 g + j],    j = 0..7
//   C / D            lane (g, n), register r <-> feature 16 t + 4 g + r of row n
// (C / D as in edge_ws.h, so the segmented-max epilogue is shared).  A lane's
// eight bf16 are four u32, element 2 m in the low half.  The image
// pgnn_pack_fc_bf16x3 writes is [kb][t][part][lane][4 u32] (1 KiB fragments),
// the layer's bias (fp32, 16 nt values) behind it.  Rows of P / Q are 304
// floats: the last 32-block of C = 300 covers features 288..319, the lanes
// with g >= 2 re-read the row's last 16 bytes (finite values) against zero
// weights.
#pragma once
#include "edge_ws.h"

namespace pgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

// (lo, hi) -> packed bf16 pair, round to nearest even
__device__ __forceinline__ u32 cvt_pk_bf16(float lo, float hi) {
  u32 r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// two floats -> their three bf16 parts, packed pairwise
__device__ __forceinline__ void split3(float a, float b, u32 &p0, u32 &p1,
                                       u32 &p2) {
  p0 = cvt_pk_bf16(a, b);
  const v2f r1 = pk_sub((v2f){a, b}, (v2f){__uint_as_float(p0 << 16),
                                           __uint_as_float(p0 & 0xffff0000u)});
  p1 = cvt_pk_bf16(r1[0], r1[1]);
  const v2f r2 = pk_sub(r1, (v2f){__uint_as_float(p1 << 16),
                                  __uint_as_float(p1 & 0xffff0000u)});
  p2 = cvt_pk_bf16(r2[0], r2[1]);
}

__device__ __forceinline__ v4f mfma_bf16(v4u a, v4u b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
      __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// tiles [tile_first, tile_last) of 16 edge rows, column tiles t0 .. t0+NTG-1
// whose fragments sit in `wl` ([KB][NTG][3][64] v4u)
template <int KB, int NTG>
__device__ __forceinline__ void edge_ws3_body(const EdgeWsArgs &a,
                                              const v4u *__restrict__ wl, int t0,
                                              const float *bias_lds,
                                              int64_t tile_first,
                                              int64_t tile_last, int lane,
                                              const int64_t E) {
  static_assert(KB % 2 == 0, "the gather runs in two halves");
  if (tile_first >= tile_last) return;
  const int n = lane & 15;
  const int64_t e_first = tile_first * 16;
  const int64_t e_end = tile_last * 16 < E ? tile_last * 16 : E;
  const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(a.P);
  const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(a.Q);
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);

  // the open run and the index prefetch: as in edge_ws_body
  int cur_d = e_first > 0 ? a.edges[2 * (e_first - 1) + 1] : -1;
  int d_after = e_end < E ? a.edges[2 * e_end + 1] : -1;
  cur_d = __builtin_amdgcn_readfirstlane(cur_d);
  d_after = __builtin_amdgcn_readfirstlane(d_after);
  bool cur_left_closed = false, cur_has = false;
  v4f carry[NTG];
#pragma unroll
  for (int t = 0; t < NTG; ++t)
    carry[t] = (v4f){kFloatLowest, kFloatLowest, kFloatLowest, kFloatLowest};
  const float inf = opaque_inf();
  bool nxt_ok = e_first + n < E;
  int2 nxt = e2[nxt_ok ? e_first + n : 0];
  for (int64_t tile = tile_first;; ++tile) {
    const bool fin = tile >= tile_last;
    const int64_t e0 = tile * 16;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int lz;  // opaque per-tile lane id: see edge_ws_body
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    const v4u *__restrict__ wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    v4f acc[NTG];
    unsigned starts = 1u;
    int my_d = -1;
    if (fin) {
#pragma unroll
      for (int t = 0; t < NTG; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    } else {
      const int my_s = nxt_ok ? nxt.x : 0;
      my_d = nxt_ok ? nxt.y : -1;
      nxt_ok = tile + 1 < tile_last && e0 + 16 + n < E;
      nxt = e2[nxt_ok ? e0 + 16 + n : 0];
      // ---- gather + split: B operands of all K blocks, three parts each ----
      v4u x0[KB], x1[KB], x2[KB];
      {
        const int dq = ((unsigned)my_d < (unsigned)a.num_segments) ? my_d : 0;
        const v4f *__restrict__ pr = P4 + (int64_t)my_s * a.ldv4;
        const v4f *__restrict__ qr = Q4 + (int64_t)dq * a.ldv4;
        const int last = a.ldv4 - 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          v4f pv[KB / 2][2], qv[KB / 2][2];
#pragma unroll
          for (int k = 0; k < KB / 2; ++k) {
            const int kb = half * (KB / 2) + k;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              int idx = 8 * kb + 2 * g + i;
              idx = idx < last ? idx : last;  // tail block: see the header
#ifdef PGNN_B16_ABL_NOGATHER  // timing ablation (wrong results): one address
              idx = i;
#endif
              pv[k][i] = pr[idx];
              qv[k][i] = qr[idx];
            }
          }
          // all loads of the half in flight before the first use
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < KB / 2; ++k) {
            const int kb = half * (KB / 2) + k;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const v2f lo = pk_sub((v2f){pv[k][i][0], pv[k][i][1]},
                                    (v2f){qv[k][i][0], qv[k][i][1]});
              const v2f hi = pk_sub((v2f){pv[k][i][2], pv[k][i][3]},
                                    (v2f){qv[k][i][2], qv[k][i][3]});
              const float h0 = max_nc(lo[0], 0.0f, inf);
              const float h1 = max_nc(lo[1], 0.0f, inf);
              const float h2 = max_nc(hi[0], 0.0f, inf);
              const float h3 = max_nc(hi[1], 0.0f, inf);
              u32 a0, a1, a2, b0, b1, b2;
#ifdef PGNN_B16_ABL_NOSPLIT  // timing ablation (wrong results): one part only
              a0 = a1 = a2 = cvt_pk_bf16(h0, h1);
              b0 = b1 = b2 = cvt_pk_bf16(h2, h3);
#else
              split3(h0, h1, a0, a1, a2);
              split3(h2, h3, b0, b1, b2);
#endif
              x0[kb][2 * i] = a0;
              x1[kb][2 * i] = a1;
              x2[kb][2 * i] = a2;
              x0[kb][2 * i + 1] = b0;
              x1[kb][2 * i + 1] = b1;
              x2[kb][2 * i + 1] = b2;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- out^T = W^T h^T, six bf16 products per block, small terms first ----
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int t = 0; t < NTG; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
      // The six terms of a block run TERM by term over the NTG column tiles
      // (consecutive MFMAs hit different accumulators: no dependent-issue
      // stall), in an order that frees a weight part's registers as early as
      // possible -- (w2 x0) | (w1 x1) (w1 x0) | (w0 x2) (w0 x1) (w0 x0) -- and
      // each part's fragments of the NEXT block are requested the moment the
      // part is done: 25 / 20 / 15 MFMAs (400 / 320 / 240 cycles) before their
      // first use.  (The first version walked tile by tile, six dependent
      // MFMAs in a row behind an lgkmcnt(0): 664 us; small-terms-first is kept
      // approximately.)
      auto frag = [&](int kb, int t, int part) -> v4u {
        const int f = (kb * NTG + t) * 3 + part;
        return wfrag[f >> 6][(f & 63) * 64];
      };
      v4u w0[NTG], w1[NTG], w2[NTG];
#pragma unroll
      for (int t = 0; t < NTG; ++t) w2[t] = frag(0, t, 2);
#pragma unroll
      for (int t = 0; t < NTG; ++t) w1[t] = frag(0, t, 1);
#pragma unroll
      for (int t = 0; t < NTG; ++t) w0[t] = frag(0, t, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PGNN_B16_ABL_NOMFMA  // timing ablation (wrong results): one block
#pragma unroll
      for (int t = 0; t < NTG; ++t)
        for (int kb = 1; kb < KB; ++kb) x0[0] ^= x1[kb] ^ x2[kb] ^ x0[kb];
#pragma unroll
      for (int kb = 0; kb < 1; ++kb) {
#else
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#endif
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w2[t], x0[kb], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < KB) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w2[t] = frag(kb + 1, t, 2);
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], x1[kb], acc[t]);
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], x0[kb], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < KB) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w1[t] = frag(kb + 1, t, 1);
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x2[kb], acc[t]);
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x1[kb], acc[t]);
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x0[kb], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < KB) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w0[t] = frag(kb + 1, t, 0);
        }
      }
      // ---- segmented max over the 16 rows: as in edge_ws_body ---------------
      if (a.prio) __builtin_amdgcn_s_setprio(3);
      const int up = __shfl_up(my_d, 1);
      const int prev = n == 0 ? cur_d : up;
      starts = (unsigned)(__ballot(my_d != prev) & 0xFFFFull);
    }  // !fin
    WsRun st = {cur_d, cur_left_closed, cur_has};
    ws_epilogue<NTG>(a, bias_lds, t0, lane, acc, carry, starts, my_d, st, fin,
                     d_after, inf);
    cur_d = st.cur_d;
    cur_left_closed = st.left_closed;
    cur_has = st.has;
    if (fin) break;
  }
  __builtin_amdgcn_s_setprio(0);
}

// a.wp: the bf16x3 image (pgnn_pack_fc_bf16x3) of the layer; static partition
// of the 16-row tiles as in edge_ws_kernel (no tile pool)
template <int KB, int NTMAX>
__global__ __launch_bounds__(64 * kWsWaves) void edge_ws_bf16x3_kernel(EdgeWsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4u *wl = reinterpret_cast<v4u *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KB * NTMAX * 3 * 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = blockIdx.x % a.xcds;
  const int local = blockIdx.x / a.xcds;
  int grp = 0;
  while (grp + 1 < a.groups && local >= a.wg0[grp + 1]) ++grp;
  const int t0 = a.tile0[grp];
  const int ntg = a.tile0[grp + 1] - t0;
  {
    // fragments (kb, t, part) of this group -> LDS [kb][t][part][lane]; all of
    // a wave's requests in flight before its first LDS write (edge_ws_kernel)
    const v4u *__restrict__ src = reinterpret_cast<const v4u *>(a.wp);
    constexpr int PER = (KB * NTMAX * 3 + kWsWaves - 1) / kWsWaves;
    const int n_frag = KB * ntg * 3;
    v4u tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      const int fc = f < n_frag ? f : 0;
      const int kb = fc / (ntg * 3), r = fc - kb * ntg * 3;  // r = t * 3 + part
      tmp[i] = src[((size_t)(kb * a.nt + t0) * 3 + r) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      if (f < n_frag) wl[(size_t)f * 64 + lane] = tmp[i];
    }
    if ((int)threadIdx.x < 16 * ntg)
      bias_lds[threadIdx.x] =
          a.wp[(size_t)KB * a.nt * 3 * 256 + 16 * t0 + threadIdx.x];
  }
  __syncthreads();
  int64_t n_edges = a.n_edges;
  if (a.n_dev) {
    const int64_t nd = *a.n_dev;
    n_edges = nd < n_edges ? nd : n_edges;
  }
  const int64_t n_wt = (n_edges + 15) / 16;
  const int64_t s_first = n_wt * slice / a.xcds;
  const int64_t s_last = n_wt * (slice + 1) / a.xcds;
  const int64_t nw = (int64_t)(a.wg0[grp + 1] - a.wg0[grp]) * kWsWaves;
  const int64_t wi = (int64_t)(local - a.wg0[grp]) * kWsWaves + wave;
  const int64_t span = s_last - s_first;
  const int64_t tile_first = s_first + span * wi / nw;
  const int64_t tile_last = s_first + span * (wi + 1) / nw;
  if (ntg == NTMAX)
    edge_ws3_body<KB, NTMAX>(a, wl, t0, bias_lds, tile_first, tile_last, lane,
                             n_edges);
  else
    edge_ws3_body<KB, NTMAX - 1>(a, wl, t0, bias_lds, tile_first, tile_last,
                                 lane, n_edges);
}

}  // namespace pgnn
