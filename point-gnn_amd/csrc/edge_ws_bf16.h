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

// (A variant that formed the residuals with v_dot2_f32_bf16 -- p . (-1, 0) + c:
// unpack and subtraction in one instruction -- was 2 % faster and gave wrong
// values as written; not pursued: session r04_s23.)
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
// whose fragments sit in `wl` ([KB][NTG][3][64] v4u).
//
// Schedule of one tile (its K blocks in two halves A, B of KB / 2):
//     Q_A (L1-hot, same dst for most rows) -> split A -> request P_B
//     -> MFMAs of A -> Q_B -> split B -> request P_A of the NEXT tile
//     -> MFMAs of B -> segmented max
// so the 16 distinct P rows of a half -- the kernel's real memory traffic: the
// first version, which gathered a whole tile and then multiplied, spent 217 of
// its 702 us waiting for them (tools/sessions/r04_s13.sh) -- travel while the
// other half's 150 MFMAs run.  Only one half's parts (60 registers) and one
// half's raw P rows (40) are live at a time.
template <int KB, int NTG>
__device__ __forceinline__ void edge_ws3_body(const EdgeWsArgs &a,
                                              const v4u *__restrict__ wl, int t0,
                                              const float *bias_lds,
                                              int64_t tile_first,
                                              int64_t tile_last, int lane,
                                              const int64_t E) {
  static_assert(KB % 2 == 0, "the K blocks run in two halves");
  constexpr int KH = KB / 2;
  if (tile_first >= tile_last) return;
  const int n = lane & 15;
  const int64_t e_first = tile_first * 16;
  const int64_t e_end = tile_last * 16 < E ? tile_last * 16 : E;
  const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(a.P);
  const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(a.Q);
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);
  const int last = a.ldv4 - 1;

  // the open run: as in edge_ws_body
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

  // raw P rows of half `half` of the tile whose row is `row` (clamped, valid):
  // constant offsets from one pointer per row, the tail block (lanes g >= 2 of
  // it lie behind the row's 4 * ldv4 floats) from its own clamped pointer --
  // an index clamp and a 64-bit address per load were 150 of the 800 VALU
  // instructions per tile, and VALU instructions do not run beside the MFMAs
  // here either (SQ_VALU_MFMA_COEXEC_CYCLES: 7 % of the MFMA-busy cycles)
  auto request_p = [&](int row, int half, int g, int tail_base,
                       v4f (&pv)[KH][2]) {
    const v4f *__restrict__ pr = P4 + (int64_t)row * a.ldv4;
    const v4f *__restrict__ pg = pr + 2 * g;
    const v4f *__restrict__ pt = pr + tail_base;
#pragma unroll
    for (int k = 0; k < KH; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kb = half * KH + k;
#ifdef PGNN_B16_ABL_NOGATHER  // timing ablation (wrong results): one address
        pv[k][i] = pr[i];
#else
        pv[k][i] = kb == KB - 1 ? pt[i] : pg[8 * kb + i];
#endif
      }
  };
  auto tail_of = [&](int g) {
    const int t = 8 * (KB - 1) + 2 * g;
    return t < last - 1 ? t : last - 1;
  };
  // (src, dst) of this tile's rows and of the next tile's, requested a tile
  // ahead; validity is applied where a pair is used (edge_ws_body)
  bool cur_ok = e_first + n < E;
  int2 cur = e2[cur_ok ? e_first + n : 0];
  bool nxt_ok = tile_first + 1 < tile_last && e_first + 16 + n < E;
  int2 nxt = e2[nxt_ok ? e_first + 16 + n : 0];
  v4f pa[KH][2];  // half A of the CURRENT tile, in flight across the loop edge
  request_p(cur_ok ? cur.x : 0, 0, lane >> 4, tail_of(lane >> 4), pa);

  for (int64_t tile = tile_first;; ++tile) {
    const bool fin = tile >= tile_last;
    const int64_t e0 = tile * 16;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    int lz;  // opaque per-tile lane id: see edge_ws_body
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    const int tail_base = tail_of(g);
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    const v4u *__restrict__ wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    auto frag = [&](int kb, int t, int part) -> v4u {
      const int f = (kb * NTG + t) * 3 + part;
      return wfrag[f >> 6][(f & 63) * 64];
    };
    v4f acc[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    unsigned starts = 1u;  // virtual tile: "row 0 opens a run"
    int my_d = -1;
    if (!fin) {
      const int my_s = cur_ok ? cur.x : 0;
      my_d = cur_ok ? cur.y : -1;
      const int dq = ((unsigned)my_d < (unsigned)a.num_segments) ? my_d : 0;
      const v4f *__restrict__ qr = Q4 + (int64_t)dq * a.ldv4;
      v4u x0[KH], x1[KH], x2[KH];
      // h = ReLU(p - q) of one half -> its three bf16 parts
      // (a variant in which only lane 0 of every quad loaded the Q row of a
      // one-segment tile and the subtraction took it through a quad_perm DPP
      // broadcast -- a quarter of the returned bytes -- changed nothing: the
      // vector-memory return path is not what the gather waits for)
      const v4f *__restrict__ qg = qr + 2 * g;
      const v4f *__restrict__ qt = qr + tail_base;
      auto split_half = [&](int half, const v4f (&pv)[KH][2]) {
        v4f qv[KH][2];
#pragma unroll
        for (int k = 0; k < KH; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int kb = half * KH + k;
            // constant offsets from one pointer per row; the tail block (whose
            // lanes g >= 2 lie behind the row) from its own clamped pointer
            qv[k][i] = kb == KB - 1 ? qt[i] : qg[8 * kb + i];
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KH; ++k)
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
            x0[k][2 * i] = a0;
            x1[k][2 * i] = a1;
            x2[k][2 * i] = a2;
            x0[k][2 * i + 1] = b0;
            x1[k][2 * i + 1] = b1;
            x2[k][2 * i + 1] = b2;
          }
      };
      // The six terms of a block run TERM by term over the NTG column tiles
      // (consecutive MFMAs hit different accumulators: no dependent-issue
      // stall), in an order that frees a weight part's registers as early as
      // possible -- (w2 x0) | (w1 x1) (w1 x0) | (w0 x2) (w0 x1) (w0 x0) -- and
      // each part's fragments of the NEXT block are requested the moment the
      // part is done: 25 / 20 / 15 MFMAs before their first use.
      v4u w0[NTG], w1[NTG], w2[NTG];
      auto mma_half = [&](int half) {
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          const int kb = half * KH + k;
#ifdef PGNN_B16_ABL_NOMFMA  // timing ablation (wrong results)
          if (k > 0) continue;
#endif
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w2[t], x0[k], acc[t]);
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < KB) {
#pragma unroll
            for (int t = 0; t < NTG; ++t) w2[t] = frag(kb + 1, t, 2);
          }
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], x1[k], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], x0[k], acc[t]);
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < KB) {
#pragma unroll
            for (int t = 0; t < NTG; ++t) w1[t] = frag(kb + 1, t, 1);
          }
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x2[k], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x1[k], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], x0[k], acc[t]);
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < KB) {
#pragma unroll
            for (int t = 0; t < NTG; ++t) w0[t] = frag(kb + 1, t, 0);
          }
        }
      };
      // ---- half A ------------------------------------------------------------
      split_half(0, pa);
      v4f pb[KH][2];
      request_p(my_s, 1, g, tail_base, pb);  // travels under the MFMAs of half A
#pragma unroll
      for (int t = 0; t < NTG; ++t) w2[t] = frag(0, t, 2);
#pragma unroll
      for (int t = 0; t < NTG; ++t) w1[t] = frag(0, t, 1);
#pragma unroll
      for (int t = 0; t < NTG; ++t) w0[t] = frag(0, t, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      mma_half(0);
      // ---- half B ------------------------------------------------------------
      if (a.prio) __builtin_amdgcn_s_setprio(3);
      split_half(1, pb);
      // the next tile: its indices become current, its half A is requested
      // (travels under the MFMAs of half B and the epilogue), the tile behind
      // it is asked for its indices
      cur = nxt;
      cur_ok = nxt_ok;
      request_p(cur_ok ? cur.x : 0, 0, g, tail_base, pa);
      nxt_ok = tile + 2 < tile_last && e0 + 32 + n < E;
      nxt = e2[nxt_ok ? e0 + 32 + n : 0];
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      mma_half(1);
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
  if (a.balanced) {
    grp = ws_who_balanced(a, slice, local, wave).grp;
  } else {
    while (grp + 1 < a.groups && local >= a.wg0[grp + 1]) ++grp;
  }
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
  int64_t s_first = n_wt * slice / a.xcds;
  int64_t s_last = n_wt * (slice + 1) / a.xcds;
  int64_t nw = (int64_t)(a.wg0[grp + 1] - a.wg0[grp]) * kWsWaves;
  int64_t wi = (int64_t)(local - a.wg0[grp]) * kWsWaves + wave;
  if (a.balanced) {
    const WsWho w = ws_who_balanced(a, slice, local, wave);
    s_first = 0;
    s_last = n_wt;
    nw = w.nw;
    wi = w.wi;
  }
  const int64_t span = s_last - s_first;
  const int64_t tile_first = s_first + span * wi / nw;
  const int64_t tile_last = s_first + span * (wi + 1) / nw;
#ifndef PGNN_B16_STAGGER
#define PGNN_B16_STAGGER 48
#endif
  // The two waves of a SIMD (w, w + 4) run the same phases of the same
  // length; started together they stay in lockstep -- both splitting (matrix
  // pipe idle), then both multiplying (VALU idle): the first versions' time
  // was the SUM of the phases.  The second wave starts half a tile late.
  if (wave >= 4 && PGNN_B16_STAGGER > 0) {
#pragma unroll 1
    for (int i = 0; i < PGNN_B16_STAGGER; ++i) __builtin_amdgcn_s_sleep(2);
  }
  if (ntg == NTMAX)
    edge_ws3_body<KB, NTMAX>(a, wl, t0, bias_lds, tile_first, tile_last, lane,
                             n_edges);
  else
    edge_ws3_body<KB, NTMAX - 1>(a, wl, t0, bias_lds, tile_first, tile_last,
                                 lane, n_edges);
}

}  // namespace pgnn
