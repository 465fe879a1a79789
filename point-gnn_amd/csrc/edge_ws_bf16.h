// Split-bf16 form of the weights-stationary edge kernel (edge_ws.h): the same
// fused stage
//     out[d] = max over edges (s -> d) of ReLU( ReLU(P[s] - Q[d]) W + b ),
// with the 300x300 product evaluated on the bf16 matrix pipe instead of the
// fp32 one.  SECONDARY path (bench.py `secondary_bf16x3`, edge_arith =
// 'bf16x3'): the fp32-MFMA kernel stays the default and the parity reference.
//
// Why.  v_mfma_f32_16x16x4_f32 runs at the VALU's rate (157 TFLOP/s, and
// nothing issues beside it: edge_ws.h); the bf16 matrix core is 16x faster.
// An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand
// bits): x = x0 + x1 + x2, every residual exact in fp32.  With both operands
// split,
//     x w = sum_{i+j<=2} x_i w_j  +  (x1 w2 + x2 w1 + x2 w2),
// and the three dropped terms are below 2^-24 |x w| -- what one fp32 FMA
// commits.  Every kept product of two bf16 values is exact in fp32, so six bf16
// MFMAs accumulating in fp32 reproduce the fp32 product to fp32 rounding; what
// remains is the accumulation's own fp32 rounding, of which this form has
// fewer steps (one per 32-wide block and term instead of one per element).
// Measured on the bench frames: max |logit - float64 oracle| no larger than the
// fp32-MFMA kernel's (tests/test_gpu_bf16x3.py prints both).
//
// Cost.  6 x v_mfma_f32_16x16x32_bf16 (16 cycles each) per 32 x 16 x 16 block
// = 96 cycles against 8 x 32 = 256 for fp32: 2.67x fewer matrix cycles; three
// bf16 images of the weights (6 bytes per weight) put 19 column tiles in FOUR
// groups of 5/5/5/4 (150 KiB) instead of three, i.e. the rows are gathered and
// split four times; and the split is VALU work per gathered element.
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
//
// Schedule (round 5; the round-4 body alternated a VALU phase -- gather, ReLU,
// split -- with an MFMA phase per half tile and left the overlap to the SIMD's
// two waves: it does not happen, profiles/r04_pmc_sq_bf16x3.txt shows VALU and
// matrix pipe busy together for 7 % of the matrix cycles and a kernel time
// equal to the SUM of the phases, at any wave priority).  What was measured,
// step by step (tools/micro/mfma_mix.hip -> profiles/r05_mfma_mix.txt,
// profiles/r05_bf16x3_steps.txt):
//   * beside a stream of v_mfma_f32_16x16x32_bf16 the MFMA takes two of the
//     four issue slots of its 16 cycles; v_sub / v_and / v_med3 / v_perm /
//     v_cvt_pk_bf16_f32 take one each: TWO per MFMA are free, every further one
//     costs 4 cycles, a DEPENDENT neighbour 3 more -- and one v_pk_add_f32
//     costs 14-16 cycles that overlap nothing (packed fp32 holds the matrix
//     pipe): as much as the MFMA itself.  The round-4 body had 12 per block.
//   * the overlap is therefore written into ONE wave's instruction stream: a
//     software pipeline over the tile's K blocks in which the 6 NTG MFMAs of
//     block kb are interleaved (sched_group_barrier: one MFMA, two VALU) with
//     the split of block kb + 1 -- 60 single-slot instructions, stage by stage
//     over the block's eight elements so that neighbours are independent --,
//     the fragment requests of block kb + 1 and the row requests of blocks
//     kb + 2 (Q) / kb + 3 (P), across the tile boundary (the next tile's first
//     blocks are requested and split under the last MFMAs of this one).
//   * the four 1 KiB row requests of a block are spread over it: issued back
//     to back they stall the in-order wave at the address path (16 cycles per
//     request): 677 us against 598.
//   * at the tile boundary the dst of the row above comes through a DPP row
//     shift, not __shfl_up: a ds_bpermute has to wait for lgkmcnt(0), i.e. for
//     the fifteen fragment requests of the next tile issued just before.
//   * tried and dropped: segment-aligned tiles with the tile's Q row held in
//     registers, distributed over the lanes and read through the DPP operand
//     row_newbcast:kb of the subtraction (v_subrev_f32_dpp; two row requests
//     per tile instead of twenty): + 4 % tiles, 594-600 us against 577 on the
//     same box at C = 300, 1438 against 1483 at C = 256 (r05_s8).
//
//   live per lane: parts of 2 blocks (24), raw P of 3 blocks (24), raw Q of 2
//   (16), 3 NTG fragments (60), accumulators + carry (40); 189 VGPRs
//
// PGNN_B16_TRUNC 2 (default): first part rounded to nearest (v_cvt_pk_bf16_f32),
//   the 16 residual bits cut in two by truncation (v_and + exact v_sub, packed
//   with v_perm): still x = x0 + x1 + x2 exactly, 60 instructions per block,
//   dropped terms 1.2e-7 against the fp32 chain's 4.9e-6 (NumPy model,
//   20 000 x 300 x 300);
//   0: all three parts rounded (68 instructions: + 3 %);
//   1: all three cut (60; dropped terms 2.6e-7, but its error against float64 is
//   1.4x the fp32 kernel's on the C = 256 test where the others' is 1.0x)
#pragma once
#include <type_traits>
#include <utility>

#include "edge_ws.h"

#ifndef PGNN_B16_TRUNC
#define PGNN_B16_TRUNC 2
#endif
// timing ablations (WRONG results; tools/ only): 2 = one part instead of
// three, 4 = the weight fragments are read once, 8 = one term of six,
// 16 = one wave per SIMD (waves 4..7 leave, the others take their tiles)
#ifndef PGNN_B16_ABL
#define PGNN_B16_ABL 0
#endif
#ifndef PGNN_B16_DP  // raw P rows are requested DP blocks ahead of their split
#define PGNN_B16_DP 3
#endif
#ifndef PGNN_B16_DQ  // ... raw Q rows (L1-hot: mostly one dst per tile) DQ blocks
#define PGNN_B16_DQ 2
#endif

namespace pgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

// (lo, hi) -> packed bf16 pair, round to nearest even: v_cvt_pk_bf16_f32 (not
// an asm statement: the scheduler's instruction groups count it as VALU)
__device__ __forceinline__ u32 cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(u32, __builtin_convertvector((v2f){lo, hi}, bf16x2));
}

// (hi & 0xffff0000) | (lo >> 16): two floats' upper halves, i.e. their bf16
// truncations, packed
__device__ __forceinline__ u32 perm_hi16(u32 hi, u32 lo) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

__device__ __forceinline__ v4f mfma_bf16(v4u a, v4u b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
      __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <class F, int... I>
__device__ __forceinline__ void for_each_int(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// ReLU(p - q) of 8 consecutive features -> three packed bf16 parts, stage by
// stage over the eight elements; no packed fp32 arithmetic (see the header)
__device__ __forceinline__ void split_block(const v4f (&p)[2], const v4f (&q)[2],
                                            v4u &x0, v4u &x1, v4u &x2,
                                            float inf) {
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = p[e >> 2][e & 3] - q[e >> 2][e & 3];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = max_nc(a[e], 0.0f, inf);
#if PGNN_B16_ABL & 2
#pragma unroll
  for (int j = 0; j < 4; ++j)
    x0[j] = x1[j] = x2[j] =
        perm_hi16(__float_as_uint(a[2 * j + 1]), __float_as_uint(a[2 * j]));
#else
  float r[8];
#if PGNN_B16_TRUNC == 1
#pragma unroll
  for (int j = 0; j < 4; ++j)
    x0[j] = perm_hi16(__float_as_uint(a[2 * j + 1]), __float_as_uint(a[2 * j]));
#pragma unroll
  for (int e = 0; e < 8; ++e)
    r[e] = __uint_as_float(__float_as_uint(a[e]) & 0xffff0000u);
#else
#pragma unroll
  for (int j = 0; j < 4; ++j) x0[j] = cvt_pk_bf16(a[2 * j], a[2 * j + 1]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[2 * j] = __uint_as_float(x0[j] << 16);
    r[2 * j + 1] = __uint_as_float(x0[j] & 0xffff0000u);
  }
#endif
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = a[e] - r[e];  // exact
#if PGNN_B16_TRUNC == 0
  float t[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) x1[j] = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[2 * j] = __uint_as_float(x1[j] << 16);
    t[2 * j + 1] = __uint_as_float(x1[j] & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = r[e] - t[e];  // exact, <= 8 bits
#pragma unroll
  for (int j = 0; j < 4; ++j) x2[j] = cvt_pk_bf16(t[2 * j], t[2 * j + 1]);
#else
  float s[8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    x1[j] = perm_hi16(__float_as_uint(r[2 * j + 1]), __float_as_uint(r[2 * j]));
#pragma unroll
  for (int e = 0; e < 8; ++e)
    s[e] = __uint_as_float(__float_as_uint(r[e]) & 0xffff0000u);
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = r[e] - s[e];  // exact, <= 8 bits
#pragma unroll
  for (int j = 0; j < 4; ++j)
    x2[j] = perm_hi16(__float_as_uint(s[2 * j + 1]), __float_as_uint(s[2 * j]));
#endif
#endif
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
  constexpr int DP = PGNN_B16_DP, DQ = PGNN_B16_DQ;
  static_assert(DP >= 2 && DQ >= 2 && DP < KB && DQ < KB, "request distances");
  constexpr int kSplitValu = PGNN_B16_TRUNC == 0 ? 68 : 60;
  constexpr int kValuPerMfma = (kSplitValu + 6 * NTG - 1) / (6 * NTG);
  constexpr int kLoadEvery = 6 * NTG / 4;  // four row requests per block
  constexpr int kNext = KB - 1 - (DP > DQ ? DP : DQ);  // block that sets up the next tile's rows
  static_assert(kNext >= 0, "request distances");
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

  // block kb of a row: v4f 8 kb + 2 g + i of the row; the tail block (lanes
  // g >= 2 of it lie behind the row's 4 * ldv4 floats) from a clamped offset.
  // 32-bit byte offsets from the (scalar) matrix bases, the tail block's
  // precomputed: no address arithmetic between the MFMAs (the launcher
  // refuses matrices of 4 GiB and more)
  struct Rows {
    u32 p, pt, q, qt;
  };
  auto rows_of = [&](int2 e, bool ok, int g, int toff) -> Rows {
    const int s = ok ? e.x : 0;
    const int d = ok ? e.y : -1;
    const int dq = ((unsigned)d < (unsigned)a.num_segments) ? d : 0;
    Rows r;
    r.p = ((u32)s * (u32)a.ldv4 + 2u * g) * 16u;
    r.q = ((u32)dq * (u32)a.ldv4 + 2u * g) * 16u;
    r.pt = r.p + 16u * toff;
    r.qt = r.q + 16u * toff;
    return r;
  };
  auto tail_off = [&](int g) {  // relative to row + 2 g
    const int t = 8 * (KB - 1) + 2 * g;
    return (t < last - 1 ? t : last - 1) - 2 * g;
  };
  auto load_blk = [&](const v4f *__restrict__ base, u32 off, u32 off_tail,
                      int kb, v4f (&o)[2]) {
    const char *__restrict__ b = reinterpret_cast<const char *>(base);
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(
        kb == KB - 1 ? b + (size_t)off_tail : b + (size_t)off + 128 * kb);
    o[0] = src[0];
    o[1] = src[1];
  };

  // ---- before the first tile: its indices, its first blocks, the parts of
  // block 0, the fragments of block 0
  bool cur_ok = e_first + n < E;
  int2 cur = e2[cur_ok ? e_first + n : 0];
  Rows rc = rows_of(cur, cur_ok, lane >> 4, tail_off(lane >> 4));
  // carried round the tile loop: raw P of blocks 1 .. DP-1, raw Q of blocks
  // 1 .. DQ-1 (in flight), the parts of block 0
  v4f pc[DP - 1][2], qc[DQ - 1][2];
  v4u X0c, X1c, X2c;
  {
    v4f p0[2], q0[2];
    load_blk(P4, rc.p, rc.pt, 0, p0);
    load_blk(Q4, rc.q, rc.qt, 0, q0);
#pragma unroll
    for (int k = 1; k < DQ; ++k) load_blk(Q4, rc.q, rc.qt, k, qc[k - 1]);
#pragma unroll
    for (int k = 1; k < DP; ++k) load_blk(P4, rc.p, rc.pt, k, pc[k - 1]);
    split_block(p0, q0, X0c, X1c, X2c, inf);
  }
  v4u w0[NTG], w1[NTG], w2[NTG];
  {
    const v4u *__restrict__ wb = wl + lane;
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
      w2[t] = wb[(t * 3 + 2) * 64];
      w1[t] = wb[(t * 3 + 1) * 64];
      w0[t] = wb[(t * 3 + 0) * 64];
    }
  }

  for (int64_t tile = tile_first;; ++tile) {
    const bool fin = tile >= tile_last;
    const int64_t e0 = tile * 16;
    int lz;  // opaque per-tile lane id: see edge_ws_body
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    const int toff = tail_off(g);
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
      my_d = cur_ok ? cur.y : -1;
      // the next tile's indices: requested now, used a few blocks before the end
      const bool nxt_ok = tile + 1 < tile_last && e0 + 16 + n < E;
      const int2 nxt = e2[nxt_ok ? e0 + 16 + n : 0];
      Rows rn = rc;
      // Pb[k], Qb[k]: raw rows of block k; k >= KB: block k - KB of the next tile
      v4f Pb[KB + DP][2], Qb[KB + DQ][2];
      v4u X0[KB + 1], X1[KB + 1], X2[KB + 1];
#pragma unroll
      for (int k = 1; k < DP; ++k) Pb[k][0] = pc[k - 1][0], Pb[k][1] = pc[k - 1][1];
#pragma unroll
      for (int k = 1; k < DQ; ++k) Qb[k][0] = qc[k - 1][0], Qb[k][1] = qc[k - 1][1];
      X0[0] = X0c, X1[0] = X1c, X2[0] = X2c;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        __builtin_amdgcn_sched_barrier(0);
        // (the next tile's row offsets, an iteration before their first use,
        // in a scheduling region of their own: the issue pattern below counts
        // this region's instructions)
        if (kb == kNext) {
          rn = rows_of(nxt, nxt_ok, g, toff);
          __builtin_amdgcn_sched_barrier(0);
        }
        // row requests (vmcnt counts in order: Q, wanted sooner, before P)
        if (kb + DQ < KB)
          load_blk(Q4, rc.q, rc.qt, kb + DQ, Qb[kb + DQ]);
        else
          load_blk(Q4, rn.q, rn.qt, kb + DQ - KB, Qb[kb + DQ]);
        if (kb + DP < KB)
          load_blk(P4, rc.p, rc.pt, kb + DP, Pb[kb + DP]);
        else
          load_blk(P4, rn.p, rn.pt, kb + DP - KB, Pb[kb + DP]);
        // parts of the next block (block 0 of the next tile behind the last)
        split_block(Pb[kb + 1], Qb[kb + 1], X0[kb + 1], X1[kb + 1], X2[kb + 1],
                    inf);
        // the six terms of this block, term-major (consecutive MFMAs hit
        // different accumulators), in an order that frees a weight part's
        // registers as early as possible -- (w2 x0) | (w1 x1) (w1 x0) |
        // (w0 x2) (w0 x1) (w0 x0) --; a part's fragments of the next block
        // (block 0 again behind the last) are requested when it is done
        const int kn = kb + 1 < KB ? kb + 1 : 0;
        constexpr bool kLds = !(PGNN_B16_ABL & 4), kAll = !(PGNN_B16_ABL & 8);
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w2[t], X0[kb], acc[t]);
        if (kLds) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w2[t] = frag(kn, t, 2);
        }
        if (kAll) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], X1[kb], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w1[t], X0[kb], acc[t]);
        }
        if (kLds) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w1[t] = frag(kn, t, 1);
        }
        if (kAll) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], X2[kb], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], X1[kb], acc[t]);
#pragma unroll
          for (int t = 0; t < NTG; ++t) acc[t] = mfma_bf16(w0[t], X0[kb], acc[t]);
        }
        if (kLds) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w0[t] = frag(kn, t, 0);
        }
        // issue order: MFMA, kValuPerMfma VALU (the split is 60 / 68
        // instructions per block: two per MFMA at five column tiles, three at
        // four); a fragment request after every second MFMA from the second
        // term on; the four row requests apart
#pragma unroll
        for (int m = 0; m < 6 * NTG; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2 /*VALU*/, kValuPerMfma, 0);
          if (m >= NTG && (m - NTG) % 2 == 1 && (m - NTG) / 2 < 3 * NTG)
            __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, 1, 0);
          if (m % kLoadEvery == 1 && m / kLoadEvery < 4)
            __builtin_amdgcn_sched_group_barrier(0x20 /*VMEM read*/, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 1; k < DP; ++k)
        pc[k - 1][0] = Pb[KB + k][0], pc[k - 1][1] = Pb[KB + k][1];
#pragma unroll
      for (int k = 1; k < DQ; ++k)
        qc[k - 1][0] = Qb[KB + k][0], qc[k - 1][1] = Qb[KB + k][1];
      X0c = X0[KB], X1c = X1[KB], X2c = X2[KB];
      rc = rn;
      cur = nxt;
      cur_ok = nxt_ok;
      // ---- segmented max over the 16 rows: as in edge_ws_body, but the dst of
      // the row above comes through a DPP row shift (lane n - 1 of the same
      // 16-lane row; lane 0 keeps the open run's id), not __shfl_up
      const int prev = __builtin_amdgcn_update_dpp(cur_d, my_d, 0x111 /*row_shr:1*/,
                                                   0xF, 0xF, false);
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
#if PGNN_B16_ABL & 16  // timing ablation: one wave per SIMD does all the tiles
  if (wave >= 4) return;
  nw /= 2;
  wi = (wi / kWsWaves) * 4 + wave;
#endif
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
