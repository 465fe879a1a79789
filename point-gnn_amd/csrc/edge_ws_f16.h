// fp16 two-part form of the weights-stationary edge kernel: the stage of
// edge_ws.h / edge_ws_bf16.h,
//     out[d] = max over edges (s -> d) of ReLU( ReLU(P[s] - Q[d]) W + b ),
// with BOTH operands of the 300x300 product represented by TWO fp16 values,
//     x ~ x0 + x1' / 2^11,   x0 = fp16(x),  x1' = fp16((x - x0) 2^11)
// (round to nearest; the residual x - x0 is exact in fp32 and the scaling keeps
// it out of fp16's subnormals), and three fp16 MFMAs per 32-wide block,
//     x w ~ x0 w0  +  (x0 w1' + x1' w0) / 2^11,
// the first into one fp32 accumulator, the other two into a second one; every
// product of two fp16 values (11 + 11 significand bits) is exact in fp32.
// SECONDARY arithmetic `edge_arith = 'f16x2'`, beside 'bf16x3' (exact three-way
// split, six products): half the matrix instructions, 44 instead of 60 split
// instructions per block and 4 bytes per weight in LDS -- THREE column groups
// (7/6/6 tiles, 140 KiB) instead of four, i.e. the rows are gathered and
// split three times.
//
// What it gives up.  Each operand carries 22 significand bits, not 24: a
// relative representation error <= 2^-22 per element (the fp32 value's own is
// 2^-24), plus the dropped x1 w1 term (2^-22).  In the sum of 300 products the
// fp32 FMA chain's accumulated rounding dominates: NumPy model (20 000 x 300 x
// 300): this form's truncation error 9.3e-8 rms / 1.0e-6 max against the fp32
// chain's 2.7e-7 / 4.5e-6 -- the distance to float64 grows by ~6 %.  The
// parity tests hold it to the same bars as 'bf16x3' (tests/conftest.py:
// edge_arith).
//
// Range.  fp16 ends at 65504.  The gathered operand h = ReLU(P[s] - Q[d]) is
// clamped there by the v_med3_f32 that is its ReLU (no extra instruction), and
// the kernel raises bit 0 of `status` when some h COULD have reached 32768 (an
// element of P or Q at or above 16384 in magnitude, or not a number: checked
// in the kernel's prologue): the caller reruns the stage in fp32 (run.py's
// frame loop does; a trained Point-GNN's activations are below 100).  The weights' image is built
// on the host, which refuses weights outside fp16's range.
//
// Layouts: v_mfma_f32_16x16x32_f16 has the operand layouts of the bf16 form
// (edge_ws_bf16.h); the image pgnn_pack_fc_f16x2 writes is
// [kb][t][part][lane][4 u32] (1 KiB fragments, part 0 = w0, 1 = w1'), the
// layer's bias (fp32, 16 nt values) behind it.  Schedule: the interleaved
// single-wave pipeline of edge_ws_bf16.h (one MFMA, two VALU; fragment and
// row requests in between).
#pragma once
#include "edge_ws_bf16.h"

namespace pgnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// request distances of the raw rows, in blocks (edge_ws_bf16.h: PGNN_B16_DP /
// _DQ); a block here is half as long as there (3 MFMAs per tile, not 6)
#ifndef PGNN_F16_DP
#define PGNN_F16_DP PGNN_B16_DP
#endif
#ifndef PGNN_F16_DQ
#define PGNN_F16_DQ PGNN_B16_DQ
#endif
// timing ablations (WRONG results; A/B builds only, tools/sessions/r05_s18.sh):
// 1 no split arithmetic, 2 every row request to row 0, 4 fragments read once
// per tile, 8 no segmented max, 32 no v_ldexp
#ifndef PGNN_F16_ABL
#define PGNN_F16_ABL 0
#endif
constexpr float kF16Scale = 2048.0f;  // 2^11
constexpr float kF16Max = 65504.0f;

__device__ __forceinline__ u32 cvt_pk_f16(float lo, float hi) {
  return __builtin_bit_cast(u32, __builtin_convertvector((v2f){lo, hi}, f16x2));
}

__device__ __forceinline__ v4f mfma_f16(v4u a, v4u b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(
      __builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// a - (float)half of `pk` (lo / hi): exact, ONE instruction (v_fma_mix_f32 with
// an fp16 first operand).  An asm statement: written in C, hipcc converts,
// subtracts with v_pk_add_f32 and scales with v_pk_mul_f32 -- packed fp32
// arithmetic holds the matrix pipe (edge_ws_bf16.h).
__device__ __forceinline__ float sub_half_lo(float a, u32 pk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a));
  return r;
}
__device__ __forceinline__ float sub_half_hi(float a, u32 pk) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=v"(r)
      : "v"(pk), "v"(a));
  return r;
}

// min(ReLU(p - q), 65504) of 8 consecutive features -> two packed fp16 parts
// (44 instructions, stage by stage); gmax: running packed-u16 maximum of x0
template <bool TRACK = true>
__device__ __forceinline__ void split_block_f16(const v4f (&p)[2], const v4f (&q)[2],
                                                v4u &x0, v4u &x1, u32 &gmax) {
#if PGNN_F16_ABL & 1
  x0 = __builtin_bit_cast(v4u, p[0]) ^ __builtin_bit_cast(v4u, q[0]);
  x1 = __builtin_bit_cast(v4u, p[1]) ^ __builtin_bit_cast(v4u, q[1]);
  return;
#endif
  float a[8], r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = p[e >> 2][e & 3] - q[e >> 2][e & 3];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = __builtin_amdgcn_fmed3f(a[e], 0.0f, kF16Max);
#pragma unroll
  for (int j = 0; j < 4; ++j) x0[j] = cvt_pk_f16(a[2 * j], a[2 * j + 1]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[2 * j] = sub_half_lo(a[2 * j], x0[j]);
    r[2 * j + 1] = sub_half_hi(a[2 * j + 1], x0[j]);
  }
#if !(PGNN_F16_ABL & 32)
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = __builtin_ldexpf(r[e], 11);  // v_ldexp_f32
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j) x1[j] = cvt_pk_f16(r[2 * j], r[2 * j + 1]);
  // (an asm statement: written with __builtin_elementwise_max on two-half
  // vectors, hipcc 7.2 keeps ONE of the four maxima)
  if constexpr (TRACK) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm("v_pk_max_u16 %0, %0, %1" : "+v"(gmax) : "v"(x0[j]));
  }
}

// tiles [tile_first, tile_last) of 16 edge rows, column tiles t0 .. t0+NTG-1
// whose fragments sit in `wl` ([KB][NTG][2][64] v4u); `gmax`: running
// v_pk_max_u16 of every first part (the range guard)
template <int KB, int NTG>
__device__ __forceinline__ void edge_ws2_body(const EdgeWsArgs &a,
                                              const v4u *__restrict__ wl, int t0,
                                              const float *bias_lds,
                                              int64_t tile_first,
                                              int64_t tile_last, int lane,
                                              const int64_t E, u32 &gmax) {
  constexpr int DP = PGNN_F16_DP, DQ = PGNN_F16_DQ;
  static_assert(DP >= 2 && DQ >= 2 && DP < KB && DQ < KB, "request distances");
  // 36 instructions of the split are visible to the scheduler's groups (the
  // eight v_fma_mix_f32 are asm statements and place themselves)
  constexpr int kValuPerMfma = (36 + 3 * NTG - 1) / (3 * NTG);
  constexpr int kLoadEvery = 3 * NTG / 4;  // four row requests per block
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
#if PGNN_F16_ABL & 2
    const int s = 0, dq = 0;
#else
    const int s = ok ? e.x : 0;
    const int d = ok ? e.y : -1;
    const int dq = ((unsigned)d < (unsigned)a.num_segments) ? d : 0;
#endif
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
  v4u X0c, X1c;
  {
    v4f p0[2], q0[2];
    load_blk(P4, rc.p, rc.pt, 0, p0);
    load_blk(Q4, rc.q, rc.qt, 0, q0);
#pragma unroll
    for (int k = 1; k < DQ; ++k) load_blk(Q4, rc.q, rc.qt, k, qc[k - 1]);
#pragma unroll
    for (int k = 1; k < DP; ++k) load_blk(P4, rc.p, rc.pt, k, pc[k - 1]);
    split_block_f16<false>(p0, q0, X0c, X1c, gmax);
  }
  v4u w0[NTG], w1[NTG];
  {
    const v4u *__restrict__ wb = wl + lane;
#pragma unroll
    for (int t = 0; t < NTG; ++t) w0[t] = wb[(t * 2 + 0) * 64];
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
      const int f = (kb * NTG + t) * 2 + part;
      return wfrag[f >> 6][(f & 63) * 64];
    };
    v4f acc[NTG], alo[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t)
      acc[t] = alo[t] = (v4f){0.f, 0.f, 0.f, 0.f};
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
      v4u X0[KB + 1], X1[KB + 1];
#pragma unroll
      for (int k = 1; k < DP; ++k) Pb[k][0] = pc[k - 1][0], Pb[k][1] = pc[k - 1][1];
#pragma unroll
      for (int k = 1; k < DQ; ++k) Qb[k][0] = qc[k - 1][0], Qb[k][1] = qc[k - 1][1];
      X0[0] = X0c, X1[0] = X1c;
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
        split_block_f16<false>(Pb[kb + 1], Qb[kb + 1], X0[kb + 1], X1[kb + 1],
                               gmax);
        // the three terms of this block, term-major (consecutive MFMAs hit
        // different accumulators): (w0 x0) -> hi | (w0 x1') -> lo | (w1' x0)
        // -> lo.  w1' of THIS block is requested under the first term (its
        // registers were last read by the previous block's last term), w0 of
        // the next block (block 0 again behind the last) under the last term.
        const int kn = kb + 1 < KB ? kb + 1 : 0;
#if PGNN_F16_ABL & 4
        if (kb == 0) {
#pragma unroll
          for (int t = 0; t < NTG; ++t) w1[t] = frag(kb, t, 1);
        }
#else
#pragma unroll
        for (int t = 0; t < NTG; ++t) w1[t] = frag(kb, t, 1);
#endif
#pragma unroll
        for (int t = 0; t < NTG; ++t) acc[t] = mfma_f16(w0[t], X0[kb], acc[t]);
#pragma unroll
        for (int t = 0; t < NTG; ++t) alo[t] = mfma_f16(w0[t], X1[kb], alo[t]);
#if !(PGNN_F16_ABL & 4)
#pragma unroll
        for (int t = 0; t < NTG; ++t) w0[t] = frag(kn, t, 0);
#endif
#pragma unroll
        for (int t = 0; t < NTG; ++t) alo[t] = mfma_f16(w1[t], X0[kb], alo[t]);
        // issue order: MFMA, kValuPerMfma VALU; a fragment request after every
        // MFMA of the first and of the last term; the four row requests apart
#pragma unroll
        for (int m = 0; m < 3 * NTG; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2 /*VALU*/, kValuPerMfma, 0);
          if (!(PGNN_F16_ABL & 4) && (m < NTG || m >= 2 * NTG))
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
      X0c = X0[KB], X1c = X1[KB];
      // out = hi + lo / 2^11
#pragma unroll
      for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[t][r] = __builtin_fmaf(alo[t][r], 1.0f / kF16Scale, acc[t][r]);
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
#if PGNN_F16_ABL & 8
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) carry[t][r] = fmaxf(carry[t][r], acc[t][r]);
    if (fin && a.relu_from == -12345) {  // (never true; keeps every tile live)
#pragma unroll
      for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[(t * 4 + r) * 64 + lane] = carry[t][r];
    }
#else
    ws_epilogue<NTG>(a, bias_lds, t0, lane, acc, carry, starts, my_d, st, fin,
                     d_after, inf);
#endif
    cur_d = st.cur_d;
    cur_left_closed = st.left_closed;
    cur_has = st.has;
    if (fin) break;
  }
}

// a.wp: the f16x2 image (pgnn_pack_fc_f16x2) of the layer; static partition of
// the 16-row tiles as in edge_ws_kernel (no tile pool).  status (nullable):
// bit 0 is set when a gathered activation could reach 32768 (see the header)
template <int KB, int NTMAX>
__global__ __launch_bounds__(64 * kWsWaves) void edge_ws_f16x2_kernel(EdgeWsArgs a, int32_t *status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4u *wl = reinterpret_cast<v4u *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KB * NTMAX * 2 * 64);
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
  // ---- range guard, before anything else (its loads overlap the weights'):
  // workgroup b scans rows b, b + gridDim.x, ... of P and Q.  No |P| or |Q| at
  // or above 16384 => every gathered ReLU(P[s] - Q[d]) is below 32768.  (One
  // pass over 8 MB shared by all workgroups instead of a running maximum of
  // the first parts in the MFMA loop: 4 instructions per block, 4 % of the
  // kernel, tools/sessions/r05_s20.sh.)
  if (status) {
    int nv = a.num_segments;
    if (a.nv_dev) {
      const int d = *a.nv_dev;
      nv = d < nv ? d : nv;
    }
    const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(a.P);
    const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(a.Q);
    float m = 0.0f;
    bool bad = false;
    const int mine = nv > (int)blockIdx.x
                         ? (nv - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x
                         : 0;   // rows blockIdx.x + j * gridDim.x, j < mine
    for (int it = threadIdx.x; it < mine * a.ldv4; it += 64 * kWsWaves) {
      const int j = it / a.ldv4, c = it - j * a.ldv4;
      const size_t at = ((size_t)blockIdx.x + (size_t)j * gridDim.x) * a.ldv4 + c;
      const v4f p = P4[at], q = Q4[at];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m = fmaxf(m, fmaxf(fabsf(p[i]), fabsf(q[i])));
        bad |= !(p[i] == p[i]) || !(q[i] == q[i]);   // NaN: fmaxf drops it
      }
    }
    if (bad || !(m < 16384.0f)) atomicOr(status, 1);
  }
  {
    // fragments (kb, t, part) of this group -> LDS [kb][t][part][lane]; all of
    // a wave's requests in flight before its first LDS write (edge_ws_kernel)
    const v4u *__restrict__ src = reinterpret_cast<const v4u *>(a.wp);
    constexpr int PER = (KB * NTMAX * 2 + kWsWaves - 1) / kWsWaves;
    const int n_frag = KB * ntg * 2;
    v4u tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      const int fc = f < n_frag ? f : 0;
      const int kb = fc / (ntg * 2), r = fc - kb * ntg * 2;  // r = t * 2 + part
      tmp[i] = src[((size_t)(kb * a.nt + t0) * 2 + r) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      if (f < n_frag) wl[(size_t)f * 64 + lane] = tmp[i];
    }
    if ((int)threadIdx.x < 16 * ntg)
      bias_lds[threadIdx.x] =
          a.wp[(size_t)KB * a.nt * 2 * 256 + 16 * t0 + threadIdx.x];
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
  u32 gmax = 0;
  if (ntg == NTMAX)
    edge_ws2_body<KB, NTMAX>(a, wl, t0, bias_lds, tile_first, tile_last, lane,
                             n_edges, gmax);
  else
    edge_ws2_body<KB, NTMAX - 1>(a, wl, t0, bias_lds, tile_first, tile_last,
                                 lane, n_edges, gmax);
  (void)gmax;  // (the running maximum is pool_ws_f16.h's guard; here: above)
}

}  // namespace pgnn
