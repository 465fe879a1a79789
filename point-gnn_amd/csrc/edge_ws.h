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

#ifndef PGNN_WS_SCHED
// placement of a K group's fragment requests (ds_read_b128 of the NEXT group):
// 1 = all in front of the group's MFMAs, 2 = one after every four MFMAs
// (measured: 997 -> 980 us for the edge kernel; left alone hipcc sinks them
// all behind the MFMAs and the next group starts on a cold lgkmcnt)
#define PGNN_WS_SCHED 2
#endif

constexpr int kWsMaxGroups = 4;
constexpr int kWsWaves = 8;
constexpr int kWsMaxSlices = 8;
// sched_ws layout: [0], [1] as in the LDS-tile kernels (claims, done count),
// then one pool counter per (row slice, column group)
constexpr int kWsSchedInts = 2 + kWsMaxSlices * kWsMaxGroups;
constexpr int kWsStampTiles = 38;
constexpr int kWsStampStride = 8 + 4 * kWsStampTiles;  // int64 per wave

struct EdgeWsArgs {
  const float *P, *Q;   // [num_segments, 4 * ldv4]
  int ldv4;
  const int32_t *edges;  // [n_edges, 2] rows (src, dst)
  int64_t n_edges;
  // capacity form (nullable): the edge count lives on the device and n_edges
  // is its upper bound -- min(*n_dev, n_edges) rows are processed
  const int32_t *n_dev;
  const float *wp;  // packed weights (pgnn_pack_fc), bias follows
  int nt;           // column tiles of the layer (= its K groups, except ROWS)
  int relu_from;
  float *out;
  int64_t ldo;
  int num_segments;
  int sorted;
  int xcds;                          // row slices (workgroup b -> slice b % xcds)
  int prio;                          // raise the wave priority outside the MFMA loop
  long long *ts;                     // profiling stamps (tools/ws_timeline.py) or null
  int32_t *sched;                    // tile-pool counters (kWsSchedInts, zero) or null
  int pool_pct;                      // share of a slice's tiles handed out dynamically
  int chunk;                         // ... in chunks of this many tiles
  int groups;                        // column groups
  int tile0[kWsMaxGroups + 1];       // group g owns column tiles [tile0[g], tile0[g+1])
  int wg0[kWsMaxGroups + 1];         // ... and local workgroups [wg0[g], wg0[g+1]) of a slice
  // Chip-wide balanced form (used without the tile pool): the workgroup
  // counts of a group may differ from slice to slice -- 12/10/10 of 32 in
  // every XCD left the 6-tile groups 4.7 % behind the 7-tile group, 94/81/81
  // of 256 over the chip balances them to 1.3 % -- and a group's row tiles are
  // divided over ALL its workgroups in (slice, local) order, so the
  // workgroups of one XCD still cover one contiguous range per group.
  int balanced;                               // 1: use the tables below
  int n_wg[kWsMaxGroups];                     // workgroups of group g on the chip
  short swg0[kWsMaxSlices][kWsMaxGroups + 1]; // slice s: local [swg0[s][g], swg0[s][g+1])
  short sbase[kWsMaxSlices][kWsMaxGroups];    // group g's workgroups in slices < s
  // training forward (EMIT kernels only): the layer's per-edge output rows
  // act(h W + b) are ALSO written, [n_edges, ld_rows] -- the backward finds the
  // arg-max rows by comparing them with `out`
  float *rows_out;
  int64_t ld_rows;
  // ... and, optionally, the gathered hidden rows ReLU(P[src] - Q[dst])
  // [n_edges, 4 * ldv4] (written by the workgroups of column group 0 only:
  // every group gathers the same rows)
  float *h1_out;
  // edge_ws_f16.h only: the vertex count of a capacity-form call (nullable;
  // min(*nv_dev, num_segments) rows of P / Q exist) for the range guard
  const int32_t *nv_dev;
};

// max(a, b) as ONE instruction: fmaxf() first canonicalises operands the
// compiler cannot prove quiet (accumulators, loop-carried values) with a
// v_max_f32 x, x each -- three VALU instructions per max.  v_med3_f32 with +inf
// needs none (no NaN can occur here: finite inputs, float lowest() identity);
// the +inf comes out of an asm so that instcombine cannot turn the median back
// into maxnum.
__device__ __forceinline__ float opaque_inf() {
  float inf;
  asm("v_mov_b32 %0, 0x7f800000" : "=v"(inf));
  return inf;
}
__device__ __forceinline__ float max_nc(float a, float b, float inf) {
  return __builtin_amdgcn_fmed3f(a, b, inf);
}

// a - b on two floats with one v_pk_add_f32 (hipcc scalarises the vector
// subtraction in front of the per-element ReLU; VALU slots next to the partner
// wave's MFMA stream are what this kernel is short of)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(d)
      : "v"(a), "v"(b));
  return d;
}

// ---- closing a run ---------------------------------------------------------
// `v`: per-lane partial maxima (pre-bias) of the run of segment d, 4 * NTG
// registers V[i], i = 4t + r, each with 16 row values in the 16 lanes of a DPP
// row.  A butterfly that reduced every register over all 16 lanes would take
// 4 steps x 4 NTG registers (and did: ~700 instructions per flush, three copies
// per kernel body -- cold code that had to come through the instruction cache
// every ~12 tiles: a segment-closing tile took 45-90k cycles,
// tools/ws_timeline.py).  Here the reduction HALVES the register set at every
// step (reduce-scatter): at the step over lane bit k, a lane keeps the half of
// the registers whose index has bit k equal to its own lane bit k and sends
// the other half to its partner n ^ (1 << k).  After four steps two registers
// are left and lane n holds max over the rows of V[n] and V[16 + n]: ~120
// instructions, and every lane finishes (bias, ReLU, store) its own two
// columns instead of lanes n == 0 finishing all of them.
//   steps over bits 0, 1: quad_perm DPP operand; bits 2, 3: ds_bpermute
//   (__shfl_xor), six of them.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <int NTG>
__device__ __forceinline__ float ws_v(const v4f (&v)[NTG], int i) {
  return i < 4 * NTG ? v[i >> 2][i & 3] : kFloatLowest;
}

// out[d][16 (t0 + t) + 4 g + r] <- act(max over rows + bias): plainly when the
// run is a whole segment, with float atomic-max otherwise.
// `bias_lds`: the group's 16 * NTG bias values.
template <int NTG, class ARGS>
__device__ __forceinline__ void ws_flush(const ARGS &a, const float *bias_lds,
                                         int t0, int lane, int d,
                                         const v4f (&v)[NTG], bool whole,
                                         float inf) {
  if (d < 0 || d >= a.num_segments) return;  // wave-uniform
  // An opaque zero keeps the lane predicates and addresses of this (rare) path
  // from being hoisted out of the caller's tile loop.
  int zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  const int n = (lane & 15) + zero, g = lane >> 4;
  const bool b0 = n & 1, b1 = n & 2, b2 = n & 4, b3 = n & 8;
  float A[16], B[8], C[4], D[2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (2 * j < 4 * NTG) {
      const float x0 = ws_v<NTG>(v, 2 * j), x1 = ws_v<NTG>(v, 2 * j + 1);
      A[j] = max_nc(b0 ? x1 : x0, dpp_mov<0xB1>(b0 ? x0 : x1), inf);
    } else {
      A[j] = kFloatLowest;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    B[k] = max_nc(b1 ? A[2 * k + 1] : A[2 * k],
                  dpp_mov<0x4E>(b1 ? A[2 * k] : A[2 * k + 1]), inf);
#pragma unroll
  for (int l = 0; l < 4; ++l)
    C[l] = max_nc(b2 ? B[2 * l + 1] : B[2 * l],
                  __shfl_xor(b2 ? B[2 * l] : B[2 * l + 1], 4), inf);
#pragma unroll
  for (int m = 0; m < 2; ++m)
    D[m] = max_nc(b3 ? C[2 * m + 1] : C[2 * m],
                  __shfl_xor(b3 ? C[2 * m] : C[2 * m + 1], 8), inf);
  float *orow = a.out + (int64_t)d * a.ldo + 16 * t0;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int i = 16 * m + n;  // register index this lane finishes
    if (i < 4 * NTG) {
      const int c = 16 * (i >> 2) + 4 * g + (i & 3);  // column inside the group
      // max_r act(a_r + b) == act(max_r a_r + b): +b and ReLU are monotone
      float x = D[m] + bias_lds[c];
      if (16 * t0 + c >= a.relu_from) x = x > 0.0f ? x : 0.0f;
      if (whole)
        orow[c] = x;
      else
        atomic_max_f32(orow + c, x + 0.0f);
    }
  }
}

// ---- segmented max of one 16-row tile ----------------------------------------
// The open run: the run of equal dst that contains the previous edge; its
// per-lane partial maxima sit in `carry`.
struct WsRun {
  int cur_d;         // its segment (-1 / foreign ids: never written)
  bool left_closed;  // its first edge lies inside this wave's range
  bool has;          // something was accumulated
};

// acc: out^T of the tile (lane (g, n), register r <-> feature 16t + 4g + r of
// row n); starts: bit r set where row r does not continue the run of the edge
// before it; my_d: lane (., n) holds dst of row n.  fin: the virtual tile
// behind the range (starts = 1): close the open run against d_after.
template <int NTG, class ARGS>
__device__ __forceinline__ void ws_epilogue(const ARGS &a, const float *bias_lds,
                                            int t0, int lane,
                                            const v4f (&acc)[NTG],
                                            v4f (&carry)[NTG], unsigned starts,
                                            int my_d, WsRun &st, bool fin,
                                            int d_after, float inf) {
  const int n = lane & 15;
  int cur_d = st.cur_d;
  bool cur_left_closed = st.left_closed, cur_has = st.has;
  int pos = 0;
  if (!(starts & 1u)) {
    // rows [0, f) continue the open run
    const int f = starts ? __builtin_ctz(starts) : 16;
    if (f == 16) {
#pragma unroll
      for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          carry[t][r] = max_nc(carry[t][r], acc[t][r], inf);
    } else {
      const bool in_run = n < f;
#pragma unroll
      for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          carry[t][r] =
              max_nc(carry[t][r], in_run ? acc[t][r] : kFloatLowest, inf);
    }
    cur_has = true;
    pos = f;
  }
  while (pos < 16) {  // wave-uniform; `pos` opens a run
    // the open run ends in front of row `pos`: the next edge has another dst
    // (virtual tile: the next edge is the one behind the range)
    if (cur_has)
      ws_flush<NTG>(a, bias_lds, t0, lane, cur_d, carry,
                    a.sorted && cur_left_closed &&
                        (!fin || d_after != cur_d),
                    inf);
    if (fin) break;
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
  st.cur_d = cur_d;
  st.left_closed = cur_left_closed;
  st.has = cur_has;
}

// tiles [tile_first, tile_last) of 16 edge rows, column tiles t0 .. t0+NTG-1
// whose fragments sit in `wl` ([KQ][NTG][64] float4)
//
// ROWS: the B operands are not gathered but READ -- `P` holds one ready row of
// 16 KQ floats per edge ([n_edges, 4 * ldv4]; `Q` and the edges' src column are
// not read): the second half of the split pooling stage (pool_split.h), where
// the rows are the point MLP's hidden activations.
template <int KQ, int NTG, bool EMIT, bool ROWS = false>
__device__ __forceinline__ void edge_ws_body(const EdgeWsArgs &a,
                                             const v4f *__restrict__ wl, int t0,
                                             const float *bias_lds,
                                             int64_t tile_first,
                                             int64_t tile_last, int lane,
                                             long long *tsw, int &stamped,
                                             const int64_t E) {
  if (tile_first >= tile_last) return;
  const int n = lane & 15;
  const int64_t e_first = tile_first * 16;
  const int64_t e_end = tile_last * 16 < E ? tile_last * 16 : E;
  const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(a.P);
  const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(a.Q);
  const int2 *__restrict__ e2 = reinterpret_cast<const int2 *>(a.edges);

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

  // (src, dst) of a tile's rows are requested one tile ahead.  The request is
  // unconditional (clamped index) and the validity select happens where the
  // pair is USED, a tile later: with the select -- or an if around the load --
  // next to the request, hipcc waits for it on the spot (vmcnt(0)) and the
  // index round trip is exposed in front of every gather.
  const float inf = opaque_inf();
  bool nxt_ok = e_first + n < E;
  int2 nxt = e2[nxt_ok ? e_first + n : 0];
  // (one more trip than tiles: the virtual tile behind the range closes the
  // run left open -- the same flush call site, so the flush code exists once
  // per kernel body)
  for (int64_t tile = tile_first;; ++tile) {
    const bool fin = tile >= tile_last;
    const int64_t e0 = tile * 16;
    // Everything outside the MFMA loop runs at raised priority: next to a
    // partner wave that streams MFMAs this wave's VALU instructions get few
    // issue slots (fp32 MFMA runs on the same FMA lanes; the fast-path
    // epilogue takes 0.6k cycles on an idle SIMD and 12k beside the partner's
    // MFMA phase, tools/ws_timeline.py).  Measured gain: ~1 % -- the two
    // waves of a SIMD end up alternating whole MFMA phases either way.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    // Left alone, hipcc keeps every loop-invariant per-lane value (g, n + 16,
    // P + 16 g, Q + 16 g, the LDS fragment address ...) in registers across
    // the tile loop, ran into the 256-VGPR limit and SPILLED two of them --
    // reloaded here with an s_waitcnt vmcnt(0) that serialised the index
    // prefetch behind it with the gather (gather phase 5.3k -> 9.1k cycles).
    // An opaque copy of the lane id makes them cheap per-tile recomputations
    // instead (256 -> ~200 VGPRs, no scratch).
    int lz;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
    const int g = lz >> 4;
    // fragment f of the group: wfrag[f >> 6] + (f & 63) * 64 float4 -- the
    // ds_read immediate offset reaches 64 KiB, so three bases cover 133 KiB
    // (opaque indices: hipcc otherwise folds the bases back into one and pays
    // a v_add per fragment beyond 64 KiB)
    int lz1 = lz + 64 * 64, lz2 = lz + 128 * 64;
    asm volatile("" : "+v"(lz1));
    asm volatile("" : "+v"(lz2));
    const v4f *__restrict__ wfrag[3] = {wl + lz, wl + lz1, wl + lz2};
    // profiling builds only: s_memtime at the four phase boundaries of the
    // first 38 tiles of every wave
    long long *tst = nullptr;
    if (tsw && stamped < kWsStampTiles) tst = tsw + 8 + 4 * stamped++;
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[0] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    v4f acc[NTG];
    unsigned starts = 1u;  // virtual tile: "row 0 opens a run"
    int my_d = -1;
    if (fin) {
      // (defined on both paths: an undef phi keeps the accumulators live
      // round the whole loop)
#pragma unroll
      for (int t = 0; t < NTG; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    } else {
    [[maybe_unused]] const int my_s = nxt_ok ? nxt.x : 0;
    my_d = nxt_ok ? nxt.y : -1;
    nxt_ok = tile + 1 < tile_last && e0 + 16 + n < E;
    nxt = e2[nxt_ok ? e0 + 16 + n : 0];
    // ---- gather: B operands of all K groups ------------------------------
    // rows past the end / foreign ids gather row 0 (finite values in rows the
    // epilogue never reads)
    v4f in[KQ];
    if constexpr (ROWS) {
      const int64_t row = e0 + n < E ? e0 + n : 0;
      const v4f *__restrict__ hr = P4 + row * a.ldv4 + g;
#pragma unroll
      for (int q = 0; q < KQ; ++q) in[q] = hr[4 * q];
      // all KQ loads in flight before the first use
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const int dq = ((unsigned)my_d < (unsigned)a.num_segments) ? my_d : 0;
      const v4f *__restrict__ pr = P4 + (int64_t)my_s * a.ldv4 + g;
      const v4f *__restrict__ qr = Q4 + (int64_t)dq * a.ldv4 + g;
      v4f pq[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
#ifdef PGNN_WS_ABL_NO_GATHER  // timing ablation (wrong results): one load pair
        in[q] = pr[0];
        pq[q] = qr[0];
#else
        in[q] = pr[4 * q];
        pq[q] = qr[4 * q];
#endif
      }
      // all 2*KQ loads in flight before the first use
      __builtin_amdgcn_sched_barrier(0);
      // (vector subtraction: two v_pk_add_f32 per float4 -- VALU slots are
      // what this kernel is short of next to the partner wave's MFMA stream)
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const v2f lo = pk_sub((v2f){in[q][0], in[q][1]}, (v2f){pq[q][0], pq[q][1]});
        const v2f hi = pk_sub((v2f){in[q][2], in[q][3]}, (v2f){pq[q][2], pq[q][3]});
        in[q][0] = max_nc(lo[0], 0.0f, inf);
        in[q][1] = max_nc(lo[1], 0.0f, inf);
        in[q][2] = max_nc(hi[0], 0.0f, inf);
        in[q][3] = max_nc(hi[1], 0.0f, inf);
      }
    }
    if constexpr (EMIT) {
      if (a.h1_out && t0 == 0 && e0 + n < E) {  // t0 == 0: column group 0
        v4f *hrow = reinterpret_cast<v4f *>(a.h1_out) + (e0 + n) * a.ldv4 + g;
#pragma unroll
        for (int q = 0; q < KQ; ++q) hrow[4 * q] = in[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tst) {
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[1] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- out^T = W^T h^T --------------------------------------------------
    __builtin_amdgcn_s_setprio(0);
    {
      v4f w[2][NTG];
#pragma unroll
      for (int t = 0; t < NTG; ++t) {
        acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
        w[0][t] = wfrag[0][t * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        // this group's fragments were requested a K group ago: ONE wait for
        // all of them here instead of hipcc's s_waitcnt in front of every
        // fragment's first MFMA (120 -> 45 waits per tile; measured neutral,
        // kept because the MFMA stream stays regular: 3 MFMAs, 1 request)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < KQ) {
#pragma unroll
          for (int t = 0; t < NTG; ++t)
#ifdef PGNN_WS_ABL_NO_LDS  // timing ablation (wrong results): no fragment reads
            w[(q + 1) & 1][t] = w[q & 1][t];
#else
            w[(q + 1) & 1][t] = wfrag[((q + 1) * NTG + t) >> 6]
                                     [(((q + 1) * NTG + t) & 63) * 64];
#endif
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < NTG; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                w[q & 1][t][s], in[q][s], acc[t], 0, 0, 0);
        // one K group of fragment prefetch, no more; see PGNN_WS_SCHED
#if PGNN_WS_SCHED == 2
        // one fragment request after every four MFMAs
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 3, 0);
          if (q + 1 < KQ)
            __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, 1, 0);
        }
        // (three MFMAs per request: the last fragment is then requested a
        // quarter of a group -- ~200 cycles -- before the group boundary's wait)
        __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, NTG, 0);
#else
        if (q + 1 < KQ)
          __builtin_amdgcn_sched_group_barrier(0x100 /*DS read*/, NTG, 0);
        __builtin_amdgcn_sched_group_barrier(0x8 /*MFMA*/, 4 * NTG, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (tst) {
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[2] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EMIT) {
      // the rows themselves (training): the same accumulators the maxima are
      // taken from, the same bias add and ReLU as in ws_flush -- the row that
      // holds a segment's maximum reproduces `out` bit for bit.  Lane (g, n)
      // holds features 16 t + 4 g .. + 3 of row n: one 16-byte store per tile.
      const int64_t row = e0 + n;
      if (row < E) {
        float *dstp = a.rows_out + row * a.ld_rows + 16 * t0 + 4 * g;
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
          const v4f bb = *reinterpret_cast<const v4f *>(bias_lds + 16 * t + 4 * g);
          v4f y;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[t][r] + bb[r];
            if (16 * (t0 + t) + 4 * g + r >= a.relu_from) x = x > 0.0f ? x : 0.0f;
            y[r] = x;
          }
          *reinterpret_cast<v4f *>(dstp + 16 * t) = y;
        }
      }
    }
    // ---- segmented max over the 16 rows ------------------------------------
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    // bit r of `starts`: row r does not continue the run of the edge before it
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
    if (tst) {
      __builtin_amdgcn_sched_barrier(0);
      const long long c = __builtin_readcyclecounter();
      if (lane == 0) tst[3] = c;
    }
  }
  __builtin_amdgcn_s_setprio(0);
}


// (group, first column tile, tile count, rank among the group's waves, the
// group's wave count) of this wave in the balanced partition
struct WsWho {
  int grp, t0, ntg;
  int64_t wi, nw;
};
__device__ __forceinline__ WsWho ws_who_balanced(const EdgeWsArgs &a, int slice,
                                                 int local, int wave) {
  int grp = 0;
  while (grp + 1 < a.groups && local >= a.swg0[slice][grp + 1]) ++grp;
  WsWho w;
  w.grp = grp;
  w.t0 = a.tile0[grp];
  w.ntg = a.tile0[grp + 1] - w.t0;
  w.nw = (int64_t)a.n_wg[grp] * kWsWaves;
  w.wi = (int64_t)(a.sbase[slice][grp] + local - a.swg0[slice][grp]) * kWsWaves +
         wave;
  return w;
}

template <int KQ, int NTMAX, bool EMIT = false, bool ROWS = false>
__global__ __launch_bounds__(64 * kWsWaves) void edge_ws_kernel(EdgeWsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v4f *wl = reinterpret_cast<v4f *>(smem);
  float *bias_lds = reinterpret_cast<float *>(wl + KQ * NTMAX * 64);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = blockIdx.x % a.xcds;
  const int local = blockIdx.x / a.xcds;
  // profiling builds only: when this wave entered the kernel (constant
  // 100 MHz clock), i.e. before the weights go to LDS
  const long long rt_entry = a.ts ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  int grp = 0;
  if (a.balanced) {
    grp = ws_who_balanced(a, slice, local, wave).grp;
  } else {
    while (grp + 1 < a.groups && local >= a.wg0[grp + 1]) ++grp;
  }
  const int t0 = a.tile0[grp];
  const int ntg = a.tile0[grp + 1] - t0;
  // this group's weight fragments -> LDS, [q][t][lane] float4 (1 KiB each),
  // and its bias values
  {
    // Every wave requests ALL its fragments (<= 17 dwordx4 per lane) before
    // the first LDS write: as a plain copy loop hipcc emits load ->
    // s_waitcnt vmcnt(0) -> ds_write per fragment, 17 dependent round trips
    // to L2 / HBM with 2048 waves asking at once -- ~25 us of a 985 us kernel
    // in which no SIMD had anything to do (tools/ws_timeline.py: the last wave
    // ends 40 us before the kernel does).  Unconditional clamped requests,
    // the validity test at the write.
    const v4f *__restrict__ src = reinterpret_cast<const v4f *>(a.wp);
    constexpr int PER = (KQ * NTMAX + kWsWaves - 1) / kWsWaves;
    const int n_frag = KQ * ntg;
    v4f tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      const int fc = f < n_frag ? f : 0;
      const int q = fc / ntg, t = fc - q * ntg;
      tmp[i] = src[((size_t)q * a.nt + t0 + t) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int f = wave + i * kWsWaves;
      if (f < n_frag) wl[(size_t)f * 64 + lane] = tmp[i];
    }
    if ((int)threadIdx.x < 16 * ntg)
      bias_lds[threadIdx.x] =
          a.wp[(size_t)KQ * a.nt * 256 + 16 * t0 + threadIdx.x];
  }
  __syncthreads();
  // 16-row tiles of this slice.  The first (100 - pool_pct) % of them are
  // divided statically among the group's waves (contiguous ranges: an open
  // segment is carried in registers from tile to tile); the rest is a pool the
  // waves take `chunk` tiles at a time when their own range is done -- slack
  // for waves that hit more segment boundaries, slower CUs, or a late start
  // behind another stream's kernels.  A pool chunk is a range of its own.
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
  if (a.balanced) {  // all row tiles over all the group's waves (no pool)
    const WsWho w = ws_who_balanced(a, slice, local, wave);
    s_first = 0;
    s_last = n_wt;
    nw = w.nw;
    wi = w.wi;
  }
  int64_t span = s_last - s_first;
  int64_t pool = a.sched ? span * a.pool_pct / 100 : 0;
  if (span - pool < 4 * nw) pool = 0;  // too little work to bother
  span -= pool;
  const int64_t pool_first = s_first + span;
  int64_t tile_first = s_first + span * wi / nw;
  int64_t tile_last = s_first + span * (wi + 1) / nw;
  long long *tsw = nullptr;
  if (a.ts) {
    tsw = a.ts + ((int64_t)blockIdx.x * kWsWaves + wave) * kWsStampStride;
    if (lane == 0) {
      tsw[0] = __builtin_readcyclecounter();
      tsw[2] = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
      tsw[4] = tile_last - tile_first;
      tsw[5] = ntg + 100 * slice;
      tsw[6] = rt_entry;
    }
  }
  int stamped = 0;
  int32_t *counter = a.sched ? a.sched + 2 + slice * kWsMaxGroups + grp : nullptr;
  for (;;) {
    if (ntg == NTMAX)
      edge_ws_body<KQ, NTMAX, EMIT, ROWS>(a, wl, t0, bias_lds, tile_first,
                                          tile_last, lane, tsw, stamped,
                                          n_edges);
    else
      edge_ws_body<KQ, NTMAX - 1, EMIT, ROWS>(a, wl, t0, bias_lds, tile_first,
                                              tile_last, lane, tsw, stamped,
                                              n_edges);
    if (pool == 0) break;
    int c = 0;
    if (lane == 0)
      c = __hip_atomic_fetch_add(counter, a.chunk, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
    c = __builtin_amdgcn_readfirstlane(c);
    if (c >= pool) break;
    tile_first = pool_first + c;
    tile_last = tile_first + a.chunk < s_last ? tile_first + a.chunk : s_last;
  }
  if (tsw && lane == 0) {
    tsw[1] = __builtin_readcyclecounter();
    tsw[3] = __builtin_amdgcn_s_memrealtime();
    tsw[7] = stamped;
  }
  if (a.sched && lane == 0) {
    // the last wave to get here re-arms the counters: every claim of this
    // launch was made before its wave counted itself done
    const int total = (int)gridDim.x * kWsWaves;
    const int done = __hip_atomic_fetch_add(&a.sched[1], 1, __ATOMIC_ACQ_REL,
                                            __HIP_MEMORY_SCOPE_AGENT);
    if (done == total - 1) {
      for (int i = 0; i < a.xcds * kWsMaxGroups; ++i)
        __hip_atomic_store(&a.sched[2 + i], 0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.sched[1], 0, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace pgnn
