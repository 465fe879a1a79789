// Workgroup-level fp32 MLP engine on MFMA (v_mfma_f32_16x16x4_f32).
//
// A workgroup (4 waves, one per SIMD) owns a tile of ROWS = 16*MSUB rows whose
// activations live in LDS, row-major with leading dimension ld = width + 8
// floats (ld == 8 mod 16: the four 16-lane groups of a ds_read_b128 A-fragment
// read land on disjoint banks).  A layer y = act(x W + b) is computed as
//   - A fragments: one ds_read_b128 per 16 rows per K-group of 16: lane l
//     holds x[row = l&15][k = 16q + 4(l>>4) + s], s = 0..3;
//   - B fragments: straight from global/L2 in the host-packed order
//     (pgnn_pack_fc): lane l holds W[16q + 4(l>>4) + s][16t + (l&15)] --
//     one coalesced 1 KiB load per (K-group, column tile), no LDS staging;
//     weights are a few hundred KB and stay L2-resident;
//   - 4 MFMA steps (s = 0..3) consume them; the k order inside a group is
//     permuted identically on both operands, so the sum is over all k.
// The N dimension is split across the 4 waves (column tile t = t0 + wave + 4j),
// every wave reads the whole A tile from LDS (LDS traffic is ~1% of MFMA
// time) and streams only its own quarter of W.  Accumulators stay in
// registers; after a barrier the activated result overwrites the input tile
// in place.  fp32 MFMA is bit-exact fp32 FMA, so results match an fp32
// reference to rounding order.
#pragma once
#include "pgnn_common.h"

namespace pgnn {

#ifndef PGNN_PF4
#define PGNN_PF4 2  // register stages of the K-group pipeline for 64-row tiles
#endif
#ifndef PGNN_PF1
#define PGNN_PF1 4  // ... for 16-row tiles (the K-row kernels)
#endif

typedef float v4f __attribute__((ext_vector_type(4)));

struct LayerDev {
  const float *wp;  // packed weights, bias follows at wp + kq*nt*256
  int kq;           // K groups of 16
  int nt;           // column tiles of 16
  int relu_from;    // ReLU on columns >= relu_from
};

struct ChainDev {
  int n;
  LayerDev l[PGNN_MAX_LAYERS];
};

constexpr int kMaxTilesPerPass = 20;  // 4 waves x NT<=5 column tiles = 320 cols

__host__ __device__ inline int lds_ld(int width16) { return width16 + 8; }

// acc[m][j] += tile[16m.., :] @ W[:, tile t0 + wave + NW*j]  (NW waves share
// the N dimension; 4 by default, 8 for the latency-bound 16-row kernels)
template <int MSUB, int NT, int NW = 4>
__device__ __forceinline__ void gemm_tile(const float *__restrict__ tile, int ld,
                                          const LayerDev &L, int t0, int wave,
                                          int lane, v4f (&acc)[MSUB][NT]) {
#pragma unroll
  for (int m = 0; m < MSUB; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  int toff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int t = t0 + wave + NW * j;
    if (t > L.nt - 1) t = L.nt - 1;  // clamp: load something valid, discard later
    toff[j] = t * 64;
  }
  const v4f *__restrict__ wp = reinterpret_cast<const v4f *>(L.wp) + lane;
  const float *arow = tile + (lane & 15) * ld + 4 * (lane >> 4);
  const int qstride = L.nt * 64;
  const int kq = L.kq;
  // Software pipeline over K-groups with PF register stages: a stage is
  // refilled (fragments of group q + PF) right after its MFMAs consumed it, so
  // L2 / LDS latency hides under PF-1 groups of matrix work.  Big tiles
  // (MSUB = 4: 80 MFMAs ~ 2.5k cycles per group) need 2 stages; 16-row tiles
  // (MSUB = 1: 20 MFMAs ~ 640 cycles per group, far less than an L2 round
  // trip) get 4.  The prefetch index is clamped, so the tail re-reads a valid
  // group instead of branching around loads.
  constexpr int PF = MSUB >= 4 ? PGNN_PF4 : (MSUB == 2 ? 3 : PGNN_PF1);
  v4f a[PF][MSUB], b[PF][NT];
  auto fetch = [&](int q, v4f (&fa)[MSUB], v4f (&fb)[NT]) {
    if (q > kq - 1) q = kq - 1;
#if defined(PGNN_KROW_ABL_NOLOAD)  // timing ablation (wrong results)
    if constexpr (NW == 8) {
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = wp[toff[j]];
    } else
#elif defined(PGNN_KROW_ABL_SAMEQ)  // timing ablation: every wave the same 1 KiB
    if constexpr (NW == 8) {
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = wp[(size_t)q * qstride];
    } else
#endif
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = wp[(size_t)q * qstride + toff[j]];
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
      fa[m] = *reinterpret_cast<const v4f *>(arow + m * 16 * ld + 16 * q);
  };
  // k-steps [s0, s1) of one K-group
  auto mma = [&](const v4f (&fa)[MSUB], const v4f (&fb)[NT], int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s)
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][s], fb[j][s],
                                                           acc[m][j], 0, 0, 0);
  };
#pragma unroll
  for (int st = 0; st < PF; ++st) fetch(st, a[st], b[st]);
  // (A branch-free body with the kq % PF leftover groups peeled off lets hipcc
  // place exact waits instead of vmcnt(0) at the loop header and looks better
  // in tools/gemm_loop_bench.hip, but the fused kernels ran 3 % SLOWER with it
  // in a same-box A/B -- 834 -> 863 us for the edge kernel -- so the guarded
  // form stays.)
#ifndef PGNN_KLOOP_BRANCHFREE
#define PGNN_KLOOP_BRANCHFREE 0  // 1: every LDS-tile kernel (A/B builds)
#endif
  // (also the 32-row LDS-tile kernels -- ped_cyl's pooling stage: 2 876 ->
  // 2 813 us in a same-box A/B; the 64-row form measured 3 % SLOWER with it in
  // round 2 and keeps the guarded loop below)
  if constexpr (NW == 8 || MSUB == 2 || PGNN_KLOOP_BRANCHFREE) {
    // The 16-row K-row kernels (rows_mlp_kernel, vertex_*_kernel): 12 MFMAs
    // per K-group and wave, far less than an L2 round trip, so the prefetch
    // distance IS the kernel's speed.  With the guarded body below hipcc's
    // wait-count pass cannot tell how many loads are outstanding at the loop
    // header (the guards make it path dependent) and emits s_waitcnt vmcnt(0)
    // there: every PF groups the pipeline drained and a layer pass took ~13 us
    // for 5 us of MFMA issue.  Branch-free main loop (whole rounds of PF
    // groups; the prefetch index is clamped, never guarded) + a peeled tail
    // without loads: the header wait becomes vmcnt(loads of PF - 1 stages).
    int q = 0;
    for (; q + PF <= kq; q += PF) {
#pragma unroll
      for (int st = 0; st < PF; ++st) {
        mma(a[st], b[st], 0, 4);
        fetch(q + st + PF, a[st], b[st]);
      }
    }
    const int rem = kq - q;  // < PF, wave-uniform
#pragma unroll
    for (int st = 0; st + 1 < PF; ++st)
      if (st < rem) mma(a[st], b[st], 0, 4);
    return;
  }
  for (int q = 0; q < kq; q += PF) {
#pragma unroll
    for (int st = 0; st < PF; ++st) {
      if (q + st < kq) {  // wave-uniform
        mma(a[st], b[st], 0, 4);
        fetch(q + st + PF, a[st], b[st]);
      }
    }
  }
}

// out[row][col - 16*t0] = act(acc + bias[col]); C/D layout of the 16x16 MFMA:
// col = lane & 15, row = 4*(lane >> 4) + r.
template <int MSUB, int NT, int NW = 4>
__device__ __forceinline__ void store_acc(float *__restrict__ out, int ldo,
                                          const LayerDev &L, int t0, int wave,
                                          int lane, const v4f (&acc)[MSUB][NT]) {
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + wave + NW * j;
    // (with NW = 8 a wave's last tile can lie beyond this pass's 20 tiles)
    if (t < L.nt && t - t0 < kMaxTilesPerPass) {
      const int col = t * 16 + (lane & 15);
      const float bv = bias[col];
      const bool relu = col >= L.relu_from;
      float *o = out + (4 * (lane >> 4)) * ldo + (col - 16 * t0);
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[m][j][r] + bv;
          if (relu) v = v > 0.0f ? v : 0.0f;
          o[(m * 16 + r) * ldo] = v;
        }
    }
  }
}

// Transposed, swizzled variant for the layer feeding the segmented max:
// stT[c][g] (g = row group of 4) at float offset c*ROWS + ((g ^ (c & SWZ)) << 2).
// A lane's four accumulator rows are contiguous here, so the whole C fragment
// goes out as one ds_write_b128 (instead of four ds_write_b32), and the
// column-wise reduction later reads 4 rows per ds_read_b128.  The XOR spreads
// both the 8-lane write groups and the 16-lane read groups over distinct
// 16-byte bank slots.
template <int MSUB, int NT>
__device__ __forceinline__ void store_acc_T(float *__restrict__ stT,
                                            const LayerDev &L, int t0, int wave,
                                            int lane, const v4f (&acc)[MSUB][NT]) {
  constexpr int ROWS = 16 * MSUB, G = ROWS / 4, SWZ = (G < 16 ? G : 16) - 1;
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + wave + 4 * j;
    if (t < L.nt) {
      const int col = t * 16 + (lane & 15);
      const float bv = bias[col];
      const bool relu = col >= L.relu_from;
      const int c = col - 16 * t0;
      float *o = stT + c * ROWS;
      const int sw = c & SWZ;
#pragma unroll
      for (int m = 0; m < MSUB; ++m) {
        const int g = 4 * m + (lane >> 4);
        v4f v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[m][j][r] + bv;
          if (relu) x = x > 0.0f ? x : 0.0f;
          v[r] = x;
        }
        *reinterpret_cast<v4f *>(o + ((g ^ sw) << 2)) = v;
      }
    }
  }
}

// Fast epilogue for the layer feeding a segmented max when ALL rows of the tile
// belong to one segment (the common case at level 1: mean fan-in ~170 vs 64-row
// tiles): the column max is taken straight from the accumulators -- 16 values
// per lane, then two cross-lane steps over the four row groups -- so the tile
// never goes back through LDS and two workgroup barriers disappear.
// max_r act(a_r + b) == act(max_r a_r + b) because +b and ReLU are monotone.
struct SegFast {
  float *carry;      // LDS, one float per output column (absolute index)
  float *out_row;    // global row of the segment (already offset to column 0)
  int merge;         // fold carry[] in first
  int defer;         // keep in carry[] instead of writing out
  int whole;         // plain store (complete segment) vs atomic max
};

template <int MSUB, int NT>
__device__ __forceinline__ void layer_pass_segmax_fast(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegFast &sf) {
  v4f acc[MSUB][NT];
  gemm_tile<MSUB, NT>(in, ld_in, L, t0, wave, lane, acc);
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + wave + 4 * j;
    if (t < L.nt) {
      float v = acc[0][j][0];
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaxf(v, acc[m][j][r]);
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      const int col = t * 16 + (lane & 15);
      v += bias[col];
      if (col >= L.relu_from) v = v > 0.0f ? v : 0.0f;
      if (lane < 16) {
        if (sf.merge) v = fmaxf(v, sf.carry[col]);
        if (sf.defer) {
          sf.carry[col] = v;
        } else if (sf.whole) {
          sf.out_row[col] = v;
        } else {
          atomic_max_f32(sf.out_row + col, v + 0.0f);
        }
      }
    }
  }
}

template <int MSUB>
__device__ __forceinline__ void layer_pass_segmax_fast_dispatch(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegFast &sf) {
  int tiles = L.nt - t0;
  if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
  const int ntw = (tiles + 3) >> 2;
  switch (ntw) {
    case 1: layer_pass_segmax_fast<MSUB, 1>(in, ld_in, L, t0, wave, lane, sf); break;
    case 2: layer_pass_segmax_fast<MSUB, 2>(in, ld_in, L, t0, wave, lane, sf); break;
    case 3: layer_pass_segmax_fast<MSUB, 3>(in, ld_in, L, t0, wave, lane, sf); break;
    case 4: layer_pass_segmax_fast<MSUB, 4>(in, ld_in, L, t0, wave, lane, sf); break;
    default: layer_pass_segmax_fast<MSUB, 5>(in, ld_in, L, t0, wave, lane, sf); break;
  }
}

// Epilogue for a tile that holds a FEW runs of equal dst (sorted edges; mean
// fan-in 120-170 vs 64-row tiles: about a third of the tiles contain one or two
// segment boundaries).  Same idea as the one-segment path, once per run with a
// row mask: a lane's accumulator element (m, i) is tile row 16m + 4(lane>>4) + i,
// so "row in [r, re)" is 4*MSUB compares per run shared by all column tiles,
// then v_cndmask/v_max over the accumulators and the two cross-lane steps.
// Replaces, for those tiles, the transposed LDS stage + two barriers + the
// column walk of consume_segmax (tools/pool_timeline.py: 24k cycles a tile).
struct SegRuns {
  unsigned long long starts;  // bit r: tile row r opens a run (row 0 always)
  int myd;                    // lane r: dst of tile row r (-1 past the end)
  int d_before, d_after;      // dst of the edge before / after the tile, or -1
  int carry_id, carry_left_closed;  // CarryState coming into the tile
  int keep_open;              // the workgroup owns the next tile too
  int sorted;                 // ids non-decreasing: closed runs are whole segments
  float *carry;               // LDS, one float per output column
  float *out;
  int64_t ldo;
  int num_segments;
};

// acc[m][j] = pre-bias accumulators of rows 16m.. and column tile t0+wave+4j
template <int MSUB, int NT>
__device__ __forceinline__ void segmax_runs_emit(const v4f (&acc)[MSUB][NT],
                                                 const LayerDev &L, int t0,
                                                 int wave, int lane,
                                                 const SegRuns &sr) {
  constexpr int ROWS = 16 * MSUB;
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
  const int g4 = 4 * (lane >> 4);
  unsigned long long rem = sr.starts;
  while (rem) {  // wave-uniform
    const int r = __builtin_ctzll(rem);
    rem &= rem - 1;
    const int re = rem ? __builtin_ctzll(rem) : ROWS;
    const int d = __builtin_amdgcn_readlane(sr.myd, r);
    if (d < 0 || d >= sr.num_segments) continue;
    bool left_closed = (r > 0) || (sr.d_before != d);
    const bool merge = r == 0 && sr.carry_id == d;
    if (merge) left_closed = sr.carry_left_closed != 0;
    const bool right_closed = (re < ROWS) || (sr.d_after != d);
    const bool defer = !right_closed && sr.keep_open;
    const bool whole = sr.sorted && left_closed && right_closed;
    float *out_row = sr.out + (int64_t)d * sr.ldo;
    const int lo = r - g4, hi = re - g4;
    bool inside[MSUB][4];
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        inside[m][i] = (16 * m + i >= lo) && (16 * m + i < hi);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = t0 + wave + 4 * j;
      if (t < L.nt) {
        float v = kFloatLowest;
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            v = fmaxf(v, inside[m][i] ? acc[m][j][i] : kFloatLowest);
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        const int col = t * 16 + (lane & 15);
        v += bias[col];
        if (col >= L.relu_from) v = v > 0.0f ? v : 0.0f;
        if (lane < 16) {
          if (merge) v = fmaxf(v, sr.carry[col]);
          if (defer) {
            sr.carry[col] = v;
          } else if (whole) {
            out_row[col] = v;
          } else {
            atomic_max_f32(out_row + col, v + 0.0f);
          }
        }
      }
    }
  }
}

template <int MSUB, int NT>
__device__ __forceinline__ void layer_pass_segmax_runs(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegRuns &sr) {
  v4f acc[MSUB][NT];
  gemm_tile<MSUB, NT>(in, ld_in, L, t0, wave, lane, acc);
  segmax_runs_emit<MSUB, NT>(acc, L, t0, wave, lane, sr);
}

template <int MSUB>
__device__ __forceinline__ void layer_pass_segmax_runs_dispatch(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegRuns &sr) {
  int tiles = L.nt - t0;
  if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
  const int ntw = (tiles + 3) >> 2;
  switch (ntw) {
    case 1: layer_pass_segmax_runs<MSUB, 1>(in, ld_in, L, t0, wave, lane, sr); break;
    case 2: layer_pass_segmax_runs<MSUB, 2>(in, ld_in, L, t0, wave, lane, sr); break;
    case 3: layer_pass_segmax_runs<MSUB, 3>(in, ld_in, L, t0, wave, lane, sr); break;
    case 4: layer_pass_segmax_runs<MSUB, 4>(in, ld_in, L, t0, wave, lane, sr); break;
    default: layer_pass_segmax_runs<MSUB, 5>(in, ld_in, L, t0, wave, lane, sr); break;
  }
}

// ---------------------------------------------------------------------------
// Balanced split for 64-row tiles whose pass has 4*NT + NR column tiles
// (NR = 1..3; C = 300 -> 19 tiles = 4*4 + 3).  The plain scheme gives every
// wave NT+1 tiles and lets the waves past the end compute a clamped duplicate
// -- 80 instead of 76 tile-units of MFMA work per wave for 19 tiles.  Here a
// wave owns NT full column tiles (all four 16-row sub-tiles) and, of each of
// the NR leftover column tiles, only sub-tile m = wave: 4*NT + NR units per
// wave, nothing discarded, all four waves equal.  The price is NR extra
// B-fragment loads per K-group (each used by one MFMA group instead of four)
// and one extra A fragment; both come from L2 / LDS with slack to spare.
template <int NT, int NR>
__device__ __forceinline__ void gemm_tile_split(
    const float *__restrict__ tile, int ld, const LayerDev &L, int t0, int wave,
    int lane, v4f (&acc)[4][NT], v4f (&accr)[NR]) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < NR; ++r) accr[r] = (v4f){0.f, 0.f, 0.f, 0.f};
  const v4f *__restrict__ wp = reinterpret_cast<const v4f *>(L.wp) + lane;
  const float *arow = tile + (lane & 15) * ld + 4 * (lane >> 4);
  const float *arow_mine = arow + wave * 16 * ld;
  const int qstride = L.nt * 64;
  const int kq = L.kq;
  constexpr int PF = PGNN_PF4;
  v4f a[PF][4], b[PF][NT], ar[PF], br[PF][NR];
  auto fetch = [&](int q, v4f (&fa)[4], v4f (&fb)[NT], v4f &fa_r,
                   v4f (&fb_r)[NR]) {
    if (q > kq - 1) q = kq - 1;
    const v4f *wq = wp + (size_t)q * qstride;
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = wq[(t0 + wave + 4 * j) * 64];
#pragma unroll
    for (int r = 0; r < NR; ++r) fb_r[r] = wq[(t0 + 4 * NT + r) * 64];
#pragma unroll
    for (int m = 0; m < 4; ++m)
      fa[m] = *reinterpret_cast<const v4f *>(arow + m * 16 * ld + 16 * q);
    fa_r = *reinterpret_cast<const v4f *>(arow_mine + 16 * q);
  };
  auto mma = [&](const v4f (&fa)[4], const v4f (&fb)[NT], const v4f &fa_r,
                 const v4f (&fb_r)[NR], int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][s], fb[j][s],
                                                           acc[m][j], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NR; ++r)
        accr[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa_r[s], fb_r[r][s],
                                                       accr[r], 0, 0, 0);
    }
  };
#pragma unroll
  for (int st = 0; st < PF; ++st) fetch(st, a[st], b[st], ar[st], br[st]);
  for (int q = 0; q < kq; q += PF) {
#pragma unroll
    for (int st = 0; st < PF; ++st) {
      if (q + st < kq) {  // wave-uniform
        mma(a[st], b[st], ar[st], br[st], 0, 4);
        fetch(q + st + PF, a[st], b[st], ar[st], br[st]);
      }
    }
  }
  // (Inline-asm weight loads with hand-placed s_waitcnt vmcnt(7) -- exact waits,
  // both stages truly in flight -- were tried here: +5 % in the bare-loop
  // micro-benchmark, but 1.5 % SLOWER in the fused kernel and the compiler,
  // unaware of the asynchronous register writes, broke a parity test.)
}

// activated value of accumulator `x` in column `col`
__device__ __forceinline__ float activate(const LayerDev &L, const float *bias,
                                          int col, float x) {
  x += bias[col];
  if (col >= L.relu_from) x = x > 0.0f ? x : 0.0f;
  return x;
}

template <int NT, int NR, bool TRANSPOSED>
__device__ __forceinline__ void store_split(float *__restrict__ out, int ldo,
                                            const LayerDev &L, int t0, int wave,
                                            int lane, const v4f (&acc)[4][NT],
                                            const v4f (&accr)[NR]) {
  if (TRANSPOSED)
    store_acc_T<4, NT>(out, L, t0, wave, lane, acc);
  else
    store_acc<4, NT>(out, ldo, L, t0, wave, lane, acc);
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
  constexpr int ROWS = 64, G = ROWS / 4, SWZ = 15;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int col = (t0 + 4 * NT + r) * 16 + (lane & 15);
    const int c = col - 16 * t0;
    v4f v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = activate(L, bias, col, accr[r][i]);
    if (TRANSPOSED) {
      const int g = 4 * wave + (lane >> 4);
      *reinterpret_cast<v4f *>(out + c * ROWS + ((g ^ (c & SWZ)) << 2)) = v;
    } else {
      float *o = out + (16 * wave + 4 * (lane >> 4)) * ldo + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i * ldo] = v[i];
    }
  }
  (void)G;
}

template <int NT, int NR, bool TRANSPOSED>
__device__ __forceinline__ void layer_pass_split(const float *in, int ld_in,
                                                 float *out, int ld_out,
                                                 const LayerDev &L, int t0,
                                                 int wave, int lane) {
  v4f acc[4][NT], accr[NR];
  gemm_tile_split<NT, NR>(in, ld_in, L, t0, wave, lane, acc, accr);
  __syncthreads();  // every wave is done reading `in` (in-place overwrite)
  store_split<NT, NR, TRANSPOSED>(out, ld_out, L, t0, wave, lane, acc, accr);
  __syncthreads();
}

// LDS float max that is correct for any sign mix (see atomic_max_f32)
__device__ __forceinline__ void lds_atomic_max_f32(float *addr, float v) {
  if (v >= 0.0f) {
    __hip_atomic_fetch_max((int *)addr, __float_as_int(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    __hip_atomic_fetch_min((unsigned int *)addr, __float_as_uint(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__device__ __forceinline__ void segfast_emit(const SegFast &sf, int col,
                                             float v) {
  if (sf.merge) v = fmaxf(v, sf.carry[col]);
  if (sf.defer) {
    sf.carry[col] = v;
  } else if (sf.whole) {
    sf.out_row[col] = v;
  } else {
    atomic_max_f32(sf.out_row + col, v + 0.0f);
  }
}

// Fast (one-segment) epilogue with the split: full tiles as in
// layer_pass_segmax_fast; of a leftover tile each wave holds only 16 of the 64
// rows, so its column maxima (pre-bias) meet in `part` (16*NR floats of LDS,
// preset to lowest()) through LDS atomics, and `segfast_finish_split` emits
// them after the workgroup's end-of-tile barrier.
template <int NT, int NR>
__device__ __forceinline__ void layer_pass_segmax_fast_split(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegFast &sf, float *part) {
  v4f acc[4][NT], accr[NR];
  gemm_tile_split<NT, NR>(in, ld_in, L, t0, wave, lane, acc, accr);
  const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float v = fmaxf(fmaxf(accr[r][0], accr[r][1]),
                    fmaxf(accr[r][2], accr[r][3]));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    if (lane < 16) lds_atomic_max_f32(part + r * 16 + lane, v + 0.0f);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float v = acc[0][j][0];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) v = fmaxf(v, acc[m][j][i]);
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    const int col = (t0 + wave + 4 * j) * 16 + (lane & 15);
    if (lane < 16) segfast_emit(sf, col, activate(L, bias, col, v));
  }
}

// after the end-of-tile barrier: threads 0 .. 16*NR-1 finish the leftover
// columns and re-arm `part`
template <int NT, int NR>
__device__ __forceinline__ void segfast_finish_split(const LayerDev &L, int t0,
                                                     const SegFast &sf,
                                                     float *part) {
  if (threadIdx.x < 16 * NR) {
    const float *bias = L.wp + (size_t)L.kq * L.nt * 256;
    const int col = (t0 + 4 * NT) * 16 + threadIdx.x;
    const float v = part[threadIdx.x];
    part[threadIdx.x] = kFloatLowest;
    segfast_emit(sf, col, activate(L, bias, col, v));
  }
}

// the split applies to 64-row tiles with 19 column tiles in the pass (C = 300)
__device__ __forceinline__ bool use_split(int msub, int tiles) {
  return msub == 4 && tiles == 19;
}

// fast one-segment pass; returns true when the split was used, in which case
// the caller runs segfast_finish_split<4, 3> after its end-of-tile barrier
template <int MSUB>
__device__ __forceinline__ bool layer_pass_segmax_fast_auto(
    const float *in, int ld_in, const LayerDev &L, int t0, int wave, int lane,
    const SegFast &sf, float *part) {
  int tiles = L.nt - t0;
  if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
  if constexpr (MSUB == 4) {
    if (use_split(MSUB, tiles)) {
      layer_pass_segmax_fast_split<4, 3>(in, ld_in, L, t0, wave, lane, sf,
                                         part);
      return true;
    }
  }
  layer_pass_segmax_fast_dispatch<MSUB>(in, ld_in, L, t0, wave, lane, sf);
  return false;
}

// ---------------------------------------------------------------------------
// Hidden layers of a narrow chain kept in REGISTERS (PointSetPooling's point
// MLP 4 -> 32 -> 64 -> 128 before its wide last layer).  The LDS-tile scheme
// above splits the N dimension over the four waves; with 2 / 4 / 8 column
// tiles that leaves waves idle and still pays two workgroup barriers and an
// LDS round trip per layer.  Here every wave owns 16 rows of the tile and
// computes the TRANSPOSED product  H'^T = W^T H^T : the weights are the MFMA
// A operand, the activations the B operand.  The 16x16x4 MFMA's C/D layout
// (lane (g, n), register r  <->  output feature 16t + 4g + r, row n) is
// exactly the B-operand layout of the next layer's K-group t, k-step r -- and
// the host-packed weight fragment packed[((q nt + t) 64 + lane) 4 + s] =
// W[16q + 4(lane>>4) + s][16t + (lane&15)] is exactly the A operand -- so a
// layer's accumulators, after bias + ReLU in place, ARE the next layer's
// operand registers: no barrier, no LDS, no shuffles between the layers, all
// four waves busy.  Every output element sees the same sequence of (K-group,
// k-step) MFMA updates as in gemm_tile, with the same four products each, so
// the values are those of the LDS path (tested bit for bit).  The price is
// that each wave streams ALL weights of these layers (42 KB for 4-32-64-128)
// instead of a quarter -- small next to the last layer's 156 KB per tile.
template <int KQ, int NT, int TBMAX = 8>
__device__ __forceinline__ void reg_layer(const LayerDev &L, int lane,
                                          const v4f (&in)[KQ], v4f (&out)[NT]) {
  // An opaque zero keeps the fragment addresses from being hoisted out of the
  // caller's tile loop: loop-invariant as they are, LLVM otherwise carries one
  // precomputed address per fragment (42+) across the whole loop and spills.
  int zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  const v4f *__restrict__ wp =
      reinterpret_cast<const v4f *>(L.wp) + lane + zero;
  const float *bias = L.wp + (size_t)KQ * NT * 256 + zero;
  // (+ zero: the per-column ReLU predicates below are loop-invariant too --
  // 56 SGPR pairs for 2 + 4 + 8 column tiles when hoisted)
  const int g4 = 4 * (lane >> 4) + zero;
  // column tiles in blocks of <= 8: two stages of 8 weight fragments (64
  // VGPRs) in flight, whatever the layer width
  constexpr int TB = NT < TBMAX ? NT : TBMAX;
  static_assert(NT % TB == 0, "column tiles come in blocks");
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TB) {
    v4f w[2][TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      out[t0 + j] = (v4f){0.f, 0.f, 0.f, 0.f};
      w[0][j] = wp[(size_t)(t0 + j) * 64];
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      if (q + 1 < KQ) {
#pragma unroll
        for (int j = 0; j < TB; ++j)
          w[(q + 1) & 1][j] = wp[((size_t)(q + 1) * NT + t0 + j) * 64];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < TB; ++j)
          out[t0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              w[q & 1][j][s], in[q][s], out[t0 + j], 0, 0, 0);
      // one K-group of weight prefetch in flight, no more: left alone, the
      // scheduler hoists the (address-independent) loads of ALL layers to the
      // top and spills
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
      const v4f b = *reinterpret_cast<const v4f *>(bias + 16 * t + g4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = out[t][r] + b[r];
        if (16 * t + g4 + r >= L.relu_from) v = v > 0.0f ? v : 0.0f;
        out[t][r] = v;
      }
    }
  }
}

// chain.l[li], chain.l[li+1], ... with NTS... column tiles each, then the rows
// go to the LDS tile (row-major, leading dimension ld) for the wide last layer
template <int KQ, int... NTS>
struct RegChain;

template <int KQ>
struct RegChain<KQ> {
  static __device__ __forceinline__ void run(const ChainDev &, int, int lane,
                                             const v4f (&in)[KQ],
                                             float *__restrict__ rows16,
                                             int ld) {
    // lane (g, n): row n, columns 16q + 4g .. 16q + 4g + 3 -- the mapping of
    // gemm_tile's A-fragment reads, conflict-free for ld = 8 mod 16
    float *o = rows16 + (lane & 15) * ld + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < KQ; ++q) *reinterpret_cast<v4f *>(o + 16 * q) = in[q];
  }
};

template <int KQ, int NT, int... REST>
struct RegChain<KQ, NT, REST...> {
  static __device__ __forceinline__ void run(const ChainDev &chain, int li,
                                             int lane, const v4f (&in)[KQ],
                                             float *__restrict__ rows16,
                                             int ld) {
    v4f out[NT];
    reg_layer<KQ, NT>(chain.l[li], lane, in, out);
    RegChain<NT, REST...>::run(chain, li + 1, lane, out, rows16, ld);
  }
};

// One pass (<= 320 output columns starting at column tile t0) of layer L:
// GEMM from `in`, barrier, activated store to `out` (may alias `in`), barrier.
template <int MSUB, int NT, bool TRANSPOSED, int NW = 4>
__device__ __forceinline__ void layer_pass(const float *in, int ld_in, float *out,
                                           int ld_out, const LayerDev &L, int t0,
                                           int wave, int lane, bool skip_gemm) {
  static_assert(!TRANSPOSED || NW == 4, "transposed stage is 4-wave only");
  v4f acc[MSUB][NT];
  if (!skip_gemm) {
    gemm_tile<MSUB, NT, NW>(in, ld_in, L, t0, wave, lane, acc);
  } else {  // ablation builds only
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[m][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();  // every wave is done reading `in` (in-place overwrite)
  if constexpr (TRANSPOSED)
    store_acc_T<MSUB, NT>(out, L, t0, wave, lane, acc);
  else
    store_acc<MSUB, NT, NW>(out, ld_out, L, t0, wave, lane, acc);
  __syncthreads();
}

// ---- layer passes of the 8-wave, 16-row K-row kernels ---------------------------
// A pass = wait for the first weight fragments (an L2 / HBM round trip),
// 19 K-groups of MFMAs, barrier, bias loads (another round trip), activated
// store, barrier.  With 16 rows there are only ~12k cycles of MFMA issue per
// pass to hide anything behind, and the two exposed round trips were a third
// of a 24k-cycle pass (tools/krow_timeline.py; ablation without weight loads:
// 20.7k).  Neither depends on the activations, so both are requested a pass
// EARLY: `krow_prefetch` asks for a layer's bias values and the weight
// fragments of its first PGNN_PF1 K-groups; a pass consumes the set requested
// for it and, right after its own K loop (before its barriers and store),
// requests the next pass's.  Same MFMA sequence per output element as
// gemm_tile / store_acc: bit-identical results.
constexpr int kKrowNT = 3;  // column tiles per wave: 20 tiles over 8 waves

// `b` doubles as the K loop's pipeline registers (stage st, column slot j): a
// pass consumes and refills it in place and leaves the NEXT pass's first
// stages in it -- no copies (a member-by-member copy into a local pipeline
// array is turned into wide vector loads from the struct, which then cannot
// be promoted out of scratch memory).
struct KrowPre {
  v4f b[PGNN_PF1][kKrowNT];
  float bias[kKrowNT];
};

__device__ __forceinline__ void krow_prefetch_w(const LayerDev &L, int t0,
                                                int wave, int lane,
                                                KrowPre &p) {
  const v4f *__restrict__ wp = reinterpret_cast<const v4f *>(L.wp) + lane;
  const int qstride = L.nt * 64;
#pragma unroll
  for (int j = 0; j < kKrowNT; ++j) {
    int t = t0 + wave + 8 * j;
    if (t > L.nt - 1) t = L.nt - 1;  // clamp: something valid, discarded later
#pragma unroll
    for (int st = 0; st < PGNN_PF1; ++st) {
      const int q = st < L.kq ? st : L.kq - 1;
      p.b[st][j] = wp[(size_t)q * qstride + t * 64];
    }
  }
}

__device__ __forceinline__ void krow_prefetch_bias(const LayerDev &L, int t0,
                                                   int wave, int lane,
                                                   KrowPre &p) {
  const float *__restrict__ bias = L.wp + (size_t)L.kq * L.nt * 256;
#pragma unroll
  for (int j = 0; j < kKrowNT; ++j) {
    int t = t0 + wave + 8 * j;
    if (t > L.nt - 1) t = L.nt - 1;
    p.bias[j] = bias[t * 16 + (lane & 15)];
  }
}

__device__ __forceinline__ void krow_prefetch(const LayerDev &L, int t0, int wave,
                                              int lane, KrowPre &p) {
  krow_prefetch_w(L, t0, wave, lane, p);
  krow_prefetch_bias(L, t0, wave, lane, p);
}

template <int NT>
__device__ __forceinline__ void layer_pass_krow(const float *in, int ld_in,
                                                float *out, int ld_out,
                                                const LayerDev &L, int t0,
                                                int wave, int lane, KrowPre &pre,
                                                bool has_next,
                                                const LayerDev next,
                                                int next_t0) {
  constexpr int PF = PGNN_PF1, NW = 8;
  static_assert(NT <= kKrowNT, "prefetch slots");
  if constexpr (NT == 0) {
    // a wave without a column tile in this pass (narrow layers): it only
    // keeps the workgroup's barriers and the prefetch chain going
    if (has_next) krow_prefetch_w(next, next_t0, wave, lane, pre);
    __syncthreads();
    if (has_next) krow_prefetch_bias(next, next_t0, wave, lane, pre);
    __syncthreads();
    return;
  } else {
  v4f acc[NT];
  int toff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    acc[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    int t = t0 + wave + NW * j;
    if (t > L.nt - 1) t = L.nt - 1;
    toff[j] = t * 64;
  }
  const v4f *__restrict__ wp = reinterpret_cast<const v4f *>(L.wp) + lane;
  const float *arow = in + (lane & 15) * ld_in + 4 * (lane >> 4);
  const int qstride = L.nt * 64;
  const int kq = L.kq;
  v4f a[PF];
#pragma unroll
  for (int st = 0; st < PF; ++st)
    a[st] = *reinterpret_cast<const v4f *>(arow + 16 * (st < kq ? st : kq - 1));
  int q = 0;
  for (; q + PF <= kq; q += PF) {
#pragma unroll
    for (int st = 0; st < PF; ++st) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              a[st][s], pre.b[st][j][s], acc[j], 0, 0, 0);
      int qn = q + st + PF;
      if (qn > kq - 1) qn = kq - 1;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        pre.b[st][j] = wp[(size_t)qn * qstride + toff[j]];
      a[st] = *reinterpret_cast<const v4f *>(arow + 16 * qn);
    }
  }
  const int rem = kq - q;  // < PF, wave-uniform
#pragma unroll
  for (int st = 0; st + 1 < PF; ++st) {
    if (st < rem) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              a[st][s], pre.b[st][j][s], acc[j], 0, 0, 0);
    }
  }
  // the next pass's first weight fragments: requested now, used two barriers on
  if (has_next) krow_prefetch_w(next, next_t0, wave, lane, pre);
  __syncthreads();  // every wave is done reading `in` (in-place overwrite)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + wave + NW * j;
    if (t < L.nt && t - t0 < kMaxTilesPerPass) {
      const int col = t * 16 + (lane & 15);
      const bool relu = col >= L.relu_from;
      const float bv = pre.bias[j];
      float *o = out + (4 * (lane >> 4)) * ld_out + (col - 16 * t0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][r] + bv;
        if (relu) v = v > 0.0f ? v : 0.0f;
        o[r * ld_out] = v;
      }
    }
  }
  // ... and its bias values, now that this pass's are used
  if (has_next) krow_prefetch_bias(next, next_t0, wave, lane, pre);
  __syncthreads();
  }
}

// one pass (<= 320 output columns from column tile t0) of layer L on the
// 16-row tile; `pre` holds what krow_prefetch requested for (L, t0) and comes
// back holding the request for (next, next_t0) when has_next.  (`next` by
// value: a pointer to an element of a by-value kernel argument would force the
// whole argument struct into scratch memory.)
__device__ __forceinline__ void krow_pass(const float *in, int ld_in, float *out,
                                          int ld_out, const LayerDev &L, int t0,
                                          int wave, int lane, KrowPre &pre,
                                          bool has_next, const LayerDev next,
                                          int next_t0) {
  int tiles = L.nt - t0;
  if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
  // column tiles of THIS wave (t0 + wave + 8 j < L.nt; wave-uniform): 19 tiles
  // are 3-3-3-2-2-2-2-2, i.e. 5-5-5-4 per SIMD.  (With ceil(tiles / 8) for
  // every wave the five waves without a third tile computed a clamped one and
  // threw it away: 6 tiles of MFMA issue per SIMD instead of 5.)
  const int ntw = tiles / 8 + (wave < tiles % 8 ? 1 : 0);
  switch (ntw) {
    case 0: layer_pass_krow<0>(in, ld_in, out, ld_out, L, t0, wave, lane, pre, has_next, next, next_t0); break;
    case 1: layer_pass_krow<1>(in, ld_in, out, ld_out, L, t0, wave, lane, pre, has_next, next, next_t0); break;
    case 2: layer_pass_krow<2>(in, ld_in, out, ld_out, L, t0, wave, lane, pre, has_next, next, next_t0); break;
    default: layer_pass_krow<3>(in, ld_in, out, ld_out, L, t0, wave, lane, pre, has_next, next, next_t0); break;
  }
}

template <int MSUB, bool TRANSPOSED, int NW = 4>
__device__ __forceinline__ void layer_pass_dispatch(const float *in, int ld_in,
                                                    float *out, int ld_out,
                                                    const LayerDev &L, int t0,
                                                    int wave, int lane,
                                                    bool skip_gemm = false) {
  int tiles = L.nt - t0;
  if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
  if constexpr (MSUB == 4 && NW == 4) {
    if (use_split(MSUB, tiles) && !skip_gemm) {
      layer_pass_split<4, 3, TRANSPOSED>(in, ld_in, out, ld_out, L, t0, wave,
                                         lane);
      return;
    }
  }
  const int ntw = (tiles + NW - 1) / NW;  // column tiles per wave (wave-uniform)
  switch (ntw) {
    case 1: layer_pass<MSUB, 1, TRANSPOSED, NW>(in, ld_in, out, ld_out, L, t0, wave, lane, skip_gemm); break;
    case 2: layer_pass<MSUB, 2, TRANSPOSED, NW>(in, ld_in, out, ld_out, L, t0, wave, lane, skip_gemm); break;
    case 3: layer_pass<MSUB, 3, TRANSPOSED, NW>(in, ld_in, out, ld_out, L, t0, wave, lane, skip_gemm); break;
    case 4: layer_pass<MSUB, 4, TRANSPOSED, NW>(in, ld_in, out, ld_out, L, t0, wave, lane, skip_gemm); break;
    default: layer_pass<MSUB, 5, TRANSPOSED, NW>(in, ld_in, out, ld_out, L, t0, wave, lane, skip_gemm); break;
  }
}

}  // namespace pgnn
