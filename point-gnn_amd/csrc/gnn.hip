// Fused GNN kernels (models/gnn.py): every kernel is "prologue -> MLP chain on
// an LDS-resident row tile (mlp_engine.h) -> epilogue", persistent over tiles.
//
//   prologue ROWS : rows of a [n, ld] matrix (optionally concat of two)
//            POOL : PointSetPooling edge features  [f(src), xyz(src)-xyz(kp(dst))]
//                   (gnn.py:256-267)
//            EDGE : GraphNetAutoCenter hidden vector ReLU(P[src] - Q[dst])
//                   (first edge layer factored per vertex, gnn.py:338-356)
//   epilogue ROWS : rows (+ residual) to HBM, coalesced
//            SEGMAX: scatter-max over runs of equal dst inside the tile
//                   (gnn.py:275-277, 362-365); runs wholly inside the tile are
//                   complete segments (sorted edges) and are stored plainly,
//                   boundary runs use float atomic-max.
// The E x C activations never reach HBM.
#include "mlp_engine.h"
#include "edge_ws.h"
#include "edge_ws_bf16.h"
#include "edge_ws_f16.h"
#include "pool_ws.h"
#include "pool_split.h"
#include "pool_ws_f16.h"

namespace pgnn {
int g_mlp_blocks_per_cu = 4;  // upper bound; LDS usually allows fewer
int g_edge_msub = 0;          // 0 = auto, else force 16*msub-row tiles
int g_pool_msub = 0;
int g_mlp_pool_pct = 12;  // share of the row tiles handed out dynamically
int g_ws_xcds = 8;        // edge_ws.h: row slices (8 = one per XCD, 1 = none)
int g_ws_prio = 1;        // edge_ws.h: raised wave priority outside the MFMA loop
int g_f16_pool = 1;       // 0: pgnn_point_set_pooling_f16x2_fwd declines; 2: its 64 -> 128 layer in fp32 (same-box A/Bs)
int g_b16_force = 0;      // tests: the split-bf16 edge kernel also for lists of a few tiles
int g_ws_pool_pct = 0;    // edge_ws.h / pool_ws.h: share of the tiles handed out
                          // dynamically (measured: a pool costs more in extra
                          // range boundaries than it returns; kept as a tested
                          // option for streams that share the GPU)
int g_ws_chunk = 2;       // edge_ws.h: pool chunk (16-row tiles)
int g_ws_balance = 1;     // edge_ws.h: chip-wide balanced workgroup counts of the
                          // column groups (EdgeWsArgs::balanced): 1 = for the
                          // split-bf16 kernel, 2 = for the fp32 kernel too,
                          // 0 = the same counts in every XCD slice
int g_ws_reserve = 0;     // CUs the weights-stationary kernels leave to the
                          // kernels of other streams (multiple of 8)
void *g_mlp_ts = nullptr;  // device buffer for per-tile timestamps (profiling)
int g_mlp_debug = 0;  // ablation mask (benchmarks only): 1 = no gather loads,
                      // 2 = no last-layer GEMM, 4 = no epilogue, 16 = print
                      // occupancy, 32 = no one-segment fast path, 64 = no
                      // prologue priority, 128 = no few-runs register epilogue,
                      // 512 = 4-wave kernel for small rows, 1024 = pooling's
                      // hidden layers through the LDS tile (not registers),
                      // 2048 = edge stage always on the LDS-tile kernel,
                      // 4096 = ... always on the weights-stationary kernel
                      // (edge_ws.h) when the layer shape allows,
                      // 8192 / 16384 = the same pair for the pooling stage
                      // (pool_ws.h)
}

namespace {
using namespace pgnn;

enum { PRO_ROWS = 0, PRO_POOL = 1, PRO_EDGE = 2,
       // PRO_POOL with the hidden layers in registers (mlp_engine.h RegChain):
       // hidden column tiles 2-4-8 = car's 32-64-128.  (ped's 32-64-128-256
       // chain would fit the same scheme, but its 512-wide last layer needs
       // the 32-row tile, where only two waves own rows.)
       PRO_POOL_R3 = 3 };
constexpr bool is_pool(int pro) { return pro == PRO_POOL || pro == PRO_POOL_R3; }
constexpr bool is_pool_reg(int pro) { return pro == PRO_POOL_R3; }
// tiles with at most this many runs of equal dst reduce in registers
// (layer_pass_segmax_runs); more runs go through the LDS stage
constexpr int kMaxRegisterRuns = 6;

struct RowsArgs {
  const float *x;
  int64_t ldx;
  int nx;
  const float *x2;
  int64_t ldx2;
  int nx2;
  const float *res;
  int64_t ldres;
  float *y;
  int64_t ldy;
  int res_gate;  // 0: y = rows + res; 1: y = res > 0 ? rows : 0 (ReluGrad of
                 // the layer below, fused into a backward dX pass)
  // taps of the chain form (rows_mlp_kernel<true>, the native training step):
  // the output of layer li < n - 1 is gated in the tile by gate[li] > 0
  // (nullable) and written to mid[li] (nullable), both [rows, 16 nt(li)]
  float *mid[PGNN_MAX_LAYERS - 1];
  const float *gate[PGNN_MAX_LAYERS - 1];
};
struct PoolArgs {
  const float *feat;
  int nfeat;
  const float *xyz;
  const int32_t *kp;
  const int32_t *edges;
  int reg_hidden;  // 0: hidden layers through the LDS tile; 1: 2-4-8 column
                   // tiles in registers (mlp_engine.h RegChain)
};
struct EdgeArgs {
  const float *P;
  const float *Q;
  int64_t ldpq;
  const int32_t *edges;
};
struct SegArgs {
  float *out;
  int64_t ldo;
  int num_segments;
  int sorted;
};

__host__ __device__ inline int ints_bytes(int rows) {
  return ((2 * rows + 2) * 4 + 15) / 16 * 16;  // dst[rows+2] + part[<=rows]
}

// Open segment carried from one tile to the next inside a workgroup's
// contiguous tile range (sorted edges only): id of the segment whose partial
// max sits in the LDS `carry` row, and whether its first edge is known to lie
// inside this workgroup's range.
struct CarryState {
  int id;           // -1: nothing carried
  int left_closed;
};

// The run left open at a tile's end (sorted ids, the workgroup owns the next
// tile too, the next edge has the same dst): it is carried instead of flushed.
// starts / myd as in consume_segmax; wave-uniform.
__device__ __forceinline__ CarryState carry_after(unsigned long long starts,
                                                  int myd, int d_before,
                                                  int d_after, const SegArgs &sa,
                                                  CarryState cs, bool keep_open) {
  CarryState out = {-1, 0};
  const int r = 63 - __builtin_clzll(starts);  // row 0 always starts a run
  const int d = __builtin_amdgcn_readlane(myd, r);
  if (sa.sorted && keep_open && d >= 0 && d < sa.num_segments && d_after == d) {
    bool left_closed = (r > 0) || (d_before != d);
    if (r == 0 && cs.id == d) left_closed = cs.left_closed != 0;
    out.id = d;
    out.left_closed = left_closed ? 1 : 0;
  }
  return out;
}

// Column-wise segmented max of stage[ROWS][ncols] keyed by dst[1..ROWS]
// (dst[0] / dst[ROWS+1] = id of the edge before / after the tile, -1 if none).
// Runs wholly inside the workgroup's range are complete segments: one plain
// coalesced row store.  A run that continues into the workgroup's next tile
// is folded into `carry` instead of being flushed, so a segment spanning many
// tiles costs one store in total; only the (at most two) runs that cross the
// workgroup's range boundary use float atomic-max.  Unsorted ids: every run
// is flushed atomically.  Control flow depends on dst[] only, so it is
// uniform across the workgroup and every thread returns the same state.
template <int ROWS>
__device__ __forceinline__ CarryState consume_segmax(
    const float *__restrict__ stT /* transposed+swizzled, see store_acc_T */,
    const int *__restrict__ dst, int col0, int ncols, const SegArgs &sa,
    float *__restrict__ carry, CarryState cs, bool keep_open) {
  constexpr int G = ROWS / 4, SWZ = (G < 16 ? G : 16) - 1;
  static_assert(ROWS <= 64, "one lane per tile row");
  // Runs of equal dst, found once per wave with a ballot (lane = tile row):
  // bit r of `starts` is set where row r opens a run.  Everything derived from
  // it (r, re, d, the closed/merge/defer flags) is wave-uniform and lives in
  // SGPRs; the serial LDS walk this replaces cost ~64 dependent reads a tile.
  const int lane = threadIdx.x & 63;
  const int myd = lane < ROWS ? dst[lane + 1] : -2;
  const int prevd = lane < ROWS ? dst[lane] : -2;
  const unsigned long long starts =
      __ballot(lane < ROWS && (lane == 0 || myd != prevd));
  const int d_before = dst[0], d_after = dst[ROWS + 1];
  // Columns outer (at most two trips for <= 320 columns), runs inner: the
  // thread's whole column (ROWS values) is fetched into registers with all
  // LDS reads in flight, then every run is reduced from registers; the
  // row-group selection compiles to scalar branches around v_max chains.
  for (int c = threadIdx.x; c < ncols; c += 256) {
    const float *col = stT + c * ROWS;
    const int sw = c & SWZ;
    v4f x[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      x[g] = *reinterpret_cast<const v4f *>(col + ((g ^ sw) << 2));
    unsigned long long rem = starts;
    while (rem) {
      const int r = __builtin_ctzll(rem);
      rem &= rem - 1;
      const int re = rem ? __builtin_ctzll(rem) : ROWS;
      const int d = __builtin_amdgcn_readlane(myd, r);
      if (d >= 0 && d < sa.num_segments) {
        bool left_closed = (r > 0) || (d_before != d);
        const bool merge = sa.sorted && r == 0 && cs.id == d;
        if (merge) left_closed = cs.left_closed != 0;
        const bool right_closed = (re < ROWS) || (d_after != d);
        const bool defer = sa.sorted && !right_closed && keep_open;
        const bool whole = sa.sorted && left_closed && right_closed;
        const int g0 = r >> 2, g1 = (re - 1) >> 2;
        float m = kFloatLowest;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g > g0 && g < g1) {
            m = fmaxf(fmaxf(m, x[g][0]), fmaxf(x[g][1], fmaxf(x[g][2], x[g][3])));
          } else if (g == g0 || g == g1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 4 * g + i;
              if (row >= r && row < re) m = fmaxf(m, x[g][i]);
            }
          }
        }
        if (merge) m = fmaxf(m, carry[col0 + c]);
        if (defer) {
          carry[col0 + c] = m;
        } else {
          float *o = sa.out + (int64_t)d * sa.ldo + col0 + c;
          if (whole)
            *o = m;
          else
            atomic_max_f32(o, m + 0.0f);
        }
      }
    }
  }
  // descriptor of the run left open at the tile's end (every thread, also
  // those without a column)
  return carry_after(starts, myd, d_before, d_after, sa, cs, keep_open);
}

template <int ROWS>
__device__ __forceinline__ void consume_rows(const float *__restrict__ stage,
                                             int ld, int64_t row0, int rows_valid,
                                             int col0, int ncols,
                                             const RowsArgs &ra) {
  const int total = rows_valid * ncols;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int r = idx / ncols, c = idx - r * ncols;
    float v = stage[r * ld + c];
    if (ra.res) {
      const float a = ra.res[(row0 + r) * ra.ldres + col0 + c];
      v = ra.res_gate ? (a > 0.0f ? v : 0.0f) : v + a;
    }
    ra.y[(row0 + r) * ra.ldy + col0 + c] = v;
  }
}

// ---- 16-row tiles of the 8-wave K-row kernels (512 threads) -------------------
// Thread (r = tid >> 5, c0 = tid & 31) owns columns c0, c0 + 32, ... of tile row
// r: ten steps cover the 320 columns a layer pass can produce.  Every phase
// that touches HBM issues ALL its loads before its first store.  The plain
// loops these replace (`y[i] = stage[i] + res[i]` over a strided index) compile
// to one load -> wait -> store round trip per element -- the compiler cannot
// move a load of `res` above a store to `y` it cannot prove disjoint -- and ten
// dependent round trips (~1.5 us each) per phase were most of a 28 us kernel
// whose MFMA work is 10 us (tools/krow_bench.py).
constexpr int kRowSteps = 10;  // 10 x 32 = 320 columns = kMaxTilesPerPass tiles

// y[row0 + r][col0 + c] = stage[r][c] (+ res[row0 + r][col0 + c]), c < ncols <= 320
__device__ __forceinline__ void consume_rows16(const float *__restrict__ stage,
                                               int ld, int64_t row0,
                                               int rows_valid, int col0,
                                               int ncols, const RowsArgs &ra) {
  const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
  const int64_t row = row0 + (r < rows_valid ? r : rows_valid - 1);
  float add[kRowSteps];
  if (ra.res) {
    const float *__restrict__ rr = ra.res + row * ra.ldres + col0;
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = c0 + 32 * j;
      add[j] = rr[c < ncols ? c : ncols - 1];  // unconditional, clamped
    }
  }
  float *__restrict__ yr = ra.y + row * ra.ldy + col0;
#pragma unroll
  for (int j = 0; j < kRowSteps; ++j) {
    const int c = c0 + 32 * j;
    if (r < rows_valid && c < ncols) {
      float v = stage[r * ld + c];
      if (ra.res) v = ra.res_gate ? (add[j] > 0.0f ? v : 0.0f) : v + add[j];
      yr[c] = v;
    }
  }
}

// A layer's output rows on their way through the chain: tile[r][c] = gate[row][c]
// > 0 ? tile[r][c] : 0 (when gated), mid[row][c] = tile[r][c] (when tapped);
// gate and mid are [rows, ncols].  Same thread layout and the same rule -- all
// loads of a round before its first store -- as consume_rows16.
__device__ __forceinline__ void tap_rows16(float *tile, int ld, int64_t row0,
                                           int rows_valid, int ncols,
                                           const float *__restrict__ gate,
                                           float *__restrict__ mid) {
  const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
  const int64_t row = row0 + (r < rows_valid ? r : rows_valid - 1);
  for (int cb = 0; cb < ncols; cb += 32 * kRowSteps) {
    const int left = ncols - cb;
    float g[kRowSteps];
    if (gate) {
      const float *__restrict__ gr = gate + row * ncols + cb;
#pragma unroll
      for (int j = 0; j < kRowSteps; ++j) {
        const int c = c0 + 32 * j;
        g[j] = gr[c < left ? c : left - 1];  // unconditional, clamped
      }
    }
    float *tr = tile + r * ld + cb;
    float *__restrict__ mr = mid ? mid + row * ncols + cb : nullptr;
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = c0 + 32 * j;
      if (c < left) {
        float v = tr[c];
        if (gate) {
          v = g[j] > 0.0f ? v : 0.0f;
          tr[c] = v;
        }
        if (mr && r < rows_valid) mr[c] = v;
      }
    }
  }
  if (gate) __syncthreads();  // the next pass reads the gated tile
}

// tile[r][c] = c < nx ? x[row0 + r][c] : 0 for c < kc (rows past the end: 0);
// 320 columns per round, all of a round's loads in flight before its LDS
// writes (wider inputs -- ped_cyl's 512 -- take a second round)
__device__ __forceinline__ void load_rows16(const float *__restrict__ x,
                                            int64_t ldx, int nx, float *tile,
                                            int ld0, int kc, int64_t row0,
                                            int rows_valid) {
  const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
  const float *__restrict__ xr =
      x + (row0 + (r < rows_valid ? r : rows_valid - 1)) * ldx;
  for (int cb = 0; cb < kc; cb += 32 * kRowSteps) {
    float v[kRowSteps];
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = cb + c0 + 32 * j;
      v[j] = xr[c < nx ? c : nx - 1];
    }
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = cb + c0 + 32 * j;
      if (c < kc) tile[r * ld0 + c] = (r < rows_valid && c < nx) ? v[j] : 0.0f;
    }
  }
}

template <int MSUB, int PRO>
__global__ __launch_bounds__(256, 2) void fused_mlp_kernel(
    ChainDev chain, int64_t n_rows, RowsArgs ra, PoolArgs pa, EdgeArgs ea,
    SegArgs sa, int stage_off /* floats from tile base; < 0: in place */,
    int dbg, long long *ts /* optional per-tile timestamps (profiling) */,
    int32_t *sched /* nullable: {next pool tile, finished workgroups}, zero */,
    int64_t n_static /* tiles below this index are partitioned statically */,
    const int32_t *n_dev /* nullable, capacity form: the row count lives on
                            the device, n_rows is its upper bound */,
    int dyn_pool_pct /* with n_dev: the pool's share of the tiles */) {
  constexpr int ROWS = 16 * MSUB;
  if (n_dev) {
    const int64_t nd = *n_dev;
    n_rows = nd < n_rows ? nd : n_rows;
  }
#ifndef PGNN_DIAG
  // the timing ablations (bits 1 / 2 / 4: drop the gather loads / the last
  // GEMM / the epilogue -- WRONG results) exist only in -DPGNN_DIAG builds
  // (tools/); here the bits are constant-folded away with their code
  dbg &= ~7;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *dst = reinterpret_cast<int *>(smem);
  // 16 * 3 floats behind dst[]: meeting point of the per-wave partial maxima of
  // the balanced split's leftover column tiles (mlp_engine.h), armed to lowest()
  float *part = reinterpret_cast<float *>(dst + ROWS + 2);
  if (MSUB == 4 && threadIdx.x < 48) part[threadIdx.x] = kFloatLowest;
  float *carry = reinterpret_cast<float *>(smem + ints_bytes(ROWS));
  float *tile = carry + 16 * chain.l[chain.n - 1].nt;
  float *stage = stage_off >= 0 ? tile + stage_off : tile;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_tiles = (n_rows + ROWS - 1) / ROWS;
  const int ld0 = lds_ld(16 * chain.l[0].kq);

  // Contiguous tile range per workgroup (remainder spread over the first ones)
  // over the first n_static tiles; the tiles behind them form a POOL that the
  // workgroups take one at a time through an atomic counter once their own
  // range is done.  The static part keeps the cheap contiguous-range machinery
  // (open segment carried from tile to tile, next tile's indices prefetched);
  // the pool is the slack that absorbs a late start: these kernels fill every
  // CU completely, so a workgroup of another stream's kernel that holds a CU
  // when the grid starts delays one persistent workgroup by its whole duration
  // -- with a purely static partition the kernel then ends that much later,
  // with a pool the others pick up the difference, a tile (~30 us) at a time.
  // A pool tile is a range of its own: nothing carried in or out, boundary
  // runs flushed atomically.  sched == nullptr: everything static.
  if (n_dev) {  // the host could not size the pool: launch_fused's rule, here
    if (sched && n_tiles >= 6 * (int64_t)gridDim.x)
      n_static = n_tiles - n_tiles * dyn_pool_pct / 100;
    else
      sched = nullptr;
  }
  if (!sched) n_static = n_tiles;
  const int64_t tq = n_static / gridDim.x, trem = n_static % gridDim.x;
  const int64_t tile_first =
      blockIdx.x * tq + (blockIdx.x < trem ? blockIdx.x : trem);
  int64_t tile_last = tile_first + tq + (blockIdx.x < trem ? 1 : 0);
  __shared__ int s_claim;
  CarryState cs = {-1, 0};
  // prefetched (src, dst[, keypoint]) of the next tile's rows (EDGE / POOL)
  int nxt_s = 0, nxt_d = -1, nxt_k = 0;
  bool fresh = true;  // first tile of a contiguous range: fetch its own indices
  int tile_seq = 0;
  for (int64_t tile_id = tile_first;; ++tile_id, ++tile_seq) {
    if (tile_id >= tile_last) {  // wave-uniform
      if (tile_id >= n_tiles || !sched) break;
      if (threadIdx.x == 0)
        s_claim = __hip_atomic_fetch_add(&sched[0], 1, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      tile_id = n_static + __builtin_amdgcn_readfirstlane(s_claim);
      if (tile_id >= n_tiles) break;
      tile_last = tile_id + 1;
      cs.id = -1;
      cs.left_closed = 0;
      fresh = true;
    }
    const int64_t row0 = tile_id * ROWS;
    const int rows_valid =
        (int)((n_rows - row0 < ROWS) ? (n_rows - row0) : ROWS);
    long long *tsp = nullptr;
    if (ts && tile_seq < 32 && threadIdx.x == 0) {
      tsp = ts + ((int64_t)blockIdx.x * 32 + tile_seq) * 8;
      tsp[0] = __builtin_readcyclecounter();
      tsp[3] = 0;
      tsp[6] = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
    }
    // ------------------------------------------------------------ prologue
    // The co-resident workgroup is usually streaming MFMAs on the same SIMDs;
    // measured (tools/tile_timeline.py): this short, latency-critical phase
    // takes 9.4k cycles alone but 22k (p90 52k) next to a GEMM phase at equal
    // priority.  Raising the wave priority here costs the partner a few
    // hundred issue slots and gets this workgroup back to the matrix pipe
    // sooner; priority returns to 0 before the MFMA loop.
    if (!(dbg & 64)) __builtin_amdgcn_s_setprio(3);
    if constexpr (PRO == PRO_ROWS) {
      const int kc = 16 * chain.l[0].kq;
      if (ra.nx2 == 0) {
        // 32 threads per row (8 rows per sweep), columns c0, c0 + 32, ...;
        // unconditional clamped loads, select on the value (rows_mlp_kernel)
        const int c0 = threadIdx.x & 31;
        for (int r = threadIdx.x >> 5; r < ROWS; r += 8) {
          const float *xr =
              ra.x + (row0 + (r < rows_valid ? r : rows_valid - 1)) * ra.ldx;
          for (int cb = 0; cb < kc; cb += 32 * 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = cb + c0 + 32 * j;
              v[j] = xr[c < ra.nx ? c : ra.nx - 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = cb + c0 + 32 * j;
              if (c < kc)
                tile[r * ld0 + c] = (r < rows_valid && c < ra.nx) ? v[j] : 0.0f;
            }
          }
        }
      } else {
      for (int idx = threadIdx.x; idx < ROWS * kc; idx += 256) {
        const int r = idx / kc, c = idx - r * kc;
        float v = 0.0f;
        if (r < rows_valid) {
          if (c < ra.nx)
            v = ra.x[(row0 + r) * ra.ldx + c];
          else if (c < ra.nx + ra.nx2)
            v = ra.x2[(row0 + r) * ra.ldx2 + (c - ra.nx)];
        }
        tile[r * ld0 + c] = v;
      }
      }
    } else if constexpr (is_pool_reg(PRO)) {
      // wave w owns rows 16w .. 16w+15; lane (g, n) holds input features
      // 4g .. 4g+3 of row 16w + n, which IS the B operand of the first layer
      const int g = lane >> 4;
      const int r = 16 * wave + (lane & 15);
      const int64_t e = row0 + r;
      if (fresh) {
        nxt_s = 0;
        nxt_d = -1;
        nxt_k = 0;
        if (e < n_rows) {
          nxt_s = pa.edges[2 * e];
          nxt_d = pa.edges[2 * e + 1];
          nxt_k = pa.kp[nxt_d];
        }
      }
      const int s_ = nxt_s, d_ = nxt_d, k_ = nxt_k;
      v4f x[1];
      x[0] = (v4f){0.f, 0.f, 0.f, 0.f};
      if (e < n_rows) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 4 * g + i;  // input column
          float v = 0.0f;
          if (c < pa.nfeat) {
            v = pa.feat[(int64_t)s_ * pa.nfeat + c];
          } else if (c < pa.nfeat + 3) {
            // points within a set use coordinates relative to its keypoint
            const int a = c - pa.nfeat;
            v = pa.xyz[3 * (int64_t)s_ + a] - pa.xyz[3 * (int64_t)k_ + a];
          }
          x[0][i] = v;
        }
      }
      if (g == 0) dst[r + 1] = e < n_rows ? d_ : -1;
      nxt_s = 0;
      nxt_d = -1;
      nxt_k = 0;
      if (e + ROWS < n_rows && tile_id + 1 < tile_last) {
        nxt_s = pa.edges[2 * (e + ROWS)];
        nxt_d = pa.edges[2 * (e + ROWS) + 1];
      }
      if (threadIdx.x == 64)
        dst[0] = row0 > 0 ? pa.edges[2 * (row0 - 1) + 1] : -1;
      if (threadIdx.x == 65)
        dst[ROWS + 1] =
            row0 + ROWS < n_rows ? pa.edges[2 * (row0 + ROWS) + 1] : -1;
      __builtin_amdgcn_s_setprio(0);  // the MFMA chain below is not a prologue
      const int ld_last = lds_ld(16 * chain.l[chain.n - 1].kq);
      float *rows16 = tile + 16 * wave * ld_last;
      // keep the compiler from hoisting the last layer's weight prefetch over
      // the chain (it spills otherwise: both want ~150 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
      RegChain<1, 2, 4, 8>::run(chain, 0, lane, x, rows16, ld_last);
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (PRO == PRO_POOL) {  // first layer has kq == 1 (host check)
      if (threadIdx.x < ROWS) {
        const int r = threadIdx.x;
        const int64_t e = row0 + r;
        float f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = 0.0f;
        // (src, dst, keypoint) of this row were requested during the previous
        // tile (nxt_*): one global round trip here instead of three
        if (fresh) {
          nxt_s = 0;
          nxt_d = -1;
          nxt_k = 0;
          if (e < n_rows) {
            nxt_s = pa.edges[2 * e];
            nxt_d = pa.edges[2 * e + 1];
            nxt_k = pa.kp[nxt_d];
          }
        }
        int d = -1;
        if (e < n_rows) {
          const int s = nxt_s;
          d = nxt_d;
          const int k = nxt_k;
#pragma unroll
          for (int i = 0; i < 13; ++i)
            if (i < pa.nfeat) f[i] = pa.feat[(int64_t)s * pa.nfeat + i];
          // points within a set use coordinates relative to its keypoint
          const float rx = pa.xyz[3 * (int64_t)s] - pa.xyz[3 * (int64_t)k];
          const float ry = pa.xyz[3 * (int64_t)s + 1] - pa.xyz[3 * (int64_t)k + 1];
          const float rz = pa.xyz[3 * (int64_t)s + 2] - pa.xyz[3 * (int64_t)k + 2];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i == pa.nfeat) f[i] = rx;
            if (i == pa.nfeat + 1) f[i] = ry;
            if (i == pa.nfeat + 2) f[i] = rz;
          }
        }
        dst[r + 1] = d;
#pragma unroll
        for (int c = 0; c < 16; ++c) tile[r * ld0 + c] = f[c];
        nxt_s = 0;
        nxt_d = -1;
        nxt_k = 0;
        if (e + ROWS < n_rows && tile_id + 1 < tile_last) {
          nxt_s = pa.edges[2 * (e + ROWS)];
          nxt_d = pa.edges[2 * (e + ROWS) + 1];
        }
      }
      if (threadIdx.x == 64)
        dst[0] = row0 > 0 ? pa.edges[2 * (row0 - 1) + 1] : -1;
      if (threadIdx.x == 65)
        dst[ROWS + 1] =
            row0 + ROWS < n_rows ? pa.edges[2 * (row0 + ROWS) + 1] : -1;
    } else {  // PRO_EDGE
      // Row-per-wave gather with scalar row addressing: a wave owns RPW rows;
      // src/dst of a row are wave-uniform (readlane), so the row base lives in
      // SGPRs and each lane only adds its fixed 16-byte column offset -- a few
      // instructions per row instead of per-element index arithmetic (the tile
      // timeline showed the prologue to be instruction-issue bound while the
      // co-resident workgroup streams MFMAs).  Full 64-float4 column blocks of
      // all RPW rows are requested first (the accumulators are not live yet, so
      // ~160 VGPRs of loads can be in flight); a tail of <= 16 float4 per row
      // (76 = 64 + 12 for C = 300) is fetched four rows per instruction.
      constexpr int RPW = ROWS / 4;
      const int ldv4 = (int)(ea.ldpq >> 2);
      const v4f *__restrict__ P4 = reinterpret_cast<const v4f *>(ea.P);
      const v4f *__restrict__ Q4 = reinterpret_cast<const v4f *>(ea.Q);
      const int64_t ebase = row0 + wave * RPW;
      // (src, dst) of this tile's rows were requested one tile ago (they sit
      // in nxt_*), so the row gathers below start without a dependent index
      // round trip; the next tile's pair is requested now and lands during
      // this tile's gather + GEMM.
      if (fresh) {
        nxt_s = 0;
        nxt_d = -1;
        if (lane < RPW && ebase + lane < n_rows) {
          nxt_s = ea.edges[2 * (ebase + lane)];
          nxt_d = ea.edges[2 * (ebase + lane) + 1];
        }
      }
      const int my_s = nxt_s, my_d = nxt_d;
      nxt_s = 0;
      nxt_d = -1;
      if (lane < RPW && ebase + ROWS + lane < n_rows &&
          tile_id + 1 < tile_last) {
        nxt_s = ea.edges[2 * (ebase + ROWS + lane)];
        nxt_d = ea.edges[2 * (ebase + ROWS + lane) + 1];
      }
      if (lane < RPW) dst[wave * RPW + lane + 1] = my_d;
      if (threadIdx.x == 0)
        dst[0] = row0 > 0 ? ea.edges[2 * (row0 - 1) + 1] : -1;
      if (threadIdx.x == 64)
        dst[ROWS + 1] =
            row0 + ROWS < n_rows ? ea.edges[2 * (row0 + ROWS) + 1] : -1;
      if (tsp) {
        tsp[3] = __builtin_readcyclecounter();  // index loads issued
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        tsp[4] = __builtin_readcyclecounter();  // ... and landed
      }
      const bool no_loads = (dbg & 1) != 0;
      const int nfull = ldv4 >> 6, tail = ldv4 & 63;
      float *tbase = tile + (wave * RPW) * ld0;
      // tail columns (<= 16 float4 per row), four rows per instruction: lanes
      // 16*sub .. 16*sub+tail-1 serve row 4*r4+sub.  Requested first so they
      // are in flight together with the full column blocks.
      const bool tail4 = tail > 0 && tail <= 16;
      const int sub = lane >> 4, tl = lane & 15;
      const int c4t = tail4 ? 64 * nfull + (tl < tail ? tl : 0) : 0;
      v4f pt[RPW / 4], qt[RPW / 4];
      int ddt[RPW / 4];
#pragma unroll
      for (int r4 = 0; r4 < RPW / 4; ++r4) {
        const int s_ = __shfl(my_s, 4 * r4 + sub);
        ddt[r4] = __shfl(my_d, 4 * r4 + sub);
        const int d_ = ddt[r4] < 0 ? 0 : ddt[r4];
        pt[r4] = P4[(int64_t)s_ * ldv4 + c4t];
        qt[r4] = Q4[(int64_t)d_ * ldv4 + c4t];
      }
      for (int fb = 0; fb < nfull; ++fb) {
        v4f p[RPW], q[RPW];
        const int c4 = 64 * fb + lane;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const int s_ = __builtin_amdgcn_readlane(my_s, r);
          int d_ = __builtin_amdgcn_readlane(my_d, r);
          d_ = d_ < 0 ? 0 : d_;
          p[r] = P4[(int64_t)s_ * ldv4 + c4];
          q[r] = Q4[(int64_t)d_ * ldv4 + c4];
        }
        // keep all 2*RPW loads in flight: without this fence hipcc software-
        // pipelines the two loops at depth 1 with a vmcnt(0) per row pair,
        // i.e. RPW serial L2 round trips (seen in the .s; 9k cycles per tile)
        __builtin_amdgcn_sched_barrier(0);
        // rows past the end (dst < 0) gathered row 0 of P and Q: finite values
        // in MFMA rows that no segment ever reads, so no validity select here
        // (v_sub + v_max per element; measured equal to the select form)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          v4f h;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            h[i] = no_loads ? 0.0f : fmaxf(p[r][i] - q[r][i], 0.0f);
          *reinterpret_cast<v4f *>(tbase + r * ld0 + 4 * c4) = h;
        }
      }
      if (tail4) {
        const int c4 = c4t;
#pragma unroll
        for (int r4 = 0; r4 < RPW / 4; ++r4) {
          v4f h;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            h[i] = no_loads ? 0.0f : fmaxf(pt[r4][i] - qt[r4][i], 0.0f);
          if (tl < tail)
            *reinterpret_cast<v4f *>(tbase + (4 * r4 + sub) * ld0 + 4 * c4) = h;
        }
      } else if (tail > 16) {
        const int c4 = 64 * nfull + (lane < tail ? lane : 0);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          const int s_ = __builtin_amdgcn_readlane(my_s, r);
          const int dr = __builtin_amdgcn_readlane(my_d, r);
          const int d_ = dr < 0 ? 0 : dr;
          const v4f p = P4[(int64_t)s_ * ldv4 + c4];
          const v4f q = Q4[(int64_t)d_ * ldv4 + c4];
          v4f h;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = p[i] - q[i];
            h[i] = (dr >= 0 && !no_loads && t > 0.0f) ? t : 0.0f;
          }
          if (lane < tail)
            *reinterpret_cast<v4f *>(tbase + r * ld0 + 4 * c4) = h;
        }
      }
    }
    if (tsp) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      tsp[5] = __builtin_readcyclecounter();  // wave 0 finished its rows
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    fresh = false;
    if (is_pool(PRO) && (threadIdx.x < ROWS || is_pool_reg(PRO)) && nxt_d >= 0)
      nxt_k = pa.kp[nxt_d];  // second level of the next tile's index chain
    if (tsp) tsp[1] = __builtin_readcyclecounter();
    // ------------------------------------------------------------ hidden layers
    constexpr bool hidden_done = is_pool_reg(PRO);
    for (int li = 0; li + 1 < chain.n && !hidden_done; ++li) {
      const LayerDev &L = chain.l[li];
      layer_pass_dispatch<MSUB, false>(tile, lds_ld(16 * L.kq), tile,
                                       lds_ld(16 * L.nt), L, 0, wave, lane);
    }
    if (tsp) tsp[7] = __builtin_readcyclecounter();  // hidden layers done
    // ------------------------------------------------------------ last layer
    {
      const LayerDev &L = chain.l[chain.n - 1];
      const int ld_in = lds_ld(16 * L.kq);
      // runs of equal dst among the tile's rows (lane = row), once per wave
      unsigned long long starts = 1;
      int myd = -2;
      if (PRO != PRO_ROWS) {
        myd = lane < ROWS ? dst[lane + 1] : -2;
        const int prevd = lane < ROWS ? dst[lane] : -2;
        starts = __ballot(lane < ROWS && (lane == 0 || myd != prevd));
      }
      for (int t0 = 0; t0 < L.nt; t0 += kMaxTilesPerPass) {
        int tiles = L.nt - t0;
        if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
        const int ncols = 16 * tiles;
        const int ld_st = lds_ld(ncols);
        SegFast sf;
        bool finish_split = false;
        if (PRO == PRO_ROWS) {
          layer_pass_dispatch<MSUB, false>(tile, ld_in, stage, ld_st, L, t0, wave,
                                           lane, (dbg & 2) != 0);
          consume_rows<ROWS>(stage, ld_st, row0, rows_valid, 16 * t0, ncols, ra);
        } else if (sa.sorted && !(dbg & (2 | 4 | 32)) && dst[1] >= 0 &&
                   dst[1] < sa.num_segments && dst[1] == dst[ROWS]) {
          // whole tile = one run of one segment: reduce in registers
          if (tsp && is_pool(PRO)) tsp[3] = 1;
          const int d = dst[1];
          const bool merge = cs.id == d;
          const bool left_closed = merge ? cs.left_closed != 0 : dst[0] != d;
          const bool right_closed = dst[ROWS + 1] != d;
          const bool defer = !right_closed && tile_id + 1 < tile_last;
          sf.carry = carry;
          sf.out_row = sa.out + (int64_t)d * sa.ldo;
          sf.merge = merge;
          sf.defer = defer;
          sf.whole = left_closed && right_closed;
          finish_split = layer_pass_segmax_fast_auto<MSUB>(
              tile, ld_in, L, t0, wave, lane, sf, part);
          if (t0 + kMaxTilesPerPass >= L.nt) {
            cs.id = defer ? d : -1;
            cs.left_closed = left_closed ? 1 : 0;
          }
        } else if (sa.sorted && !(dbg & (2 | 4 | 32 | 128)) &&
                   __builtin_popcountll(starts) <= kMaxRegisterRuns) {
          // a few runs: masked column maxima straight from the accumulators
          if (tsp && is_pool(PRO)) tsp[3] = 2;
          SegRuns sr;
          sr.starts = starts;
          sr.myd = myd;
          sr.d_before = dst[0];
          sr.d_after = dst[ROWS + 1];
          sr.carry_id = cs.id;
          sr.carry_left_closed = cs.left_closed;
          sr.keep_open = tile_id + 1 < tile_last;
          sr.sorted = 1;
          sr.carry = carry;
          sr.out = sa.out;
          sr.ldo = sa.ldo;
          sr.num_segments = sa.num_segments;
          layer_pass_segmax_runs_dispatch<MSUB>(tile, ld_in, L, t0, wave, lane,
                                                sr);
          if (t0 + kMaxTilesPerPass >= L.nt)
            cs = carry_after(starts, myd, sr.d_before, sr.d_after, sa, cs,
                             sr.keep_open != 0);
        } else {
          layer_pass_dispatch<MSUB, true>(tile, ld_in, stage, ld_st, L, t0, wave,
                                          lane, (dbg & 2) != 0);
          if (!(dbg & 4)) {
            const CarryState nxt = consume_segmax<ROWS>(
                stage, dst, 16 * t0, ncols, sa, carry, cs,
                tile_id + 1 < tile_last);
            if (t0 + kMaxTilesPerPass >= L.nt) cs = nxt;  // after the last pass
          }
        }
        // LDS-only barrier: the tile buffer may be overwritten by the next
        // prologue, but the global stores / atomics above need not drain
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (MSUB == 4) {
          // leftover column tiles of the balanced split: every wave has added
          // its 16 rows to `part`; emit them (next touch of carry / part is
          // behind the next prologue's barrier)
          if (finish_split) segfast_finish_split<4, 3>(L, t0, sf, part);
        }
        if (tsp) tsp[2] = __builtin_readcyclecounter();
      }
    }
  }
  if (sched && threadIdx.x == 0) {
    // the last workgroup to get here re-arms the two counters: every claim of
    // this launch was consumed before its workgroup counted itself done
    const int done = __hip_atomic_fetch_add(&sched[1], 1, __ATOMIC_ACQ_REL,
                                            __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(&sched[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sched[1], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void offset_apply_kernel(const float *__restrict__ xyz,
                                    const float *__restrict__ delta,
                                    int64_t ld_delta, int64_t n,
                                    const float *__restrict__ wx,
                                    float *__restrict__ xyz_out,
                                    float *__restrict__ Q, int64_t ld_q) {
  const int64_t total = n * ld_q;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ld_q;
    const int c = (int)(idx - r * ld_q);
    float x0 = xyz[3 * r], x1 = xyz[3 * r + 1], x2 = xyz[3 * r + 2];
    if (delta) {
      x0 = x0 + delta[r * ld_delta];
      x1 = x1 + delta[r * ld_delta + 1];
      x2 = x2 + delta[r * ld_delta + 2];
    }
    if (c < 3 && xyz_out) xyz_out[3 * r + c] = c == 0 ? x0 : (c == 1 ? x1 : x2);
    Q[idx] = (x0 * wx[c] + x1 * wx[ld_q + c]) + x2 * wx[2 * ld_q + c];
  }
}

// ---- host side -----------------------------------------------------------------
struct Plan {
  ChainDev chain;
  int tile_floats_per_row;  // max ld over in-place activations
  int stage_cols;           // 0: last layer in place; else staged pass width
};

int make_plan(const pgnn_fc_layer *layers, int32_t n_layers, int first_k,
              Plan &p) {
  PGNN_REQUIRE(layers && n_layers >= 1 && n_layers <= PGNN_MAX_LAYERS,
               PGNN_E_INVALID, "mlp: 1..PGNN_MAX_LAYERS layers required");
  p.chain.n = n_layers;
  int prev_nt = -1;
  int max_ld = 0;
  for (int i = 0; i < n_layers; ++i) {
    const pgnn_fc_layer &l = layers[i];
    PGNN_REQUIRE(l.packed && l.k_in > 0 && l.n_out > 0, PGNN_E_INVALID,
                 "mlp: bad layer");
    LayerDev &d = p.chain.l[i];
    d.wp = l.packed;
    d.kq = (l.k_in + 15) / 16;
    d.nt = (l.n_out + 15) / 16;
    d.relu_from = l.relu_from;
    if (i == 0) {
      PGNN_REQUIRE(first_k <= l.k_in, PGNN_E_INVALID,
                   "mlp: input wider than first layer");
    } else {
      PGNN_REQUIRE(d.kq == prev_nt, PGNN_E_INVALID,
                   "mlp: layer widths do not chain");
    }
    if (i + 1 < n_layers)
      PGNN_REQUIRE(d.nt <= kMaxTilesPerPass, PGNN_E_UNSUPPORTED,
                   "mlp: hidden width > 320 not supported");
    prev_nt = d.nt;
    const int ld_in = lds_ld(16 * d.kq);
    if (ld_in > max_ld) max_ld = ld_in;
    if (i + 1 < n_layers || d.nt <= kMaxTilesPerPass) {
      const int ld_out = lds_ld(16 * d.nt);
      if (ld_out > max_ld) max_ld = ld_out;
    }
  }
  p.tile_floats_per_row = max_ld;
  p.stage_cols =
      p.chain.l[n_layers - 1].nt > kMaxTilesPerPass ? 16 * kMaxTilesPerPass : 0;
  return 0;
}

size_t plan_lds_bytes(const Plan &p, int rows) {
  size_t floats = (size_t)rows * p.tile_floats_per_row;
  floats += 16 * (size_t)p.chain.l[p.chain.n - 1].nt;  // carry row
  if (p.stage_cols) floats += (size_t)rows * lds_ld(p.stage_cols);
  return ints_bytes(rows) + floats * 4;
}

template <int MSUB, int PRO>
int launch_fused(const Plan &p, int64_t n_rows, const RowsArgs &ra,
                 const PoolArgs &pa, const EdgeArgs &ea, const SegArgs &sa,
                 hipStream_t stream, int32_t *sched = nullptr,
                 const int32_t *n_dev = nullptr) {
  // n_dev: capacity form -- n_rows sizes the grid, the kernel reads the count
  constexpr int ROWS = 16 * MSUB;
  const size_t lds = plan_lds_bytes(p, ROWS);
  PGNN_REQUIRE(lds <= 160 * 1024, PGNN_E_UNSUPPORTED,
               "mlp: layer too wide for the LDS tile");
  auto kern = fused_mlp_kernel<MSUB, PRO>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  const int64_t n_tiles = (n_rows + ROWS - 1) / ROWS;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > g_mlp_blocks_per_cu) per_cu = g_mlp_blocks_per_cu;
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)stream_cu_count(stream) * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  // tile pool (see the kernel): g_mlp_pool_pct % of the tiles, when every
  // workgroup still keeps a static range of a few tiles
  int64_t n_static = n_tiles;
  if (g_mlp_ts || g_mlp_pool_pct <= 0 || (!n_dev && n_tiles < 6 * grid))
    sched = nullptr;
  if (sched) n_static = n_tiles - n_tiles * g_mlp_pool_pct / 100;
  PGNN_HIP((hipError_t)arm_sched(sched, stream));
  const int stage_off = p.stage_cols ? ROWS * p.tile_floats_per_row : -1;
  if (g_mlp_debug & 16) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &nb, reinterpret_cast<const void *>(kern), 256, lds);
    fprintf(stderr, "[pgnn] fused<%d,%d> lds=%zu per_cu=%d occupancy_api=%d (%d) grid=%lld tiles=%lld\n",
            MSUB, PRO, lds, per_cu, nb, (int)e, (long long)grid, (long long)n_tiles);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p.chain,
                     n_rows, ra, pa, ea, sa, stage_off, g_mlp_debug,
                     (long long *)g_mlp_ts, sched, n_static, n_dev,
                     g_mlp_pool_pct);
  PGNN_HIP(hipGetLastError());
  return 0;
}

// Plain row MLP (PRO_ROWS) for SMALL row counts: 16-row tiles and EIGHT waves
// per workgroup.  With K ~ 3k vertices there are fewer tiles than CUs, so the
// 4-wave kernel leaves one wave per SIMD running a serial chain of
// (L2 load -> 20 MFMAs) groups with nothing to hide the latency behind; eight
// waves split the output columns 8 ways (<= 3 column tiles each) and give
// every SIMD two waves.
constexpr int kRowsWaves = 8;  // (16 waves: 128 VGPRs, spills, 13 % slower)

template <bool TAPS>
__global__ __launch_bounds__(64 * kRowsWaves) void rows_mlp_kernel(
    ChainDev chain, int64_t n_rows, RowsArgs ra, int stage_off,
    const int32_t *n_dev /* nullable: capacity form, see fused_mlp_kernel */) {
  constexpr int ROWS = 16, NW = kRowsWaves;
  if (n_dev) {
    const int64_t nd = *n_dev;
    n_rows = nd < n_rows ? nd : n_rows;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);
  float *stage = stage_off >= 0 ? tile + stage_off : tile;
  const int lane_id = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_tiles = (n_rows + ROWS - 1) / ROWS;
  const int ld0 = lds_ld(16 * chain.l[0].kq);
  for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    // (an opaque per-tile copy of the lane id: the per-lane weight addresses
    // of every layer are loop invariant, and hoisted out of the tile loop they
    // cost ~100 VGPRs -- spills in the capacity-form kernels)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_id));
    const int64_t row0 = tile_id * ROWS;
    const int rows_valid =
        (int)((n_rows - row0 < ROWS) ? (n_rows - row0) : ROWS);
    const int kc = 16 * chain.l[0].kq;
    // the first layer's weights and bias travel while the tile is loaded
    KrowPre pre;
    krow_prefetch(chain.l[0], 0, wave, lane, pre);
    if (ra.nx2 == 0) {
      // 32 threads per row, columns c0, c0 + 32, ...: every load is
      // UNCONDITIONAL (clamped row / column, select on the value) -- a load
      // under a predicate is waited for on the spot -- and all of them are in
      // flight before the first LDS write
      static_assert(64 * NW == 32 * ROWS, "one 32-thread group per row");
      load_rows16(ra.x, ra.ldx, ra.nx, tile, ld0, kc, row0, rows_valid);
    } else {
    for (int idx = threadIdx.x; idx < ROWS * kc; idx += 64 * NW) {
      const int r = idx / kc, c = idx - r * kc;
      float v = 0.0f;
      if (r < rows_valid) {
        if (c < ra.nx)
          v = ra.x[(row0 + r) * ra.ldx + c];
        else if (c < ra.nx + ra.nx2)
          v = ra.x2[(row0 + r) * ra.ldx2 + (c - ra.nx)];
      }
      tile[r * ld0 + c] = v;
    }
    }
    __syncthreads();
    for (int li = 0; li + 1 < chain.n; ++li) {
      const LayerDev &L = chain.l[li];
      krow_pass(tile, lds_ld(16 * L.kq), tile, lds_ld(16 * L.nt), L, 0, wave,
                lane, pre, true, chain.l[li + 1], 0);
      if constexpr (TAPS)
        tap_rows16(tile, lds_ld(16 * L.nt), row0, rows_valid, 16 * L.nt,
                   ra.gate[li], ra.mid[li]);
    }
    const LayerDev &L = chain.l[chain.n - 1];
    const int ld_in = lds_ld(16 * L.kq);
    for (int t0 = 0; t0 < L.nt; t0 += kMaxTilesPerPass) {
      int tiles = L.nt - t0;
      if (tiles > kMaxTilesPerPass) tiles = kMaxTilesPerPass;
      const int ncols = 16 * tiles;
      const int ld_st = lds_ld(ncols);
      const bool more = t0 + kMaxTilesPerPass < L.nt;
      krow_pass(tile, ld_in, stage, ld_st, L, t0, wave, lane, pre, more, L,
                t0 + kMaxTilesPerPass);
      consume_rows16(stage, ld_st, row0, rows_valid, 16 * t0, ncols, ra);
      __syncthreads();
    }
  }
}

template <bool TAPS = false>
int launch_rows8(const Plan &p, int64_t n_rows, const RowsArgs &ra,
                 hipStream_t stream, const int32_t *n_dev = nullptr,
                 int64_t grid_tiles = 0 /* capacity form: workgroups to
                     launch (the kernel strides over the tiles) */) {
  const size_t lds = plan_lds_bytes(p, 16);
  PGNN_REQUIRE(lds <= 160 * 1024, PGNN_E_UNSUPPORTED,
               "mlp: layer too wide for the LDS tile");
  auto kern = rows_mlp_kernel<TAPS>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  const int64_t n_tiles = (n_rows + 15) / 16;
  const int stage_off = p.stage_cols ? 16 * p.tile_floats_per_row : -1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(grid_tiles > 0 ? grid_tiles : n_tiles)),
                     dim3(64 * kRowsWaves), lds,
                     stream, p.chain, n_rows, ra, stage_off, n_dev);
  PGNN_HIP(hipGetLastError());
  return 0;
}

void ws_balance(EdgeWsArgs &a, int cus, const double *cost);  // below

// Weights-stationary edge kernel (edge_ws.h): one workgroup per CU, the column
// tiles in groups that fit the LDS, the 16-row tiles in one slice per XCD.
// ROWS: `ea.P` holds one ready input row per edge (pool_split.h).
template <int KQ, int NTMAX, bool ROWS = false>
int launch_edge_ws(const LayerDev &L, const EdgeArgs &ea, int64_t n_edges,
                   const SegArgs &sa, int cus, int32_t *sched,
                   hipStream_t stream, float *rows_out = nullptr,
                   int64_t ld_rows = 0, float *h1_out = nullptr,
                   const int32_t *n_dev = nullptr) {
  EdgeWsArgs a = {};
  a.n_dev = n_dev;
  a.rows_out = rows_out;
  a.ld_rows = ld_rows;
  a.h1_out = h1_out;
  a.P = ea.P;
  a.Q = ea.Q;
  a.ldv4 = (int)(ea.ldpq >> 2);
  a.edges = ea.edges;
  a.n_edges = n_edges;
  a.wp = L.wp;
  a.nt = L.nt;
  a.relu_from = L.relu_from;
  a.out = sa.out;
  a.ldo = sa.ldo;
  a.num_segments = sa.num_segments;
  a.sorted = sa.sorted;
  a.xcds = (g_ws_xcds >= 1 && cus % g_ws_xcds == 0) ? g_ws_xcds : 8;
  a.prio = g_ws_prio;
  a.ts = (long long *)g_mlp_ts;
  a.sched = (g_ws_pool_pct > 0 && a.xcds <= kWsMaxSlices) ? sched : nullptr;
  PGNN_HIP((hipError_t)arm_sched(a.sched, stream));
  a.pool_pct = g_ws_pool_pct;
  a.chunk = g_ws_chunk;
  a.groups = (L.nt + NTMAX - 1) / NTMAX;
  const int per_slice = cus / a.xcds;
  PGNN_REQUIRE(a.groups <= kWsMaxGroups && per_slice >= a.groups,
               PGNN_E_UNSUPPORTED, "edge_ws: too few CUs for the column groups");
  // column tiles as evenly as possible (C = 300: 7/6/6, C = 256: 8/8) and the
  // slice's workgroups in proportion (largest remainder)
  const int base = L.nt / a.groups, extra = L.nt % a.groups;
  int size[kWsMaxGroups], cnt[kWsMaxGroups], frac[kWsMaxGroups], used = 0;
  a.tile0[0] = 0;
  for (int g = 0; g < a.groups; ++g) {
    size[g] = base + (g < extra ? 1 : 0);
    a.tile0[g + 1] = a.tile0[g] + size[g];
    cnt[g] = per_slice * size[g] / L.nt;
    if (cnt[g] < 1) cnt[g] = 1;
    frac[g] = per_slice * size[g] % L.nt;
    used += cnt[g];
  }
  PGNN_REQUIRE(base >= NTMAX - 1 && base + (extra ? 1 : 0) <= NTMAX &&
                   used <= per_slice,
               PGNN_E_UNSUPPORTED, "edge_ws: column tiles do not group");
  while (used < per_slice) {
    int best = 0;
    for (int g = 1; g < a.groups; ++g)
      if (frac[g] > frac[best]) best = g;
    ++cnt[best];
    frac[best] = -1;
    ++used;
  }
  a.wg0[0] = 0;
  for (int g = 0; g < a.groups; ++g) a.wg0[g + 1] = a.wg0[g] + cnt[g];
  {
    // relative cost of a row tile: 4 KQ MFMAs of 32 cycles per column tile +
    // ~1.9k cycles of gather VALU / running max per tile (tools/ws_timeline.py)
    double cost[kWsMaxGroups];
    for (int g = 0; g < a.groups; ++g) cost[g] = 128.0 * KQ * size[g] + 1900.0;
    // (measured on the fp32 kernel: 977 vs 971 us, no gain -- its 12/10/10
    // imbalance is smaller than the noise between its XCDs; the split-bf16
    // kernel, 9/8/8/7 for 5/5/5/4 tiles, gains 1.7 %.  ws_balance = 2 turns it
    // on here too.)
    if (g_ws_balance >= 2) ws_balance(a, cus, cost);
  }
  const size_t lds = (size_t)KQ * NTMAX * 1024 + 16 * NTMAX * sizeof(float);
  if constexpr (!ROWS) {
    if (rows_out) {  // training forward: the rows are written as well
      auto kern = edge_ws_kernel<KQ, NTMAX, true>;
      const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
      if (lrc) return lrc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(per_slice * a.xcds)),
                         dim3(64 * kWsWaves), lds, stream, a);
      PGNN_HIP(hipGetLastError());
      return 0;
    }
  }
  auto kern = edge_ws_kernel<KQ, NTMAX, false, ROWS>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_slice * a.xcds)),
                     dim3(64 * kWsWaves), lds, stream, a);
  PGNN_HIP(hipGetLastError());
  return 0;
}

// the shapes edge_ws.h is instantiated for: one square layer of 19 (C = 300)
// or 16 (C = 256) column tiles, a CU count that splits into 8 slices
bool edge_ws_applies(const Plan &p, int64_t n_edges, int cus) {
  if (g_mlp_debug & 2048) return false;
  const LayerDev &L = p.chain.l[0];
  if (p.chain.n != 1 || L.kq != L.nt || (L.nt != 19 && L.nt != 16)) return false;
  if (cus < 64 || cus % 8 != 0) return false;
  // one column group of the layer's fragments + bias must fit a workgroup's
  // LDS (133 KiB / 128 KiB on gfx950's 160 KiB); a device with less takes the
  // LDS-tile kernel
  if ((size_t)L.kq * (L.nt == 19 ? 7 : 8) * 1024 + 16 * 8 * sizeof(float) >
      device_max_lds())
    return false;
  if (g_mlp_debug & 4096) return true;
  // below ~2 tiles per wave the fixed cost (133 KiB of weights per workgroup
  // into LDS) is not amortised
  return n_edges >= (int64_t)16 * 2 * kWsWaves * cus;
}

// Weights-stationary pooling kernel (pool_ws.h): car's 4-32-64-128-300 chain.
bool pool_ws_applies(const Plan &p, int64_t n_edges, int cus) {
  if (g_mlp_debug & (8192 | 1024)) return false;
  const ChainDev &c = p.chain;
  if (c.n != 4 || c.l[0].kq != 1 || c.l[0].nt != 2 || c.l[1].nt != 4 ||
      c.l[2].nt != 8 || c.l[3].kq != 8 || c.l[3].nt != 19)
    return false;
  if (cus < 8) return false;
  if ((size_t)8 * 19 * 1024 + 16 * 19 * sizeof(float) > device_max_lds())
    return false;
  if (g_mlp_debug & 16384) return true;
  return n_edges >= (int64_t)16 * 2 * kWsWaves * cus;
}

int launch_pool_ws(const Plan &p, const PoolArgs &pa, int64_t n_edges,
                   const SegArgs &sa, int cus, int32_t *sched,
                   hipStream_t stream, float *const *acts = nullptr,
                   int64_t ld4 = 0, const int32_t *n_dev = nullptr) {
  PoolWsArgs a = {};
  a.n_dev = n_dev;
  if (acts) {
    a.a1_out = acts[0];
    a.a2_out = acts[1];
    a.a3_out = acts[2];
    a.a4_out = acts[3];
    a.ld4 = ld4;
  }
  a.feat = pa.feat;
  a.nfeat = pa.nfeat;
  a.xyz = pa.xyz;
  a.kp = pa.kp;
  a.edges = pa.edges;
  a.n_edges = n_edges;
  a.l0 = p.chain.l[0];
  a.l1 = p.chain.l[1];
  a.l2 = p.chain.l[2];
  a.wp = p.chain.l[3].wp;
  a.kq = p.chain.l[3].kq;
  a.nt = p.chain.l[3].nt;
  a.relu_from = p.chain.l[3].relu_from;
  a.out = sa.out;
  a.ldo = sa.ldo;
  a.num_segments = sa.num_segments;
  a.sorted = sa.sorted;
  a.prio = g_ws_prio;
  a.ts = (long long *)g_mlp_ts;
  a.sched = g_ws_pool_pct > 0 ? sched : nullptr;
  PGNN_HIP((hipError_t)arm_sched(a.sched, stream));
  a.pool_pct = g_ws_pool_pct;
  a.chunk = 1;
  const size_t lds = (size_t)8 * 19 * 1024 + 16 * 19 * sizeof(float);
  if (acts) {  // training forward: the layers' activations are written too
    auto kern = pool_ws_kernel<true>;
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
    hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(64 * kWsWaves), lds,
                       stream, a);
    PGNN_HIP(hipGetLastError());
    return 0;
  }
  auto kern = pool_ws_kernel<false>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(64 * kWsWaves), lds, stream,
                     a);
  PGNN_HIP(hipGetLastError());
  return 0;
}

// Split pooling stage (pool_split.h): ped_cyl's 4-32-64-128-256-512 chain, the
// hidden rows [n_edges, 256] through a caller-provided workspace.  (car's
// 4-32-64-128-300 chain on the same form -- 64 -> 128 in LDS, rows [n_edges,
// 128], 128 -> 300 as edge_ws_kernel<8, 7, ROWS> -- measured 310 us against
// pool_ws.h's 296: its last layer is only 8 K groups deep, so a row tile
// carries half the MFMA work over the same per-tile cost; not kept.)
constexpr int kPoolSplitHidden = 256;
bool pool_split_applies(const Plan &p, int64_t n_edges, int cus) {
  if (g_mlp_debug & (8192 | 1024)) return false;
  const ChainDev &c = p.chain;
  if (c.n != 5 || c.l[0].kq != 1 || c.l[0].nt != 2 || c.l[1].nt != 4 ||
      c.l[2].nt != 8 || c.l[3].kq != 8 || c.l[3].nt != 16 || c.l[4].kq != 16 ||
      c.l[4].nt != 32)
    return false;
  if (cus < 64 || cus % 8 != 0) return false;
  if ((size_t)16 * 8 * 1024 + 16 * 16 * sizeof(float) > device_max_lds())
    return false;
  if (g_mlp_debug & 16384) return true;
  return n_edges >= (int64_t)16 * 2 * kWsWaves * cus;
}

int launch_pool_split(const Plan &p, const PoolArgs &pa, int64_t n_edges,
                      const SegArgs &sa, int cus, int32_t *sched,
                      hipStream_t stream, const int32_t *n_dev, float *hidden) {
  PoolWsArgs a = {};
  a.n_dev = n_dev;
  a.feat = pa.feat;
  a.nfeat = pa.nfeat;
  a.xyz = pa.xyz;
  a.kp = pa.kp;
  a.edges = pa.edges;
  a.n_edges = n_edges;
  a.l0 = p.chain.l[0];
  a.l1 = p.chain.l[1];
  a.l2 = p.chain.l[2];
  a.wp = p.chain.l[3].wp;
  a.kq = p.chain.l[3].kq;
  a.nt = p.chain.l[3].nt;
  a.relu_from = p.chain.l[3].relu_from;
  a.num_segments = sa.num_segments;
  a.prio = g_ws_prio;
  a.a4_out = hidden;
  a.ld4 = kPoolSplitHidden;
  a.slices = 8;
  const size_t lds = (size_t)8 * 16 * 1024 + 16 * 16 * sizeof(float);
  auto kern = pool_hidden_kernel;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(64 * kPoolHWaves), lds,
                     stream, a);
  PGNN_HIP(hipGetLastError());
  const EdgeArgs ea = {hidden, nullptr, kPoolSplitHidden, pa.edges};
  return launch_edge_ws<16, 8, true>(p.chain.l[4], ea, n_edges, sa, cus, sched,
                                     stream, nullptr, 0, nullptr, n_dev);
}

int fill_lowest(float *out, int64_t count, hipStream_t stream) {
  PGNN_HIP(hipMemsetD32Async((hipDeviceptr_t)out, (int)kFloatLowestBits,
                             (size_t)count, stream));
  return 0;
}

// Capacity form (include/pointgnn_hip.h, pgnn_dyn_count): the count an
// operator works on lives in device memory; the ordinary size argument is
// its capacity (grids, buffers) and `hint` the expected value, which only
// chooses between kernels whose results are identical.
struct Dyn {
  const int32_t *dev;
  int64_t hint;
};
inline Dyn dyn_of(const pgnn_dyn_count *c) {
  Dyn d = {nullptr, 0};
  if (c) {
    d.dev = c->dev;
    d.hint = c->hint;
  }
  return d;
}
// the size the host-side choices are made for
inline int64_t expected(const Dyn &d, int64_t cap) {
  return (d.dev && d.hint > 0 && d.hint < cap) ? d.hint : cap;
}

__global__ void fill_rows_dyn_kernel(float *__restrict__ out, int64_t ld,
                                     int64_t cap_rows,
                                     const int32_t *__restrict__ n_dev) {
  const int64_t nd = *n_dev;
  const int64_t total = (nd < cap_rows ? nd : cap_rows) * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = kFloatLowest;
}
int fill_lowest_rows(float *out, int64_t ld, int64_t rows, const Dyn &d,
                     hipStream_t stream) {
  if (!d.dev) return fill_lowest(out, rows * ld, stream);
  int64_t blocks = (expected(d, rows) * ld + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_rows_dyn_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     stream, out, ld, rows, d.dev);
  PGNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace

namespace {
// workgroups for a capacity-form launch of a 16-row-tile kernel that strides
// over its tiles: the expected tiles plus a quarter (a larger count is still
// covered by the stride), never more than the capacity needs
int64_t dyn_grid_tiles(const Dyn &d, int64_t cap_rows) {
  const int64_t cap_tiles = (cap_rows + 15) / 16;
  if (!d.dev) return cap_tiles;
  int64_t t = (expected(d, cap_rows) + 15) / 16;
  t += t / 4 + 8;
  return t < cap_tiles ? t : cap_tiles;
}

int mlp_fwd_impl(const float *x, int64_t ld_x, int32_t nx, const float *x2,
                 int64_t ld_x2, int32_t nx2, int64_t n_rows,
                 const pgnn_fc_layer *layers, int32_t n_layers,
                 const float *residual, int64_t ld_res, float *y, int64_t ld_y,
                 hipStream_t stream, const Dyn &rows, int res_gate = 0) {
  PGNN_REQUIRE(n_rows >= 0 && nx > 0 && nx2 >= 0, PGNN_E_INVALID,
               "mlp_fwd: bad sizes");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(x && y && ld_x >= nx && (nx2 == 0 || (x2 && ld_x2 >= nx2)),
               PGNN_E_INVALID, "mlp_fwd: bad input");
  Plan p;
  int rc = make_plan(layers, n_layers, nx + nx2, p);
  if (rc) return rc;
  const int out_cols = 16 * p.chain.l[n_layers - 1].nt;
  PGNN_REQUIRE(ld_y >= out_cols && (!residual || ld_res >= out_cols),
               PGNN_E_INVALID, "mlp_fwd: ld_y/ld_res < padded output width");
  RowsArgs ra = {x, ld_x, nx, x2, ld_x2, nx2, residual, ld_res, y, ld_y,
                 res_gate};
  PoolArgs pa = {};
  EdgeArgs ea = {};
  SegArgs sa = {};
  // fewer 16-row tiles than ~2 per CU: the 8-wave kernel (latency bound);
  // small row counts: 16-row tiles keep all CUs busy; large: 64-row tiles.
  // (Every output element is the same MFMA sequence over k in all three, so
  // the choice -- made on the expected count in the capacity form -- does not
  // change a bit of the result.)
  const int64_t n_sel = expected(rows, n_rows);
  if (n_sel <= 32 * (int64_t)device_cu_count() && !(g_mlp_debug & 512))
    return launch_rows8(p, n_rows, ra, stream, rows.dev,
                        rows.dev ? dyn_grid_tiles(rows, n_rows) : 0);
  if (n_sel <= 64 * (int64_t)device_cu_count())
    return launch_fused<1, PRO_ROWS>(p, n_rows, ra, pa, ea, sa, stream, nullptr,
                                     rows.dev);
  return launch_fused<4, PRO_ROWS>(p, n_rows, ra, pa, ea, sa, stream, nullptr,
                                   rows.dev);
}
}  // namespace

// Library-internal (the native training step, trainer.hip): y = gate > 0 ?
// x W : 0 -- a layer's backward dX = dY W^T with the ReluGrad of the layer
// below applied as the rows leave the kernel (gate = that layer's output).
namespace pgnn {
int mlp_rows_gated(const float *x, int64_t ld_x, int32_t nx, int64_t n_rows,
                   const pgnn_fc_layer *layer, const float *gate, int64_t ld_gate,
                   float *y, int64_t ld_y, hipStream_t stream) {
  PGNN_REQUIRE(gate != nullptr, PGNN_E_INVALID, "mlp_rows_gated: null gate");
  return mlp_fwd_impl(x, ld_x, nx, nullptr, 0, 0, n_rows, layer, 1, gate, ld_gate,
                      y, ld_y, stream, dyn_of(nullptr), 1);
}
}  // namespace pgnn

// Library-internal (trainer.hip): a chain of layers on few rows in ONE launch
// of the 8-wave kernel with every intermediate layer's output written out
// (taps[li].mid, [n_rows, 16 ceil(n_out / 16)]) and, for backward dX chains,
// gated first (taps[li].gate, same shape: out = gate > 0 ? out : 0).  The last
// layer leaves through `residual` / res_gate as in mlp_rows_gated / pgnn_mlp_fwd.
// PGNN_E_UNSUPPORTED, having done nothing, when the row count belongs to the
// larger-tile kernels or the chain does not fit the LDS: run it layer by layer.
namespace pgnn {
int mlp_rows_chain(const float *x, int64_t ld_x, int32_t nx, int64_t n_rows,
                   const pgnn_fc_layer *layers, int32_t n_layers,
                   const RowsTap *taps, const float *residual, int64_t ld_res,
                   int res_gate, float *y, int64_t ld_y, hipStream_t stream) {
  PGNN_REQUIRE(n_rows >= 0 && nx > 0 && n_layers >= 1 &&
                   n_layers <= PGNN_MAX_LAYERS && (taps || n_layers == 1),
               PGNN_E_INVALID, "mlp_rows_chain: bad sizes");
  if (n_rows == 0) return 0;
  if (n_rows > 32 * (int64_t)device_cu_count()) return PGNN_E_UNSUPPORTED;
  PGNN_REQUIRE(x && y && layers && ld_x >= nx, PGNN_E_INVALID,
               "mlp_rows_chain: bad input");
  for (int i = 0; i + 1 < n_layers; ++i)  // widths that make_plan refuses
    if ((layers[i].n_out + 15) / 16 != (layers[i + 1].k_in + 15) / 16 ||
        (layers[i].n_out + 15) / 16 > kMaxTilesPerPass)
      return PGNN_E_UNSUPPORTED;
  Plan p;
  int rc = make_plan(layers, n_layers, nx, p);
  if (rc) return rc;
  if (plan_lds_bytes(p, 16) > 160 * 1024) return PGNN_E_UNSUPPORTED;
  const int out_cols = 16 * p.chain.l[n_layers - 1].nt;
  PGNN_REQUIRE(ld_y >= out_cols && (!residual || ld_res >= out_cols),
               PGNN_E_INVALID, "mlp_rows_chain: ld_y/ld_res < padded output width");
  RowsArgs ra = {x, ld_x, nx, nullptr, 0, 0, residual, ld_res, y, ld_y, res_gate};
  for (int li = 0; li + 1 < n_layers; ++li) {
    ra.mid[li] = taps[li].mid;
    ra.gate[li] = taps[li].gate;
  }
  return launch_rows8<true>(p, n_rows, ra, stream);
}
}  // namespace pgnn

extern "C" int pgnn_mlp_fwd(const float *x, int64_t ld_x, int32_t nx,
                            const float *x2, int64_t ld_x2, int32_t nx2,
                            int64_t n_rows, const pgnn_fc_layer *layers,
                            int32_t n_layers, const float *residual,
                            int64_t ld_res, float *y, int64_t ld_y,
                            void *stream_) {
  PGNN_GUARD_BEGIN
  return mlp_fwd_impl(x, ld_x, nx, x2, ld_x2, nx2, n_rows, layers, n_layers,
                      residual, ld_res, y, ld_y, (hipStream_t)stream_,
                      dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_mlp_fwd_dyn(const float *x, int64_t ld_x, int32_t nx,
                                const float *x2, int64_t ld_x2, int32_t nx2,
                                int64_t rows_cap, const pgnn_fc_layer *layers,
                                int32_t n_layers, const float *residual,
                                int64_t ld_res, float *y, int64_t ld_y,
                                const pgnn_dyn_count *rows, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(rows && rows->dev, PGNN_E_INVALID, "mlp_fwd_dyn: null count");
  return mlp_fwd_impl(x, ld_x, nx, x2, ld_x2, nx2, rows_cap, layers, n_layers,
                      residual, ld_res, y, ld_y, (hipStream_t)stream_,
                      dyn_of(rows));
  PGNN_GUARD_END
}

namespace {
int pooling_fwd_impl(const float *point_features, int32_t n_feat,
                     const float *point_xyz, const int32_t *keypoint_indices,
                     const int32_t *edges, int64_t n_edges,
                     int32_t num_keypoints, const pgnn_fc_layer *layers,
                     int32_t n_layers, int32_t edges_sorted, float *out,
                     int64_t ld_out, int32_t *sched_ws, hipStream_t stream,
                     const Dyn &de, const Dyn &dk, void *workspace = nullptr,
                     size_t workspace_bytes = 0) {
  PGNN_REQUIRE(n_edges >= 0 && num_keypoints >= 0 && n_feat >= 0 && n_feat <= 13,
               PGNN_E_INVALID, "pooling: bad sizes (n_feat <= 13)");
  Plan p;
  int rc = make_plan(layers, n_layers, n_feat + 3, p);
  if (rc) return rc;
  PGNN_REQUIRE(p.chain.l[0].kq == 1, PGNN_E_UNSUPPORTED,
               "pooling: first layer k_in must be <= 16");
  const int out_cols = 16 * p.chain.l[n_layers - 1].nt;
  PGNN_REQUIRE(ld_out >= out_cols, PGNN_E_INVALID,
               "pooling: ld_out < padded output width");
  if (num_keypoints == 0) return 0;  // (an empty output has no buffer)
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "pooling: null output");
  rc = fill_lowest_rows(out, ld_out, num_keypoints, dk, stream);
  if (rc) return rc;
  if (n_edges == 0) return 0;
  PGNN_REQUIRE((n_feat == 0 || point_features) && point_xyz &&
                   keypoint_indices && edges,
               PGNN_E_INVALID, "pooling: null input");
  RowsArgs ra = {};
  PoolArgs pa = {point_features, n_feat, point_xyz, keypoint_indices, edges, 0};
  EdgeArgs ea = {};
  SegArgs sa = {out, ld_out, num_keypoints, edges_sorted};
  int msub = g_pool_msub;
  if (msub != 2 && msub != 4)
    msub = (plan_lds_bytes(p, 64) <= 80 * 1024 ||
            plan_lds_bytes(p, 32) > 80 * 1024) ? 4 : 2;
  // hidden layers in registers when their widths are car's 32-64-128 and the
  // tile has 64 rows; anything else takes the LDS tile
  if (msub == 4 && !(g_mlp_debug & 1024)) {
    const ChainDev &c = p.chain;
    auto nts = [&](int i) { return c.l[i].nt; };
    if (c.n == 4 && nts(0) == 2 && nts(1) == 4 && nts(2) == 8)
      pa.reg_hidden = 1;
  }
  {
    int cus = stream_cu_count(stream);
    if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
    // (capacity form: chosen on the expected count; the two kernels give
    // bit-identical maxima, tests/test_gpu_parity.py)
    if (pool_ws_applies(p, expected(de, n_edges), cus))
      return launch_pool_ws(p, pa, n_edges, sa, cus, sched_ws, stream, nullptr,
                            0, de.dev);
    // (likewise bit-identical; needs the hidden rows' workspace)
    if (workspace && pool_split_applies(p, expected(de, n_edges), cus)) {
      PGNN_REQUIRE((uintptr_t)workspace % 16 == 0, PGNN_E_INVALID,
                   "pooling: workspace not 16-byte aligned");
      PGNN_REQUIRE(workspace_bytes >=
                       (size_t)n_edges * kPoolSplitHidden * sizeof(float),
                   PGNN_E_WORKSPACE, "pooling: workspace too small");
      return launch_pool_split(p, pa, n_edges, sa, cus, sched_ws, stream, de.dev,
                               static_cast<float *>(workspace));
    }
  }
  if (msub == 4 && pa.reg_hidden == 1)
    return launch_fused<4, PRO_POOL_R3>(p, n_edges, ra, pa, ea, sa, stream,
                                        sched_ws, de.dev);
  if (msub == 4)
    return launch_fused<4, PRO_POOL>(p, n_edges, ra, pa, ea, sa, stream,
                                     sched_ws, de.dev);
  return launch_fused<2, PRO_POOL>(p, n_edges, ra, pa, ea, sa, stream, sched_ws,
                                   de.dev);
}

int edge_fwd_impl(const float *P, const float *Q, int64_t ld_pq, int32_t width,
                  const int32_t *edges, int64_t n_edges, int32_t num_vertices,
                  const pgnn_fc_layer *layers, int32_t n_layers,
                  int32_t edges_sorted, float *out, int64_t ld_out,
                  int32_t *sched_ws, hipStream_t stream, const Dyn &de,
                  const Dyn &dk) {
  PGNN_REQUIRE(n_edges >= 0 && num_vertices >= 0 && width > 0, PGNN_E_INVALID,
               "edge_mlp: bad sizes");
  Plan p;
  int rc = make_plan(layers, n_layers, width, p);
  if (rc) return rc;
  PGNN_REQUIRE(ld_pq == 16 * p.chain.l[0].kq, PGNN_E_INVALID,
               "edge_mlp: ld_pq must equal the padded width 16*ceil(width/16)");
  const int out_cols = 16 * p.chain.l[n_layers - 1].nt;
  PGNN_REQUIRE(ld_out >= out_cols, PGNN_E_INVALID,
               "edge_mlp: ld_out < padded output width");
  if (num_vertices == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "edge_mlp: null output");
  if (!(edges_sorted & 2)) {  // bit 1: caller already filled `out` with lowest()
    rc = fill_lowest_rows(out, ld_out, num_vertices, dk, stream);
    if (rc) return rc;
  }
  if (n_edges == 0) return 0;
  PGNN_REQUIRE(P && Q && edges, PGNN_E_INVALID, "edge_mlp: null input");
  PGNN_REQUIRE(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0),
               PGNN_E_INVALID, "edge_mlp: P/Q must be 16-byte aligned");
  RowsArgs ra = {};
  PoolArgs pa = {};
  EdgeArgs ea = {P, Q, ld_pq, edges};
  SegArgs sa = {out, ld_out, num_vertices, edges_sorted & 1};
  {
    int cus = stream_cu_count(stream);
    if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
    if (edge_ws_applies(p, expected(de, n_edges), cus)) {
      if (p.chain.l[0].nt == 19)
        return launch_edge_ws<19, 7>(p.chain.l[0], ea, n_edges, sa, cus,
                                     sched_ws, stream, nullptr, 0, nullptr,
                                     de.dev);
      return launch_edge_ws<16, 8>(p.chain.l[0], ea, n_edges, sa, cus, sched_ws,
                                   stream, nullptr, 0, nullptr, de.dev);
    }
  }
  int msub = g_edge_msub;
  if (msub != 2 && msub != 4)
    msub = (plan_lds_bytes(p, 64) <= 80 * 1024 ||
            plan_lds_bytes(p, 32) > 80 * 1024) ? 4 : 2;
  if (msub == 4)
    return launch_fused<4, PRO_EDGE>(p, n_edges, ra, pa, ea, sa, stream,
                                     sched_ws, de.dev);
  return launch_fused<2, PRO_EDGE>(p, n_edges, ra, pa, ea, sa, stream, sched_ws,
                                   de.dev);
}
}  // namespace

namespace {
// Balanced workgroup counts over the whole chip for the column groups whose
// tile counts are a.tile0 (see EdgeWsArgs::balanced).  cost[g] = relative time
// of one row tile in group g (its MFMA issue + the per-tile fixed part).
void ws_balance(EdgeWsArgs &a, int cus, const double *cost) {
  const int per_slice = cus / a.xcds, total = per_slice * a.xcds;
  a.balanced = 0;
  if (a.xcds > kWsMaxSlices || a.groups < 2 || a.sched) return;
  double sum = 0;
  for (int g = 0; g < a.groups; ++g) sum += cost[g];
  // largest-remainder share of `total` workgroups
  int n[kWsMaxGroups], used = 0;
  double frac[kWsMaxGroups];
  for (int g = 0; g < a.groups; ++g) {
    const double x = total * cost[g] / sum;
    n[g] = (int)x;
    if (n[g] < a.xcds) n[g] = a.xcds;  // at least one per slice
    frac[g] = x - n[g];
    used += n[g];
  }
  while (used < total) {
    int best = 0;
    for (int g = 1; g < a.groups; ++g)
      if (frac[g] > frac[best]) best = g;
    ++n[best];
    frac[best] -= 1.0;
    ++used;
  }
  if (used != total) return;  // (cannot happen for the shipped shapes)
  // per slice: cumulative rounding, then repair each slice's sum to per_slice
  int prev[kWsMaxGroups] = {0, 0, 0, 0};
  for (int s = 0; s < a.xcds; ++s) {
    int c[kWsMaxGroups], tot = 0;
    for (int g = 0; g < a.groups; ++g) {
      const int cum = (int)((int64_t)n[g] * (s + 1) / a.xcds);
      c[g] = cum - prev[g];
      tot += c[g];
    }
    // (the last slice closes every group exactly; earlier slices borrow from /
    // lend to the group that is furthest ahead / behind its share)
    for (int guard = 0; tot != per_slice && guard < 64; ++guard) {
      int pick = -1;
      double worst = 0;
      for (int g = 0; g < a.groups; ++g) {
        const double ideal = (double)n[g] * (s + 1) / a.xcds;
        const double ahead = prev[g] + c[g] - ideal;
        if (tot > per_slice ? (c[g] > 1 && (pick < 0 || ahead > worst))
                            : (prev[g] + c[g] < n[g] &&
                               (pick < 0 || -ahead > worst))) {
          pick = g;
          worst = tot > per_slice ? ahead : -ahead;
        }
      }
      if (pick < 0) return;
      c[pick] += tot > per_slice ? -1 : 1;
      tot += tot > per_slice ? -1 : 1;
    }
    if (tot != per_slice) return;
    a.swg0[s][0] = 0;
    for (int g = 0; g < a.groups; ++g) {
      a.sbase[s][g] = (short)prev[g];
      a.swg0[s][g + 1] = (short)(a.swg0[s][g] + c[g]);
      prev[g] += c[g];
    }
  }
  for (int g = 0; g < a.groups; ++g) {
    if (prev[g] != n[g]) return;
    a.n_wg[g] = n[g];
  }
  a.balanced = 1;
}
}  // namespace

namespace {
// column groups of a weights-stationary launch: tiles as evenly as possible,
// the slice's workgroups in proportion (largest remainder) -- launch_edge_ws
int ws_partition(EdgeWsArgs &a, int nt, int ntmax, int cus) {
  a.groups = (nt + ntmax - 1) / ntmax;
  const int per_slice = cus / a.xcds;
  PGNN_REQUIRE(a.groups <= kWsMaxGroups && per_slice >= a.groups,
               PGNN_E_UNSUPPORTED, "edge_ws: too few CUs for the column groups");
  const int base = nt / a.groups, extra = nt % a.groups;
  int size[kWsMaxGroups], cnt[kWsMaxGroups], frac[kWsMaxGroups], used = 0;
  a.tile0[0] = 0;
  for (int g = 0; g < a.groups; ++g) {
    size[g] = base + (g < extra ? 1 : 0);
    a.tile0[g + 1] = a.tile0[g] + size[g];
    cnt[g] = per_slice * size[g] / nt;
    if (cnt[g] < 1) cnt[g] = 1;
    frac[g] = per_slice * size[g] % nt;
    used += cnt[g];
  }
  PGNN_REQUIRE(base >= ntmax - 1 && base + (extra ? 1 : 0) <= ntmax &&
                   used <= per_slice,
               PGNN_E_UNSUPPORTED, "edge_ws: column tiles do not group");
  while (used < per_slice) {
    int best = 0;
    for (int g = 1; g < a.groups; ++g)
      if (frac[g] > frac[best]) best = g;
    ++cnt[best];
    frac[best] = -1;
    ++used;
  }
  a.wg0[0] = 0;
  for (int g = 0; g < a.groups; ++g) a.wg0[g + 1] = a.wg0[g] + cnt[g];
  return 0;
}

template <int KB, int NTMAX>
int launch_edge_ws3(EdgeWsArgs &a, int nt, int cus, hipStream_t stream) {
  const int rc = ws_partition(a, nt, NTMAX, cus);
  if (rc) return rc;
  {
    // relative cost of a row tile: 60 MFMAs of 16 cycles per column tile + the
    // part of the gather / split / segmented max that does not hide behind them
    // (round 4's 2.4k cycles, when nothing did; the interleaved body leaves
    // about a third: tools/sessions/r05_s9.sh)
    double cost[kWsMaxGroups];
    for (int g = 0; g < a.groups; ++g)
      cost[g] = 960.0 * KB / 10 * (a.tile0[g + 1] - a.tile0[g]) + 2400.0;
    if (g_ws_balance) ws_balance(a, cus, cost);
  }
  const size_t lds = (size_t)KB * NTMAX * 3 * 1024 + 16 * NTMAX * sizeof(float);
  PGNN_REQUIRE(lds <= device_max_lds(), PGNN_E_UNSUPPORTED,
               "edge_bf16x3: column group does not fit the LDS");
  auto kern = edge_ws_bf16x3_kernel<KB, NTMAX>;
  const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
  if (lrc) return lrc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(cus / a.xcds * a.xcds)),
                     dim3(64 * kWsWaves), lds, stream, a);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_edge_mlp_scatter_max_bf16x3_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t edges_cap, int32_t vertices_cap,
    const void *image, int32_t n_out, int32_t relu_from, int32_t edges_sorted,
    float *out, int64_t ld_out, const pgnn_dyn_count *n_edges,
    const pgnn_dyn_count *num_vertices, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(edges_cap >= 0 && vertices_cap >= 0 && width > 0 && n_out > 0 &&
                   image,
               PGNN_E_INVALID, "edge_bf16x3: bad argument");
  const Dyn de = dyn_of(n_edges), dk = dyn_of(num_vertices);
  const int kq = (width + 15) / 16, nt = (n_out + 15) / 16;
  const int kb = (width + 31) / 32;
  PGNN_REQUIRE(ld_pq == 16 * kq && ld_out >= 16 * nt, PGNN_E_INVALID,
               "edge_bf16x3: ld_pq / ld_out do not match the padded widths");
  int cus = stream_cu_count(stream);
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  // the shapes the kernel is instantiated for, and enough rows to amortise
  // 150 KiB of weights per workgroup: otherwise the caller runs the fp32 entry
  if (!((kb == 10 && nt == 19) || (kb == 8 && nt == 16)) || cus < 64 ||
      cus % 8 != 0 ||
      (!g_b16_force &&
       expected(de, edges_cap) < (int64_t)16 * 2 * kWsWaves * cus) ||
      // (the kernel addresses P / Q rows with 32-bit byte offsets)
      (int64_t)vertices_cap * ld_pq * 4 >= ((int64_t)1 << 32))
    return PGNN_E_UNSUPPORTED;  // (no message: an expected answer)
  if (vertices_cap == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "edge_bf16x3: null output");
  if (!(edges_sorted & 2)) {
    const int rc = fill_lowest_rows(out, ld_out, vertices_cap, dk, stream);
    if (rc) return rc;
  }
  if (edges_cap == 0) return 0;
  PGNN_REQUIRE(P && Q && edges, PGNN_E_INVALID, "edge_bf16x3: null input");
  PGNN_REQUIRE(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0) &&
                   ((uintptr_t)image % 16 == 0),
               PGNN_E_INVALID, "edge_bf16x3: P / Q / image must be 16-byte aligned");
  EdgeWsArgs a = {};
  a.P = P;
  a.Q = Q;
  a.ldv4 = (int)(ld_pq >> 2);
  a.edges = edges;
  a.n_edges = edges_cap;
  a.n_dev = de.dev;
  a.wp = reinterpret_cast<const float *>(image);
  a.nt = nt;
  a.relu_from = relu_from;
  a.out = out;
  a.ldo = ld_out;
  a.num_segments = vertices_cap;
  a.sorted = edges_sorted & 1;
  a.xcds = (g_ws_xcds >= 1 && cus % g_ws_xcds == 0) ? g_ws_xcds : 8;
  a.prio = g_ws_prio;
  if (kb == 10) return launch_edge_ws3<10, 5>(a, nt, cus, stream);
  return launch_edge_ws3<8, 5>(a, nt, cus, stream);
  PGNN_GUARD_END
}

namespace {
template <int KB, int NTMAX>
int launch_edge_ws2(EdgeWsArgs &a, int nt, int cus, int32_t *status,
                    hipStream_t stream) {
  const int rc = ws_partition(a, nt, NTMAX, cus);
  if (rc) return rc;
  {
    // relative cost of a row tile: 30 MFMAs of 16 cycles per column tile + the
    // part of the gather / split / segmented max that does not hide behind them
    double cost[kWsMaxGroups];
    for (int g = 0; g < a.groups; ++g)
      cost[g] = 480.0 * KB / 10 * (a.tile0[g + 1] - a.tile0[g]) + 1600.0;
    if (g_ws_balance) ws_balance(a, cus, cost);
  }
  const size_t lds = (size_t)KB * NTMAX * 2 * 1024 + 16 * NTMAX * sizeof(float);
  PGNN_REQUIRE(lds <= device_max_lds(), PGNN_E_UNSUPPORTED,
               "edge_f16x2: column group does not fit the LDS");
  auto kern = edge_ws_f16x2_kernel<KB, NTMAX>;
  const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
  if (lrc) return lrc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(cus / a.xcds * a.xcds)),
                     dim3(64 * kWsWaves), lds, stream, a, status);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_edge_mlp_scatter_max_f16x2_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t edges_cap, int32_t vertices_cap,
    const void *image, int32_t n_out, int32_t relu_from, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *status, const pgnn_dyn_count *n_edges,
    const pgnn_dyn_count *num_vertices, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(edges_cap >= 0 && vertices_cap >= 0 && width > 0 && n_out > 0 &&
                   image,
               PGNN_E_INVALID, "edge_f16x2: bad argument");
  const Dyn de = dyn_of(n_edges), dk = dyn_of(num_vertices);
  const int kq = (width + 15) / 16, nt = (n_out + 15) / 16;
  const int kb = (width + 31) / 32;
  PGNN_REQUIRE(ld_pq == 16 * kq && ld_out >= 16 * nt, PGNN_E_INVALID,
               "edge_f16x2: ld_pq / ld_out do not match the padded widths");
  int cus = stream_cu_count(stream);
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  if (!((kb == 10 && nt == 19) || (kb == 8 && nt == 16)) || cus < 64 ||
      cus % 8 != 0 ||
      (!g_b16_force &&
       expected(de, edges_cap) < (int64_t)16 * 2 * kWsWaves * cus) ||
      (int64_t)vertices_cap * ld_pq * 4 >= ((int64_t)1 << 32))
    return PGNN_E_UNSUPPORTED;  // (no message: an expected answer)
  if (vertices_cap == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "edge_f16x2: null output");
  if (!(edges_sorted & 2)) {
    const int rc = fill_lowest_rows(out, ld_out, vertices_cap, dk, stream);
    if (rc) return rc;
  }
  if (edges_cap == 0) return 0;
  PGNN_REQUIRE(P && Q && edges, PGNN_E_INVALID, "edge_f16x2: null input");
  PGNN_REQUIRE(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0) &&
                   ((uintptr_t)image % 16 == 0),
               PGNN_E_INVALID, "edge_f16x2: P / Q / image must be 16-byte aligned");
  EdgeWsArgs a = {};
  a.P = P;
  a.Q = Q;
  a.ldv4 = (int)(ld_pq >> 2);
  a.edges = edges;
  a.n_edges = edges_cap;
  a.n_dev = de.dev;
  a.wp = reinterpret_cast<const float *>(image);
  a.nt = nt;
  a.relu_from = relu_from;
  a.out = out;
  a.ldo = ld_out;
  a.num_segments = vertices_cap;
  a.sorted = edges_sorted & 1;
  a.xcds = (g_ws_xcds >= 1 && cus % g_ws_xcds == 0) ? g_ws_xcds : 8;
  a.prio = 0;
  a.nv_dev = dk.dev;
  if (kb == 10) return launch_edge_ws2<10, 7>(a, nt, cus, status, stream);
  return launch_edge_ws2<8, 6>(a, nt, cus, status, stream);
  PGNN_GUARD_END
}

extern "C" int pgnn_point_set_pooling_f16x2_fwd(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t edges_cap,
    int32_t keypoints_cap, const pgnn_fc_layer *layers, int32_t n_layers,
    const void *last_image, const void *hidden_image, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *sched_ws, int32_t *status,
    const pgnn_dyn_count *n_edges, const pgnn_dyn_count *num_keypoints,
    void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(edges_cap >= 0 && keypoints_cap >= 0 && n_feat >= 0 &&
                   n_feat <= 13 && layers && last_image,
               PGNN_E_INVALID, "pooling_f16x2: bad argument");
  const Dyn de = dyn_of(n_edges), dk = dyn_of(num_keypoints);
  Plan p;
  int rc = make_plan(layers, n_layers, n_feat + 3, p);
  if (rc) return rc;
  int cus = stream_cu_count(stream);
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  // the shapes and sizes of pool_ws_kernel (car's 4-32-64-128-300 chain)
  if (!g_f16_pool ||
      !pool_ws_applies(p, g_b16_force ? ((int64_t)1 << 40)
                                      : expected(de, edges_cap), cus))
    return PGNN_E_UNSUPPORTED;  // (no message: an expected answer)
  PGNN_REQUIRE(ld_out >= 16 * 19, PGNN_E_INVALID,
               "pooling_f16x2: ld_out < padded output width");
  if (keypoints_cap == 0) return 0;
  PGNN_REQUIRE(out != nullptr, PGNN_E_INVALID, "pooling_f16x2: null output");
  rc = fill_lowest_rows(out, ld_out, keypoints_cap, dk, stream);
  if (rc) return rc;
  if (edges_cap == 0) return 0;
  PGNN_REQUIRE((n_feat == 0 || point_features) && point_xyz &&
                   keypoint_indices && edges,
               PGNN_E_INVALID, "pooling_f16x2: null input");
  PGNN_REQUIRE((uintptr_t)last_image % 16 == 0 &&
                   (uintptr_t)hidden_image % 16 == 0,
               PGNN_E_INVALID,
               "pooling_f16x2: the images must be 16-byte aligned");
  PoolWsArgs a = {};
  a.n_dev = de.dev;
  a.feat = point_features;
  a.nfeat = n_feat;
  a.xyz = point_xyz;
  a.kp = keypoint_indices;
  a.edges = edges;
  a.n_edges = edges_cap;
  a.l0 = p.chain.l[0];
  a.l1 = p.chain.l[1];
  a.l2 = p.chain.l[2];
  a.wp = reinterpret_cast<const float *>(last_image);
  a.kq = 8;
  a.nt = 19;
  a.relu_from = p.chain.l[3].relu_from;
  a.out = out;
  a.ldo = ld_out;
  a.num_segments = keypoints_cap;
  a.sorted = edges_sorted & 1;
  a.sched = g_ws_pool_pct > 0 ? sched_ws : nullptr;
  PGNN_HIP((hipError_t)arm_sched(a.sched, stream));
  a.pool_pct = g_ws_pool_pct;
  a.chunk = 1;
  const size_t lds = (size_t)4 * 19 * 2 * 1024 + 16 * 19 * sizeof(float);
  a.l2_f16 = g_f16_pool >= 2 ? nullptr : hidden_image;
  auto kern = a.l2_f16 ? pool_ws_f16x2_kernel<true> : pool_ws_f16x2_kernel<false>;
  rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(64 * kWsWaves), lds, stream,
                     a, status);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_point_set_pooling_fwd(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t n_edges,
    int32_t num_keypoints, const pgnn_fc_layer *layers, int32_t n_layers,
    int32_t edges_sorted, float *out, int64_t ld_out, int32_t *sched_ws,
    void *stream_) {
  PGNN_GUARD_BEGIN
  return pooling_fwd_impl(point_features, n_feat, point_xyz, keypoint_indices,
                          edges, n_edges, num_keypoints, layers, n_layers,
                          edges_sorted, out, ld_out, sched_ws,
                          (hipStream_t)stream_, dyn_of(nullptr),
                          dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_point_set_pooling_fwd_dyn(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t edges_cap,
    int32_t keypoints_cap, const pgnn_fc_layer *layers, int32_t n_layers,
    int32_t edges_sorted, float *out, int64_t ld_out, int32_t *sched_ws,
    const pgnn_dyn_count *n_edges, const pgnn_dyn_count *num_keypoints,
    void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_edges && n_edges->dev && num_keypoints && num_keypoints->dev,
               PGNN_E_INVALID, "pooling_dyn: null count");
  return pooling_fwd_impl(point_features, n_feat, point_xyz, keypoint_indices,
                          edges, edges_cap, keypoints_cap, layers, n_layers,
                          edges_sorted, out, ld_out, sched_ws,
                          (hipStream_t)stream_, dyn_of(n_edges),
                          dyn_of(num_keypoints));
  PGNN_GUARD_END
}

extern "C" int pgnn_point_set_pooling_workspace_bytes(
    const pgnn_fc_layer *layers, int32_t n_layers, int32_t n_feat,
    int64_t edges_cap, int64_t edges_hint, size_t *bytes) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(bytes && layers && edges_cap >= 0 && n_feat >= 0 && n_feat <= 13,
               PGNN_E_INVALID, "pooling_workspace_bytes: bad arguments");
  *bytes = 0;
  Plan p;
  int rc = make_plan(layers, n_layers, n_feat + 3, p);
  if (rc) return rc;
  int cus = device_cu_count();
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  const int64_t n_sel =
      (edges_hint > 0 && edges_hint < edges_cap) ? edges_hint : edges_cap;
  if (pool_split_applies(p, n_sel, cus))
    *bytes = (size_t)edges_cap * kPoolSplitHidden * sizeof(float);
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_point_set_pooling_fwd_ws(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t edges_cap,
    int32_t keypoints_cap, const pgnn_fc_layer *layers, int32_t n_layers,
    int32_t edges_sorted, float *out, int64_t ld_out, int32_t *sched_ws,
    const pgnn_dyn_count *n_edges, const pgnn_dyn_count *num_keypoints,
    void *workspace, size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE((n_edges == nullptr) == (num_keypoints == nullptr) &&
                   (!n_edges || (n_edges->dev && num_keypoints->dev)),
               PGNN_E_INVALID, "pooling_ws: counts come as a pair");
  return pooling_fwd_impl(point_features, n_feat, point_xyz, keypoint_indices,
                          edges, edges_cap, keypoints_cap, layers, n_layers,
                          edges_sorted, out, ld_out, sched_ws,
                          (hipStream_t)stream_, dyn_of(n_edges),
                          dyn_of(num_keypoints), workspace, workspace_bytes);
  PGNN_GUARD_END
}

extern "C" int pgnn_edge_mlp_scatter_max_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t n_edges, int32_t num_vertices,
    const pgnn_fc_layer *layers, int32_t n_layers, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *sched_ws, void *stream_) {
  PGNN_GUARD_BEGIN
  return edge_fwd_impl(P, Q, ld_pq, width, edges, n_edges, num_vertices, layers,
                       n_layers, edges_sorted, out, ld_out, sched_ws,
                       (hipStream_t)stream_, dyn_of(nullptr), dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_edge_mlp_scatter_max_fwd_dyn(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t edges_cap, int32_t vertices_cap,
    const pgnn_fc_layer *layers, int32_t n_layers, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *sched_ws,
    const pgnn_dyn_count *n_edges, const pgnn_dyn_count *num_vertices,
    void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_edges && n_edges->dev && num_vertices && num_vertices->dev,
               PGNN_E_INVALID, "edge_mlp_dyn: null count");
  return edge_fwd_impl(P, Q, ld_pq, width, edges, edges_cap, vertices_cap,
                       layers, n_layers, edges_sorted, out, ld_out, sched_ws,
                       (hipStream_t)stream_, dyn_of(n_edges),
                       dyn_of(num_vertices));
  PGNN_GUARD_END
}

// ---- vertex side of one GraphNetAutoCenter iteration, before the edge kernel ----
// gnn.py:341-356 per vertex: delta = offset MLP(h); Q = (x + delta) W1[C:];
// P = [h, x] W1 + b1; plus the lowest() fill of the aggregation buffer.  Three
// launches + a memset in the unfused form (~46 us per iteration at K = 2.9k,
// each a 184-workgroup grid that ends before it fills the chip); here one
// 16-row tile per (8-wave, see rows_mlp_kernel) workgroup keeps [h, x] in LDS
// for both chains.
namespace {

struct PreEdgeArgs {
  const float *h;
  int64_t ld_h;
  int c;
  const float *xyz;
  const float *wx;
  int64_t n;
  const int32_t *n_dev;  // nullable: capacity form (the count is on the device)
  float *P, *Q;
  int64_t ld_pq;
  float *agg;
  int64_t ld_agg;
  int ld_tile, ld_scratch;  // LDS leading dimensions
};

// What vertex_pre_edge_kernel does once the tile holds [h | x | 0] (and the
// workgroup has met at a barrier): offset chain, Q, P.  Shared with the fused
// update + pre-edge kernel below.
__device__ __forceinline__ void pre_edge_tail(const ChainDev &off,
                                              const LayerDev &pl,
                                              const PreEdgeArgs &a, float *tile,
                                              float *scratch, float *stage,
                                              int64_t row0, int rows_valid,
                                              int wave, int lane, KrowPre &pre) {
  // (`pre`: krow_prefetch of the first pass below -- off.l[0], or pl without
  // an offset chain)
  // offset chain: first layer reads the h columns of the tile (its packed
  // weights are zero beyond k_in, so the x columns do not contribute)
  const float *delta = nullptr;
  int ld_delta = 0;
  for (int li = 0; li < off.n; ++li) {
    const LayerDev &L = off.l[li];
    const float *in = li == 0 ? tile : scratch;
    const int ld_in = li == 0 ? a.ld_tile : lds_ld(16 * L.kq);
    krow_pass(in, ld_in, scratch, lds_ld(16 * L.nt), L, 0, wave, lane, pre,
              true, li + 1 < off.n ? off.l[li + 1] : pl, 0);
    delta = scratch;
    ld_delta = lds_ld(16 * L.nt);
  }
  // Q = (x + delta) @ wx, the same expression as offset_apply_kernel; thread
  // (r, c0) as in consume_rows16, the three wx rows requested up front
  {
    const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
    const int ldq = (int)a.ld_pq;  // <= 320
    const int64_t row = row0 + (r < rows_valid ? r : rows_valid - 1);
    float x0 = a.xyz[row * 3], x1 = a.xyz[row * 3 + 1], x2 = a.xyz[row * 3 + 2];
    float w0[kRowSteps], w1[kRowSteps], w2[kRowSteps];
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = c0 + 32 * j;
      const int cc = c < ldq ? c : ldq - 1;
      w0[j] = a.wx[cc];
      w1[j] = a.wx[ldq + cc];
      w2[j] = a.wx[2 * ldq + cc];
    }
    if (delta) {
      x0 = x0 + delta[r * ld_delta];
      x1 = x1 + delta[r * ld_delta + 1];
      x2 = x2 + delta[r * ld_delta + 2];
    }
    float *__restrict__ qr = a.Q + row * a.ld_pq;
#pragma unroll
    for (int j = 0; j < kRowSteps; ++j) {
      const int c = c0 + 32 * j;
      if (r < rows_valid && c < ldq)
        qr[c] = (x0 * w0[j] + x1 * w1[j]) + x2 * w2[j];
    }
  }
  // P = [h, x] @ W1 + b1
  const int ld_st = lds_ld(16 * pl.nt);
  krow_pass(tile, a.ld_tile, stage, ld_st, pl, 0, wave, lane, pre, false, pl, 0);
  RowsArgs ra = {};
  ra.y = a.P;
  ra.ldy = a.ld_pq;
  consume_rows16(stage, ld_st, row0, rows_valid, 0, 16 * pl.nt, ra);
}

template <bool STRIDE>
__global__ __launch_bounds__(64 * kRowsWaves) void vertex_pre_edge_kernel(
    ChainDev off, LayerDev pl, PreEdgeArgs a) {
  constexpr int ROWS = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);      // [16][ld_tile]: [h | x | 0]
  float *scratch = tile + ROWS * a.ld_tile;           // offset-chain activations
  float *stage = scratch + ROWS * a.ld_scratch;       // P before it leaves
  const int lane_id = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t n_rows = a.n;
  if (a.n_dev) {
    const int64_t nd = *a.n_dev;
    n_rows = nd < n_rows ? nd : n_rows;
  }
  // one tile per workgroup when the host knows the count (STRIDE = false: the
  // loop below runs once and folds away); the capacity form launches for the
  // expected count and strides
  for (int64_t row0 = (int64_t)blockIdx.x * ROWS; row0 < n_rows;
       row0 += STRIDE ? (int64_t)gridDim.x * ROWS : n_rows) {
    // (an opaque per-tile copy of the lane id: the per-lane weight addresses
    // of every layer are loop invariant, and hoisted out of the tile loop they
    // cost ~100 VGPRs -- spills in the capacity-form kernels)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_id));
  const int rows_valid =
      (int)((n_rows - row0 < ROWS) ? (n_rows - row0) : ROWS);
  const int kc = 16 * pl.kq;
  KrowPre pre;
  krow_prefetch(off.n > 0 ? off.l[0] : pl, 0, wave, lane, pre);
  {
    // 32 threads per row; unconditional clamped loads, select on the value,
    // all in flight before the first LDS write (see load_rows16)
    static_assert(64 * kRowsWaves == 32 * ROWS, "one 32-thread group per row");
    const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
    const int64_t rowc = row0 + (r < rows_valid ? r : rows_valid - 1);
    const float *__restrict__ hr = a.h + rowc * a.ld_h;
    const float *__restrict__ xr = a.xyz + rowc * 3;
    for (int cb = 0; cb < kc; cb += 32 * kRowSteps) {
      float v[kRowSteps];
#pragma unroll
      for (int j = 0; j < kRowSteps; ++j) {
        const int c = cb + c0 + 32 * j;
        const int d = c - a.c;
        // the three coordinate columns follow the features
        const float hv = hr[c < a.c ? c : a.c - 1];
        const float xv = xr[d < 0 ? 0 : (d > 2 ? 2 : d)];
        v[j] = c < a.c ? hv : xv;
      }
#pragma unroll
      for (int j = 0; j < kRowSteps; ++j) {
        const int c = cb + c0 + 32 * j;
        if (c < kc)
          tile[r * a.ld_tile + c] =
              (r < rows_valid && c < a.c + 3) ? v[j] : 0.0f;
      }
    }
  }
  // lowest() rows of the aggregation buffer the edge kernel maxes into
  if (a.agg) {
    for (int idx = threadIdx.x; idx < rows_valid * (int)a.ld_agg; idx += 64 * kRowsWaves)
      a.agg[row0 * a.ld_agg + idx] = kFloatLowest;
  }
  __syncthreads();
  pre_edge_tail(off, pl, a, tile, scratch, stage, row0, rows_valid, wave, lane,
                pre);
  if (STRIDE) __syncthreads();  // the next tile overwrites tile / scratch / stage
  }
}

// ---- two vertex stages in ONE launch -------------------------------------------
// Between two edge stages the per-vertex work is a chain of small K-row MLPs:
//   update MLP(agg) + h  (gnn.py:367-372, the END of iteration i)
//   offset MLP, Q, P     (gnn.py:341-356, the START of iteration i + 1)
// (and, round the model's ends, the pooling stage's output MLP in front of
// iteration 1 and the predictor heads behind iteration T).  As separate
// launches each is a 16-row-tile kernel of ~29 us for ~10 us of MFMA work per
// workgroup -- launch ramp, tile load, result store and drain are paid per
// launch, and h makes a round trip through HBM in between.  Here ONE 8-wave
// workgroup per 16-row tile runs the `front` chain (+ residual), writes its
// result rows y (the operator's output: the next stage's residual, the
// caller's features) AND keeps them in LDS as the next stage's input.  Every
// layer pass is the same layer_pass_dispatch call on the same operands as in
// rows_mlp_kernel / vertex_pre_edge_kernel, so the results are bit for bit
// those of the separate launches (tested).
struct FrontArgs {
  const float *x;   // [n, ldx], nx valid columns: the front chain's input rows
  int64_t ldx;
  int nx;
  const float *res;  // nullable residual, added to the front chain's output
  int64_t ldres;
  float *y;          // [n, ldy]: front chain's output rows (padded width)
  int64_t ldy;
  int ld_buf;        // floats per row of the LDS tile buffer (max over all passes)
  int ld_stage;      // ... of the stage buffer
  long long *ts;     // profiling stamps (tools/krow_timeline.py) or null:
                     // 16 int64 per workgroup, shader-clock at the phase ends
};

// front chain on the tile (first layer's input already in `tile` with leading
// dimension lds_ld(16 kq0)); its last layer's activated output lands in `stage`
// (`pre`: krow_prefetch of front.l[0]; comes back holding `after`'s, the first
// layer of whatever follows the front chain)
__device__ __forceinline__ void front_chain(const ChainDev &front, float *tile,
                                            float *stage, int ld_stage,
                                            int wave, int lane, KrowPre &pre,
                                            const LayerDev after,
                                            long long *tsw, int &n_stamp) {
  for (int li = 0; li + 1 < front.n; ++li) {
    const LayerDev &L = front.l[li];
    krow_pass(tile, lds_ld(16 * L.kq), tile, lds_ld(16 * L.nt), L, 0, wave,
              lane, pre, true, front.l[li + 1], 0);
    if (tsw && n_stamp < 14) {  // profiling: the front chain layer by layer
      const long long cyc = __builtin_readcyclecounter();
      if (threadIdx.x == 0) tsw[n_stamp] = cyc;
      ++n_stamp;
    }
  }
  const LayerDev &L = front.l[front.n - 1];
  krow_pass(tile, lds_ld(16 * L.kq), stage, ld_stage, L, 0, wave, lane, pre,
            true, after, 0);
}

template <bool STRIDE>
__global__ __launch_bounds__(64 * kRowsWaves) void vertex_update_pre_edge_kernel(
    ChainDev front, FrontArgs f, ChainDev off, LayerDev pl, PreEdgeArgs a) {
  constexpr int ROWS = 16;
  static_assert(64 * kRowsWaves == 32 * ROWS, "one 32-thread group per row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);
  float *scratch = tile + ROWS * f.ld_buf;
  float *stage = scratch + ROWS * a.ld_scratch;
  const int lane_id = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t n_rows = a.n;
  if (a.n_dev) {
    const int64_t nd = *a.n_dev;
    n_rows = nd < n_rows ? nd : n_rows;
  }
  const int kc0 = 16 * front.l[0].kq;
  const int ncols_y = 16 * front.l[front.n - 1].nt;
  const int kc = 16 * pl.kq;  // >= ncols_y (k_in = c + 3 > c)
  long long *tsw = f.ts ? f.ts + (int64_t)blockIdx.x * 16 : nullptr;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (tsw && n_stamp < 15) {
      __builtin_amdgcn_sched_barrier(0);
      const long long cyc = __builtin_readcyclecounter();
      if (threadIdx.x == 0) tsw[n_stamp] = cyc;
      ++n_stamp;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (tsw && threadIdx.x == 0) tsw[15] = __builtin_amdgcn_s_memrealtime();
  for (int64_t row0 = (int64_t)blockIdx.x * ROWS; row0 < n_rows;
       row0 += STRIDE ? (int64_t)gridDim.x * ROWS : n_rows) {
    // (an opaque per-tile copy of the lane id: the per-lane weight addresses
    // of every layer are loop invariant, and hoisted out of the tile loop they
    // cost ~100 VGPRs -- spills in the capacity-form kernels)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_id));
    const int rows_valid =
        (int)((n_rows - row0 < ROWS) ? (n_rows - row0) : ROWS);
    stamp();  // 0: entry
    KrowPre pre;
    krow_prefetch(front.l[0], 0, wave, lane, pre);
    load_rows16(f.x, f.ldx, f.nx, tile, lds_ld(kc0), kc0, row0, rows_valid);
    // lowest() rows of the aggregation buffer the NEXT edge kernel maxes into
    if (a.agg) {
      for (int idx = threadIdx.x; idx < rows_valid * (int)a.ld_agg;
           idx += 64 * kRowsWaves)
        a.agg[row0 * a.ld_agg + idx] = kFloatLowest;
    }
    __syncthreads();
    stamp();  // 1: tile loaded
    front_chain(front, tile, stage, f.ld_stage, wave, lane, pre,
                off.n > 0 ? off.l[0] : pl, tsw, n_stamp);
    stamp();  // front chain done
    // y = stage (+ residual): out to HBM, and [y[:, :c] | x | 0] into the tile
    // -- exactly what vertex_pre_edge_kernel would have read back.  Thread
    // (r, c0) as in consume_rows16; kc <= 320.
    {
      const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
      const int64_t row = row0 + (r < rows_valid ? r : rows_valid - 1);
      float add[kRowSteps];
      if (f.res) {
        const float *__restrict__ rr = f.res + row * f.ldres;
#pragma unroll
        for (int j = 0; j < kRowSteps; ++j) {
          const int c = c0 + 32 * j;
          add[j] = rr[c < ncols_y ? c : ncols_y - 1];
        }
      }
      const float xv0 = a.xyz[row * 3], xv1 = a.xyz[row * 3 + 1],
                  xv2 = a.xyz[row * 3 + 2];
      float *__restrict__ yr = f.y + row * f.ldy;
#pragma unroll
      for (int j = 0; j < kRowSteps; ++j) {
        const int c = c0 + 32 * j;
        if (c < kc) {
          float v = 0.0f;
          if (r < rows_valid) {
            if (c < ncols_y) {
              float sv = stage[r * f.ld_stage + c];
              if (f.res) sv += add[j];
              yr[c] = sv;
              if (c < a.c) v = sv;
            }
            const int d = c - a.c;
            if (d >= 0 && d < 3) v = d == 0 ? xv0 : (d == 1 ? xv1 : xv2);
          }
          tile[r * a.ld_tile + c] = v;
        }
      }
    }
    __syncthreads();
    stamp();  // y written, tile refilled
    pre_edge_tail(off, pl, a, tile, scratch, stage, row0, rows_valid, wave, lane,
                  pre);
    stamp();  // offset chain + Q + P
    if (STRIDE) __syncthreads();
  }
  if (tsw && threadIdx.x == 0) tsw[14] = __builtin_amdgcn_s_memrealtime();
}

// front chain (+ residual) -> y, then a second chain on y -> out (the last
// update MLP and the predictor heads)
template <bool STRIDE>
__global__ __launch_bounds__(64 * kRowsWaves) void vertex_mlp2_kernel(
    ChainDev front, FrontArgs f, ChainDev back, RowsArgs out, int64_t n_rows,
    const int32_t *n_dev) {
  constexpr int ROWS = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *tile = reinterpret_cast<float *>(smem);
  float *stage = tile + ROWS * f.ld_buf;
  const int lane_id = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n_dev) {
    const int64_t nd = *n_dev;
    n_rows = nd < n_rows ? nd : n_rows;
  }
  const int kc0 = 16 * front.l[0].kq;
  const int ncols_y = 16 * front.l[front.n - 1].nt;
  const int kcb = 16 * back.l[0].kq;
  const int ld_b = lds_ld(kcb);
  for (int64_t row0 = (int64_t)blockIdx.x * ROWS; row0 < n_rows;
       row0 += STRIDE ? (int64_t)gridDim.x * ROWS : n_rows) {
    // (an opaque per-tile copy of the lane id: the per-lane weight addresses
    // of every layer are loop invariant, and hoisted out of the tile loop they
    // cost ~100 VGPRs -- spills in the capacity-form kernels)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_id));
    const int rows_valid =
        (int)((n_rows - row0 < ROWS) ? (n_rows - row0) : ROWS);
    KrowPre pre;
    krow_prefetch(front.l[0], 0, wave, lane, pre);
    load_rows16(f.x, f.ldx, f.nx, tile, lds_ld(kc0), kc0, row0, rows_valid);
    __syncthreads();
    int no_stamp = 0;
    front_chain(front, tile, stage, f.ld_stage, wave, lane, pre, back.l[0],
                nullptr, no_stamp);
    // the back chain reads the first out.nx columns of y (its packed weights
    // are zero beyond k_in; rows_mlp_kernel zero-fills the same way)
    const int kmax = kcb > ncols_y ? kcb : ncols_y;  // <= 320
    {
      const int r = threadIdx.x >> 5, c0 = threadIdx.x & 31;
      const int64_t row = row0 + (r < rows_valid ? r : rows_valid - 1);
      float add[kRowSteps];
      if (f.res) {
        const float *__restrict__ rr = f.res + row * f.ldres;
#pragma unroll
        for (int j = 0; j < kRowSteps; ++j) {
          const int c = c0 + 32 * j;
          add[j] = rr[c < ncols_y ? c : ncols_y - 1];
        }
      }
      float *__restrict__ yr = f.y + row * f.ldy;
#pragma unroll
      for (int j = 0; j < kRowSteps; ++j) {
        const int c = c0 + 32 * j;
        if (c < kmax) {
          float v = 0.0f;
          if (r < rows_valid && c < ncols_y) {
            float sv = stage[r * f.ld_stage + c];
            if (f.res) sv += add[j];
            yr[c] = sv;
            if (c < out.nx) v = sv;
          }
          if (c < kcb) tile[r * ld_b + c] = v;
        }
      }
    }
    __syncthreads();
    for (int li = 0; li + 1 < back.n; ++li) {
      const LayerDev &L = back.l[li];
      krow_pass(tile, lds_ld(16 * L.kq), tile, lds_ld(16 * L.nt), L, 0, wave,
                lane, pre, true, back.l[li + 1], 0);
    }
    const LayerDev &L = back.l[back.n - 1];
    const int ld_st = lds_ld(16 * L.nt);
    krow_pass(tile, lds_ld(16 * L.kq), stage, ld_st, L, 0, wave, lane, pre,
              false, L, 0);
    consume_rows16(stage, ld_st, row0, rows_valid, 0, 16 * L.nt, out);
    if (STRIDE) __syncthreads();
  }
}

}  // namespace

namespace {
int pre_edge_impl(
    const float *h, int64_t ld_h, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t n_vertices, float *P,
    float *Q, int64_t ld_pq, float *agg, int64_t ld_agg, hipStream_t stream,
    const Dyn &dk) {
  PGNN_REQUIRE(n_vertices >= 0 && c > 0 && ld_h >= c && n_offset_layers >= 0 &&
                   n_offset_layers <= PGNN_MAX_LAYERS && p_layer,
               PGNN_E_INVALID, "vertex_pre_edge: bad argument");
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(h && xyz && wx && P && Q && (!agg || ld_agg > 0), PGNN_E_INVALID,
               "vertex_pre_edge: null pointer");
  Plan pp;
  int rc = make_plan(p_layer, 1, c + 3, pp);
  if (rc) return rc;
  const LayerDev pl = pp.chain.l[0];
  PGNN_REQUIRE(p_layer->k_in == c + 3 && pl.nt <= kMaxTilesPerPass &&
                   ld_pq == 16 * pl.nt,
               PGNN_E_INVALID,
               "vertex_pre_edge: P layer must be [c+3 -> n], ld_pq = padded n");
  ChainDev off = {};
  int scratch_ld = lds_ld(16);
  if (n_offset_layers > 0) {
    Plan po;
    rc = make_plan(offset_layers, n_offset_layers, c, po);
    if (rc) return rc;
    off = po.chain;
    PGNN_REQUIRE(offset_layers[0].k_in == c &&
                     offset_layers[n_offset_layers - 1].n_out >= 3 &&
                     off.l[0].kq <= pl.kq &&
                     off.l[n_offset_layers - 1].nt <= kMaxTilesPerPass,
                 PGNN_E_INVALID, "vertex_pre_edge: offset chain must be [c -> ... -> 3]");
    for (int i = 0; i < n_offset_layers; ++i) {
      const int ld = lds_ld(16 * off.l[i].nt);
      if (ld > scratch_ld) scratch_ld = ld;
      if (i > 0 && lds_ld(16 * off.l[i].kq) > scratch_ld)
        scratch_ld = lds_ld(16 * off.l[i].kq);
    }
  }
  PreEdgeArgs a;
  a.h = h;
  a.ld_h = ld_h;
  a.c = c;
  a.xyz = xyz;
  a.wx = wx;
  a.n = n_vertices;
  a.n_dev = dk.dev;
  a.P = P;
  a.Q = Q;
  a.ld_pq = ld_pq;
  a.agg = agg;
  a.ld_agg = ld_agg;
  a.ld_tile = lds_ld(16 * pl.kq);
  a.ld_scratch = scratch_ld;
  const size_t lds =
      (size_t)16 * (a.ld_tile + a.ld_scratch + lds_ld(16 * pl.nt)) * 4;
  PGNN_REQUIRE(lds <= 160 * 1024, PGNN_E_UNSUPPORTED,
               "vertex_pre_edge: layer too wide for the LDS tile");
  auto kern = dk.dev ? vertex_pre_edge_kernel<true> : vertex_pre_edge_kernel<false>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)dyn_grid_tiles(dk, n_vertices)),
                     dim3(64 * kRowsWaves), lds, stream, off, pl, a);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

namespace {
// front chain description shared by the two fused entries
int plan_front(const float *x, int64_t ld_x, int32_t nx,
               const pgnn_fc_layer *front_layers, int32_t n_front,
               const float *residual, int64_t ld_res, float *y, int64_t ld_y,
               Plan &pf, FrontArgs &f) {
  PGNN_REQUIRE(x && y && nx > 0 && ld_x >= nx, PGNN_E_INVALID,
               "fused vertex stage: bad front input");
  int rc = make_plan(front_layers, n_front, nx, pf);
  if (rc) return rc;
  const int nt_last = pf.chain.l[n_front - 1].nt;
  PGNN_REQUIRE(nt_last <= kMaxTilesPerPass, PGNN_E_UNSUPPORTED,
               "fused vertex stage: front chain wider than 320");
  PGNN_REQUIRE(ld_y >= 16 * nt_last && (!residual || ld_res >= 16 * nt_last),
               PGNN_E_INVALID,
               "fused vertex stage: ld_y / ld_res < padded front width");
  f.x = x;
  f.ldx = ld_x;
  f.nx = nx;
  f.res = residual;
  f.ldres = ld_res;
  f.y = y;
  f.ldy = ld_y;
  f.ld_buf = pf.tile_floats_per_row;
  f.ld_stage = lds_ld(16 * nt_last);
  f.ts = (long long *)g_mlp_ts;
  return 0;
}

int update_pre_edge_impl(
    const float *x, int64_t ld_x, int32_t nx, const pgnn_fc_layer *front_layers,
    int32_t n_front, const float *residual, int64_t ld_res, float *y,
    int64_t ld_y, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t n_vertices, float *P,
    float *Q, int64_t ld_pq, float *agg, int64_t ld_agg, hipStream_t stream,
    const Dyn &dk) {
  PGNN_REQUIRE(n_vertices >= 0 && c > 0 && n_offset_layers >= 0 &&
                   n_offset_layers <= PGNN_MAX_LAYERS && p_layer,
               PGNN_E_INVALID, "vertex_update_pre_edge: bad argument");
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(xyz && wx && P && Q && (!agg || ld_agg > 0) && agg != x,
               PGNN_E_INVALID, "vertex_update_pre_edge: null / aliased pointer");
  // (the 8-wave 16-row kernels are the small-K form: above ~32 rows per CU the
  // separate launches pick 64-row tiles, which this kernel does not have)
  PGNN_REQUIRE(expected(dk, n_vertices) <= 32 * (int64_t)device_cu_count(),
               PGNN_E_UNSUPPORTED, "vertex_update_pre_edge: too many rows");
  Plan pf;
  FrontArgs f;
  int rc = plan_front(x, ld_x, nx, front_layers, n_front, residual, ld_res, y,
                      ld_y, pf, f);
  if (rc) return rc;
  PGNN_REQUIRE(16 * pf.chain.l[n_front - 1].nt >= c &&
                   16 * pf.chain.l[n_front - 1].nt < c + 16,
               PGNN_E_INVALID,
               "vertex_update_pre_edge: front chain must produce c features");
  Plan pp;
  rc = make_plan(p_layer, 1, c + 3, pp);
  if (rc) return rc;
  const LayerDev pl = pp.chain.l[0];
  PGNN_REQUIRE(p_layer->k_in == c + 3 && pl.nt <= kMaxTilesPerPass &&
                   ld_pq == 16 * pl.nt,
               PGNN_E_INVALID,
               "vertex_update_pre_edge: P layer must be [c+3 -> n], ld_pq = "
               "padded n");
  PGNN_REQUIRE(pl.kq <= kMaxTilesPerPass, PGNN_E_UNSUPPORTED,
               "vertex_update_pre_edge: more than 317 features");
  ChainDev off = {};
  int scratch_ld = lds_ld(16);
  if (n_offset_layers > 0) {
    Plan po;
    rc = make_plan(offset_layers, n_offset_layers, c, po);
    if (rc) return rc;
    off = po.chain;
    PGNN_REQUIRE(offset_layers[0].k_in == c &&
                     offset_layers[n_offset_layers - 1].n_out >= 3 &&
                     off.l[0].kq <= pl.kq &&
                     off.l[n_offset_layers - 1].nt <= kMaxTilesPerPass,
                 PGNN_E_INVALID,
                 "vertex_update_pre_edge: offset chain must be [c -> ... -> 3]");
    for (int i = 0; i < n_offset_layers; ++i) {
      const int ld = lds_ld(16 * off.l[i].nt);
      if (ld > scratch_ld) scratch_ld = ld;
      if (i > 0 && lds_ld(16 * off.l[i].kq) > scratch_ld)
        scratch_ld = lds_ld(16 * off.l[i].kq);
    }
  }
  PreEdgeArgs a;
  a.h = nullptr;  // the tile is filled from the front chain, not from HBM
  a.ld_h = 0;
  a.c = c;
  a.xyz = xyz;
  a.wx = wx;
  a.n = n_vertices;
  a.n_dev = dk.dev;
  a.P = P;
  a.Q = Q;
  a.ld_pq = ld_pq;
  a.agg = agg;
  a.ld_agg = ld_agg;
  a.ld_tile = lds_ld(16 * pl.kq);
  a.ld_scratch = scratch_ld;
  if (f.ld_buf < a.ld_tile) f.ld_buf = a.ld_tile;
  if (f.ld_stage < lds_ld(16 * pl.nt)) f.ld_stage = lds_ld(16 * pl.nt);
  const size_t lds = (size_t)16 * (f.ld_buf + a.ld_scratch + f.ld_stage) * 4;
  PGNN_REQUIRE(lds <= 160 * 1024, PGNN_E_UNSUPPORTED,
               "vertex_update_pre_edge: layers too wide for the LDS tile");
  auto kern = dk.dev ? vertex_update_pre_edge_kernel<true>
                     : vertex_update_pre_edge_kernel<false>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)dyn_grid_tiles(dk, n_vertices)),
                     dim3(64 * kRowsWaves), lds, stream, pf.chain, f, off, pl, a);
  PGNN_HIP(hipGetLastError());
  return 0;
}

int mlp2_impl(const float *x, int64_t ld_x, int32_t nx,
              const pgnn_fc_layer *front_layers, int32_t n_front,
              const float *residual, int64_t ld_res, float *y, int64_t ld_y,
              int32_t ny, const pgnn_fc_layer *back_layers, int32_t n_back,
              float *out, int64_t ld_out, int64_t n_rows, hipStream_t stream,
              const Dyn &dk) {
  PGNN_REQUIRE(n_rows >= 0 && ny > 0, PGNN_E_INVALID, "mlp2: bad sizes");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(out, PGNN_E_INVALID, "mlp2: null output");
  PGNN_REQUIRE(expected(dk, n_rows) <= 32 * (int64_t)device_cu_count(),
               PGNN_E_UNSUPPORTED, "mlp2: too many rows for the 16-row kernel");
  Plan pf;
  FrontArgs f;
  int rc = plan_front(x, ld_x, nx, front_layers, n_front, residual, ld_res, y,
                      ld_y, pf, f);
  if (rc) return rc;
  PGNN_REQUIRE(ny <= 16 * pf.chain.l[n_front - 1].nt, PGNN_E_INVALID,
               "mlp2: ny wider than the front chain's output");
  Plan pb;
  rc = make_plan(back_layers, n_back, ny, pb);
  if (rc) return rc;
  const int nt_out = pb.chain.l[n_back - 1].nt;
  PGNN_REQUIRE(nt_out <= kMaxTilesPerPass &&
                   pb.chain.l[0].kq <= kMaxTilesPerPass,
               PGNN_E_UNSUPPORTED, "mlp2: back chain wider than 320");
  PGNN_REQUIRE(ld_out >= 16 * nt_out, PGNN_E_INVALID,
               "mlp2: ld_out < padded output width");
  if (f.ld_buf < pb.tile_floats_per_row) f.ld_buf = pb.tile_floats_per_row;
  if (f.ld_stage < lds_ld(16 * nt_out)) f.ld_stage = lds_ld(16 * nt_out);
  RowsArgs ro = {};
  ro.nx = ny;  // columns of y the back chain reads
  ro.y = out;
  ro.ldy = ld_out;
  const size_t lds = (size_t)16 * (f.ld_buf + f.ld_stage) * 4;
  PGNN_REQUIRE(lds <= 160 * 1024, PGNN_E_UNSUPPORTED,
               "mlp2: layers too wide for the LDS tile");
  auto kern = dk.dev ? vertex_mlp2_kernel<true> : vertex_mlp2_kernel<false>;
  {
    const int lrc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (lrc) return lrc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)dyn_grid_tiles(dk, n_rows)),
                     dim3(64 * kRowsWaves), lds, stream, pf.chain, f, pb.chain,
                     ro, n_rows, dk.dev);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_vertex_update_pre_edge_fwd(
    const float *x, int64_t ld_x, int32_t nx, const pgnn_fc_layer *front_layers,
    int32_t n_front, const float *residual, int64_t ld_res, float *y,
    int64_t ld_y, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t n_vertices, float *P,
    float *Q, int64_t ld_pq, float *agg, int64_t ld_agg, void *stream_) {
  PGNN_GUARD_BEGIN
  return update_pre_edge_impl(x, ld_x, nx, front_layers, n_front, residual,
                              ld_res, y, ld_y, c, xyz, offset_layers,
                              n_offset_layers, p_layer, wx, n_vertices, P, Q,
                              ld_pq, agg, ld_agg, (hipStream_t)stream_,
                              dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_vertex_update_pre_edge_fwd_dyn(
    const float *x, int64_t ld_x, int32_t nx, const pgnn_fc_layer *front_layers,
    int32_t n_front, const float *residual, int64_t ld_res, float *y,
    int64_t ld_y, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t vertices_cap,
    float *P, float *Q, int64_t ld_pq, float *agg, int64_t ld_agg,
    const pgnn_dyn_count *n_vertices, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_vertices && n_vertices->dev, PGNN_E_INVALID,
               "vertex_update_pre_edge_dyn: null count");
  return update_pre_edge_impl(x, ld_x, nx, front_layers, n_front, residual,
                              ld_res, y, ld_y, c, xyz, offset_layers,
                              n_offset_layers, p_layer, wx, vertices_cap, P, Q,
                              ld_pq, agg, ld_agg, (hipStream_t)stream_,
                              dyn_of(n_vertices));
  PGNN_GUARD_END
}

extern "C" int pgnn_mlp2_fwd(const float *x, int64_t ld_x, int32_t nx,
                             const pgnn_fc_layer *front_layers, int32_t n_front,
                             const float *residual, int64_t ld_res, float *y,
                             int64_t ld_y, int32_t ny,
                             const pgnn_fc_layer *back_layers, int32_t n_back,
                             float *out, int64_t ld_out, int64_t n_rows,
                             void *stream_) {
  PGNN_GUARD_BEGIN
  return mlp2_impl(x, ld_x, nx, front_layers, n_front, residual, ld_res, y,
                   ld_y, ny, back_layers, n_back, out, ld_out, n_rows,
                   (hipStream_t)stream_, dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_mlp2_fwd_dyn(const float *x, int64_t ld_x, int32_t nx,
                                 const pgnn_fc_layer *front_layers,
                                 int32_t n_front, const float *residual,
                                 int64_t ld_res, float *y, int64_t ld_y,
                                 int32_t ny, const pgnn_fc_layer *back_layers,
                                 int32_t n_back, float *out, int64_t ld_out,
                                 int64_t rows_cap, const pgnn_dyn_count *rows,
                                 void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(rows && rows->dev, PGNN_E_INVALID, "mlp2_dyn: null count");
  return mlp2_impl(x, ld_x, nx, front_layers, n_front, residual, ld_res, y,
                   ld_y, ny, back_layers, n_back, out, ld_out, rows_cap,
                   (hipStream_t)stream_, dyn_of(rows));
  PGNN_GUARD_END
}

extern "C" int pgnn_vertex_pre_edge_fwd(
    const float *h, int64_t ld_h, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t n_vertices, float *P,
    float *Q, int64_t ld_pq, float *agg, int64_t ld_agg, void *stream_) {
  PGNN_GUARD_BEGIN
  return pre_edge_impl(h, ld_h, c, xyz, offset_layers, n_offset_layers, p_layer,
                       wx, n_vertices, P, Q, ld_pq, agg, ld_agg,
                       (hipStream_t)stream_, dyn_of(nullptr));
  PGNN_GUARD_END
}

extern "C" int pgnn_vertex_pre_edge_fwd_dyn(
    const float *h, int64_t ld_h, int32_t c, const float *xyz,
    const pgnn_fc_layer *offset_layers, int32_t n_offset_layers,
    const pgnn_fc_layer *p_layer, const float *wx, int64_t vertices_cap,
    float *P, float *Q, int64_t ld_pq, float *agg, int64_t ld_agg,
    const pgnn_dyn_count *n_vertices, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_vertices && n_vertices->dev, PGNN_E_INVALID,
               "vertex_pre_edge_dyn: null count");
  return pre_edge_impl(h, ld_h, c, xyz, offset_layers, n_offset_layers, p_layer,
                       wx, vertices_cap, P, Q, ld_pq, agg, ld_agg,
                       (hipStream_t)stream_, dyn_of(n_vertices));
  PGNN_GUARD_END
}

// Training forward of PointSetPooling: pgnn_point_set_pooling_fwd that also
// writes the four layers' activations (car chain 4-32-64-128-300 on the
// weights-stationary kernel only; PGNN_E_UNSUPPORTED otherwise, nothing done).
extern "C" int pgnn_point_set_pooling_rows_fwd(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t n_edges,
    int32_t num_keypoints, const pgnn_fc_layer *layers, int32_t n_layers,
    int32_t edges_sorted, float *out, int64_t ld_out, float *const *acts_host,
    int64_t ld_last, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_edges >= 0 && num_keypoints >= 0 && n_feat >= 0 && n_feat <= 13 &&
                   layers && acts_host,
               PGNN_E_INVALID, "point_set_pooling_rows: bad sizes");
  if (n_layers != 4) return PGNN_E_UNSUPPORTED;
  Plan p;
  int rc = make_plan(layers, n_layers, n_feat + 3, p);
  if (rc) return rc;
  const int out_cols = 16 * p.chain.l[n_layers - 1].nt;
  PGNN_REQUIRE(out && ld_out >= out_cols && ld_last >= out_cols && ld_last % 4 == 0,
               PGNN_E_INVALID, "point_set_pooling_rows: bad output");
  int cus = stream_cu_count(stream);
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  if (n_edges == 0 || num_keypoints == 0 || !pool_ws_applies(p, n_edges, cus))
    return PGNN_E_UNSUPPORTED;
  PGNN_REQUIRE(point_xyz && keypoint_indices && edges && (n_feat == 0 || point_features),
               PGNN_E_INVALID, "point_set_pooling_rows: null input");
  for (int i = 0; i < 4; ++i)
    PGNN_REQUIRE(acts_host[i] && (uintptr_t)acts_host[i] % 16 == 0, PGNN_E_INVALID,
                 "point_set_pooling_rows: activation buffers must be 16-byte "
                 "aligned");
  if (!(edges_sorted & 2)) {  // bit 1: `out` holds lowest() already
    rc = fill_lowest(out, (int64_t)num_keypoints * ld_out, stream);
    if (rc) return rc;
  }
  PoolArgs pa = {point_features, n_feat, point_xyz, keypoint_indices, edges};
  SegArgs sa = {out, ld_out, num_keypoints, edges_sorted & 1};
  return launch_pool_ws(p, pa, n_edges, sa, cus, nullptr, stream, acts_host,
                        ld_last);
  PGNN_GUARD_END
}

// Training forward of the edge stage: pgnn_edge_mlp_scatter_max_fwd that also
// writes the per-edge output rows (one fused kernel instead of materialising
// H1, a rows GEMM and a standalone scatter-max).  Only the weights-stationary
// kernel has the form; when it does not apply (few edges, other layer shapes)
// the call returns PGNN_E_UNSUPPORTED and changes nothing: the caller runs
// the three separate primitives.
extern "C" int pgnn_edge_mlp_scatter_max_rows_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t n_edges, int32_t num_vertices,
    const pgnn_fc_layer *layer, int32_t edges_sorted, float *out, int64_t ld_out,
    float *rows_out, int64_t ld_rows, float *h1_out, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_edges >= 0 && num_vertices >= 0 && width > 0 && layer,
               PGNN_E_INVALID, "edge_mlp_rows: bad sizes");
  Plan p;
  int rc = make_plan(layer, 1, width, p);
  if (rc) return rc;
  PGNN_REQUIRE(ld_pq == 16 * p.chain.l[0].kq, PGNN_E_INVALID,
               "edge_mlp_rows: ld_pq must equal the padded width");
  const int out_cols = 16 * p.chain.l[0].nt;
  PGNN_REQUIRE(out && ld_out >= out_cols && rows_out && ld_rows >= out_cols &&
                   ld_rows % 4 == 0 && (uintptr_t)rows_out % 16 == 0,
               PGNN_E_INVALID, "edge_mlp_rows: bad output");
  int cus = stream_cu_count(stream);
  if (g_ws_reserve > 0 && cus - g_ws_reserve >= 64) cus -= g_ws_reserve;
  if (n_edges == 0 || num_vertices == 0 || !edge_ws_applies(p, n_edges, cus))
    return PGNN_E_UNSUPPORTED;  // (no message: an expected answer)
  PGNN_REQUIRE(P && Q && edges, PGNN_E_INVALID, "edge_mlp_rows: null input");
  PGNN_REQUIRE(((uintptr_t)P % 16 == 0) && ((uintptr_t)Q % 16 == 0) &&
                   ((uintptr_t)h1_out % 16 == 0),
               PGNN_E_INVALID, "edge_mlp_rows: P/Q/H1 must be 16-byte aligned");
  if (!(edges_sorted & 2)) {  // bit 1: `out` holds lowest() already
    rc = fill_lowest(out, (int64_t)num_vertices * ld_out, stream);
    if (rc) return rc;
  }
  EdgeArgs ea = {P, Q, ld_pq, edges};
  SegArgs sa = {out, ld_out, num_vertices, edges_sorted & 1};
  if (p.chain.l[0].nt == 19)
    return launch_edge_ws<19, 7>(p.chain.l[0], ea, n_edges, sa, cus, nullptr,
                                 stream, rows_out, ld_rows, h1_out);
  return launch_edge_ws<16, 8>(p.chain.l[0], ea, n_edges, sa, cus, nullptr, stream,
                               rows_out, ld_rows, h1_out);
  PGNN_GUARD_END
}

extern "C" int pgnn_offset_apply(const float *xyz, const float *delta,
                                 int64_t ld_delta, int64_t n_rows,
                                 const float *wx, float *xyz_out, float *Q,
                                 int64_t ld_q, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && ld_q >= 3, PGNN_E_INVALID,
               "offset_apply: bad sizes");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(xyz && wx && Q && (!delta || ld_delta >= 3), PGNN_E_INVALID,
               "offset_apply: null input");
  const int64_t total = n_rows * ld_q;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(offset_apply_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     stream, xyz, delta, ld_delta, n_rows, wx, xyz_out, Q, ld_q);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
