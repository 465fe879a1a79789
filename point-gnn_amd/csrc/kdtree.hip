// Device replica of scikit-learn's KDTree build (see kdtree.h for why):
//   BinaryTree._recursive_build      sklearn/neighbors/_binary_tree.pxi.tp:1034
//   find_node_split_dim              ...:598 (first dimension of largest spread)
//   partition_node_indices           sklearn/neighbors/_partition_nodes.pyx
//     = std::nth_element(idx + s, idx + s + n/2, idx + e, (value, index) order)
//   init_node (KDTree)               sklearn/neighbors/_kd_tree.pyx.tp (min/max)
//
// std::nth_element is libstdc++'s introselect: repeat { median-of-3 of
// (first+1, mid, last-1) moved to `first`; Hoare partition of (first, last)
// around it; keep the side holding nth } until <= 3 elements remain, then
// insertion sort.  With a strict total order the Hoare partition has a closed
// form that parallelises: let G = positions holding an element greater than
// the pivot, ascending, and S = positions holding a smaller one, descending;
// the sequential two-pointer loop swaps G[k] with S[k] for every k with
// G[k] < S[k] and nothing else (tested against KDTree.get_arrays()).
//
// One launch per tree level, one workgroup per node.  The data are a few
// hundred KB (L2-resident); this is latency-bound integer work.
#include "kdtree.h"
#include "pgnn_common.h"

namespace pgnn {
namespace {

// One tree slot: the point and its index travel together (16 B), so a
// partition pass is one coalesced load per element and the key is rec.c[dim].
struct __attribute__((aligned(16))) Rec {
  float c[3];
  int32_t idx;
};
typedef float kd_v4 __attribute__((ext_vector_type(4)));

#ifndef PGNN_KD_NT
#define PGNN_KD_NT 1024
#endif
#ifndef PGNN_KD_LDS_CAP
#define PGNN_KD_LDS_CAP 5120
#endif
#ifndef PGNN_KD_BATCH
#define PGNN_KD_BATCH 2048
#endif
constexpr int KD_NT = PGNN_KD_NT;  // threads per workgroup
constexpr int KD_NW = KD_NT / 64;
constexpr int KD_LDS_CAP = PGNN_KD_LDS_CAP;  // records a workgroup keeps in LDS (80 KB)
constexpr int KD_BATCH = PGNN_KD_BATCH;  // swap-list entries per batch, block mode
constexpr int KD_WBATCH = 256;     // ... per wave, wave mode (same storage)
#ifndef PGNN_KD_WAVE_TAIL
#define PGNN_KD_WAVE_TAIL 512
#endif
#ifndef PGNN_KD_TOP_LEN
#define PGNN_KD_TOP_LEN 640
#endif
constexpr int KD_WAVE_TAIL = PGNN_KD_WAVE_TAIL;  // block mode hands ranges this short to wave 0
constexpr int KD_WAVE_NODE = 1536; // subtree nodes this short get one wave each
// levels whose nodes are longer than this get one launch each (one workgroup
// per node); the rest of the tree is one launch with a workgroup per subtree
constexpr int KD_TOP_LEN = PGNN_KD_TOP_LEN;
static_assert(KD_TOP_LEN <= KD_LDS_CAP, "a subtree must fit the LDS buffer");
constexpr int KD_MASK_WORDS = 8;   // chunk <= 512 elements per thread
constexpr int64_t KD_MAX_POINTS = (int64_t)KD_NT * 64 * KD_MASK_WORDS;

// (value, index) strict total order of IndexComparator
__device__ __forceinline__ bool kd_less(float va, int ia, float vb, int ib) {
  return va == vb ? ia < ib : va < vb;
}

__device__ __forceinline__ void rec_swap(Rec *a, Rec *b) {
  const kd_v4 x = *reinterpret_cast<kd_v4 *>(a);
  const kd_v4 y = *reinterpret_cast<kd_v4 *>(b);
  *reinterpret_cast<kd_v4 *>(a) = y;
  *reinterpret_cast<kd_v4 *>(b) = x;
}

// synchronisation of the threads that cooperate on one nth_element: the
// whole workgroup (barrier) or a single wave (ordering only)
template <bool WAVE>
__device__ __forceinline__ void kd_sync() {
  if (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// exclusive scan of two counters over the cooperating threads + totals
template <bool WAVE>
__device__ __forceinline__ void kd_scan2(int a, int b, int &ea, int &eb,
                                         int &ta, int &tb, int *sh) {
  const int lane = threadIdx.x & 63;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int xa = __shfl_up(ia, d), xb = __shfl_up(ib, d);
    if (lane >= d) {
      ia += xa;
      ib += xb;
    }
  }
  if (WAVE) {
    ea = ia - a;
    eb = ib - b;
    ta = __shfl(ia, 63);
    tb = __shfl(ib, 63);
    return;
  }
  const int wave = threadIdx.x >> 6;
  if (lane == 63) {
    sh[wave] = ia;
    sh[KD_NW + wave] = ib;
  }
  __syncthreads();
  int oa = 0, ob = 0, sa = 0, sb = 0;
#pragma unroll
  for (int w = 0; w < KD_NW; ++w) {
    const int va = sh[w], vb = sh[KD_NW + w];
    if (w < wave) {
      oa += va;
      ob += vb;
    }
    sa += va;
    sb += vb;
  }
  ea = oa + ia - a;
  eb = ob + ib - b;
  ta = sa;
  tb = sb;
  __syncthreads();  // sh[] is reused by the next call
}

// libstdc++ std::__adjust_heap + __push_heap (bits/stl_heap.h) on the max-heap
// H[0 .. len) under kd_less; one thread.
__device__ inline void kd_adjust_heap(Rec *H, int dim, int hole, int len,
                                      const Rec value) {
  const float vk = value.c[dim];
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (kd_less(H[child].c[dim], H[child].idx, H[child - 1].c[dim],
                H[child - 1].idx))
      --child;
    H[hole] = H[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    H[hole] = H[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && kd_less(H[parent].c[dim], H[parent].idx, vk, value.idx)) {
    H[hole] = H[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  H[hole] = value;
}

// What __introselect does once its depth limit (2 * floor(log2 n) partition
// rounds) is used up: std::__heap_select(first, nth + 1, last) -- a max-heap
// of [first, nth] swallows every later element smaller than its top -- then
// iter_swap(first, nth).  Rare (about one cloud in fifty has one such node)
// and short (the range left over is a handful of elements), so ONE thread
// replays libstdc++'s exact sequence of moves; the caller synchronises.
__device__ inline void kd_heap_select(Rec *R, int off, int dim, int first,
                                      int nth, int last) {
  Rec *H = R + (first - off);
  const int len = nth + 1 - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const Rec value = H[parent];
      kd_adjust_heap(H, dim, parent, len, value);
      if (parent == 0) break;
      --parent;
    }
  }
  for (int i = nth + 1; i < last; ++i) {
    Rec *e = R + (i - off);
    if (kd_less(e->c[dim], e->idx, H[0].c[dim], H[0].idx)) {  // __pop_heap
      const Rec value = *e;
      *e = H[0];
      kd_adjust_heap(H, dim, 0, len, value);
    }
  }
  rec_swap(&R[first - off], &R[nth - off]);
}

// libstdc++ __introselect iterations on slots [first, last) -- slot i lives at
// R[i - off] (R: LDS or global) -- executed by G = (WAVE ? 64 : KD_NT) threads
// with ids `tid`, until the range is <= max(3, stop_len) long.  lg/ls: swap
// lists of BATCH entries each that only these threads touch.
template <bool WAVE, int BATCH>
__device__ __forceinline__ void kd_introselect(Rec *R, int off, int dim,
                                               int &first, int &last, int nth,
                                               int &depth, int stop_len, int tid,
                                               int32_t *lg, int32_t *ls, int *sh,
                                               int32_t *status) {
  constexpr int G = WAVE ? 64 : KD_NT;
  while (last - first > 3 && last - first > stop_len) {
    if (depth == 0) {  // libstdc++ switches to heap-select here
      if (tid == 0) kd_heap_select(R, off, dim, first, nth, last);
      kd_sync<WAVE>();
      last = first;  // nothing left for the caller to do (no final sort)
      return;
    }
    --depth;
    {  // __move_median_to_first(first, first+1, mid, last-1): three loads in
       // flight at once (lanes 0..2), decision on lane 0
      const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
      if (tid < 64) {
        const int my = tid == 0 ? a : (tid == 1 ? b : c);
        float k = 0.f;
        int ix = 0;
        if (tid < 3) {
          k = R[my - off].c[dim];
          ix = R[my - off].idx;
        }
        const float ka = __shfl(k, 0), kb = __shfl(k, 1), kc = __shfl(k, 2);
        const int ia = __shfl(ix, 0), ib = __shfl(ix, 1), ic = __shfl(ix, 2);
        if (tid == 0) {
          int m;
          if (kd_less(ka, ia, kb, ib)) {
            if (kd_less(kb, ib, kc, ic)) m = b;
            else if (kd_less(ka, ia, kc, ic)) m = c;
            else m = a;
          } else if (kd_less(ka, ia, kc, ic)) m = a;
          else if (kd_less(kb, ib, kc, ic)) m = c;
          else m = b;
          rec_swap(&R[first - off], &R[m - off]);
        }
      }
    }
    kd_sync<WAVE>();
    const float pk = R[first - off].c[dim];
    const int pi = R[first - off].idx;
    // classify (first, last): every thread owns a contiguous chunk
    const int r0 = first + 1, rlen = last - r0;
    const int chunk = (rlen + G - 1) / G;
    const int c0 = min(r0 + tid * chunk, last);
    const int c1 = min(c0 + chunk, last);
    int nl = 0, ng = 0;
    // bit j of the KD_MASK_WORDS-word mask: element c0+j is smaller.  The
    // classification must be remembered: after the first batch of swaps the
    // records no longer tell which side an element started on.
    unsigned long long mask[KD_MASK_WORDS];
#pragma unroll
    for (int w = 0; w < KD_MASK_WORDS; ++w) {
      unsigned long long mw = 0;
      const int w0 = c0 + 64 * w, w1 = min(w0 + 64, c1);
#pragma unroll 4
      for (int i = w0; i < w1; ++i) {
        const float k = R[i - off].c[dim];
        const bool less = (k == pk) ? (R[i - off].idx < pi) : (k < pk);
        nl += less ? 1 : 0;
        ng += less ? 0 : 1;
        mw |= less ? (1ull << (i - w0)) : 0ull;
      }
      mask[w] = mw;
    }
    int ol, og, c_less, c_gr;
    kd_scan2<WAVE>(nl, ng, ol, og, c_less, c_gr, sh);
    const int cut = r0 + c_less;
    // Hoare partition in closed form: the k-th greater element from the left
    // is swapped with the k-th smaller element from the right while the former
    // lies left of the latter.  Lists pass through LDS, BATCH ranks at a time.
    const int pairs = min(c_less, c_gr);
    for (int b0 = 0; b0 < pairs; b0 += BATCH) {
      int rl = ol, rg = og;
#pragma unroll
      for (int w = 0; w < KD_MASK_WORDS; ++w) {
        const unsigned long long mw = mask[w];
        const int w0 = c0 + 64 * w, w1 = min(w0 + 64, c1);
        for (int i = w0; i < w1; ++i) {
          if ((mw >> (i - w0)) & 1ull) {
            const int k = c_less - 1 - rl;  // rank from the right
            if (k >= b0 && k < b0 + BATCH) ls[k - b0] = i;
            ++rl;
          } else {
            if (rg >= b0 && rg < b0 + BATCH) lg[rg - b0] = i;
            ++rg;
          }
        }
      }
      kd_sync<WAVE>();
      const int nb = min(BATCH, pairs - b0);
      const bool more = lg[nb - 1] < ls[nb - 1];  // uniform: batch fully swaps
      for (int k = tid; k < nb; k += G) {
        const int g = lg[k], sp = ls[k];
        if (g < sp) rec_swap(&R[g - off], &R[sp - off]);
      }
      kd_sync<WAVE>();
      if (!more) break;  // later ranks are already on their side
    }
    if (cut <= nth) first = cut; else last = cut;
  }
}

// Wave-synchronous introselect (one wave, ranges <= 64 * KD_WCHUNK): lane l
// owns elements r0 + 64*j + l, so "position order" is (j, lane) and every
// rank comes from ballots + mbcnt -- no shuffles, no barriers.
constexpr int KD_WCHUNK = 24;  // >= KD_WAVE_NODE / 64
__device__ __forceinline__ void kd_introselect_wave(Rec *R, int off, int dim,
                                                    int &first, int &last,
                                                    int nth, int &depth,
                                                    int lane, int32_t *lg,
                                                    int32_t *ls,
                                                    int32_t *status) {
  while (last - first > 3) {
    if (depth == 0) {  // libstdc++ switches to heap-select here
      if (lane == 0) kd_heap_select(R, off, dim, first, nth, last);
      kd_sync<true>();
      last = first;
      return;
    }
    --depth;
    {  // __move_median_to_first(first, first+1, mid, last-1)
      const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
      const int my = lane == 0 ? a : (lane == 1 ? b : c);
      float k = 0.f;
      int ix = 0;
      if (lane < 3) {
        k = R[my - off].c[dim];
        ix = R[my - off].idx;
      }
      const float ka = __builtin_bit_cast(
          float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k), 0));
      const float kb = __builtin_bit_cast(
          float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k), 1));
      const float kc = __builtin_bit_cast(
          float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, k), 2));
      const int ia = __builtin_amdgcn_readlane(ix, 0);
      const int ib = __builtin_amdgcn_readlane(ix, 1);
      const int ic = __builtin_amdgcn_readlane(ix, 2);
      int m;  // wave-uniform
      if (kd_less(ka, ia, kb, ib)) {
        if (kd_less(kb, ib, kc, ic)) m = b;
        else if (kd_less(ka, ia, kc, ic)) m = c;
        else m = a;
      } else if (kd_less(ka, ia, kc, ic)) m = a;
      else if (kd_less(kb, ib, kc, ic)) m = c;
      else m = b;
      if (lane == 0) rec_swap(&R[first - off], &R[m - off]);
    }
    kd_sync<true>();
    const float pk = R[first - off].c[dim];
    const int pi = R[first - off].idx;
    const int r0 = first + 1, rlen = last - r0;
    const int nj = (rlen + 63) >> 6;  // <= KD_WCHUNK
    unsigned mask = 0;  // bit j: my element of row j is smaller than the pivot
    int c_less = 0;
    // (unconditional loads from a clamped slot, the range test on the value:
    // a load under `if (i < last)` is waited for on the spot and the rows of a
    // pass would be read one LDS round trip after the other)
#pragma unroll 4
    for (int j = 0; j < nj; ++j) {
      const int i = r0 + 64 * j + lane;
      const Rec &e = R[(i < last ? i : last - 1) - off];
      const float k = e.c[dim];
      const int ei = e.idx;
      const bool less = i < last && ((k == pk) ? (ei < pi) : (k < pk));
      mask |= less ? (1u << j) : 0u;
      c_less += __popcll(__ballot(less));
    }
    const int c_gr = rlen - c_less;
    const int cut = r0 + c_less;
    const int pairs = min(c_less, c_gr);
    for (int b0 = 0; b0 < pairs; b0 += KD_WBATCH) {
      int sl = 0, sg = 0;  // smaller / greater elements in earlier rows
      for (int j = 0; j < nj; ++j) {
        const int i = r0 + 64 * j + lane;
        const bool valid = i < last;
        const bool less = (mask >> j) & 1u;
        const unsigned long long bl = __ballot(less);
        const unsigned long long bg = __ballot(valid && !less);
        if (valid) {
          if (less) {
            const int k = c_less - 1 - (sl + (int)__builtin_amdgcn_mbcnt_hi(
                (unsigned)(bl >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bl, 0)));
            if (k >= b0 && k < b0 + KD_WBATCH) ls[k - b0] = i;
          } else {
            const int k = sg + (int)__builtin_amdgcn_mbcnt_hi(
                (unsigned)(bg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bg, 0));
            if (k >= b0 && k < b0 + KD_WBATCH) lg[k - b0] = i;
          }
        }
        sl += __popcll(bl);
        sg += __popcll(bg);
      }
      kd_sync<true>();
      const int nb = min(KD_WBATCH, pairs - b0);
      const bool more = lg[nb - 1] < ls[nb - 1];
      for (int k = lane; k < nb; k += 64) {
        const int g = lg[k], sp = ls[k];
        if (g < sp) rec_swap(&R[g - off], &R[sp - off]);
      }
      kd_sync<true>();
      if (!more) break;
    }
    if (cut <= nth) first = cut; else last = cut;
  }
}

// __insertion_sort of the <= 3 survivors (one thread)
__device__ __forceinline__ void kd_final_sort(Rec *R, int off, int dim, int first,
                                              int last) {
  for (int i = first + 1; i < last; ++i) {
    int j = i;  // bubble element i down: swaps keep everything in memory
    while (j > first && kd_less(R[j - off].c[dim], R[j - off].idx,
                                R[j - 1 - off].c[dim], R[j - 1 - off].idx)) {
      rec_swap(&R[j - off], &R[j - 1 - off]);
      --j;
    }
  }
}

// slot range of node (level, index j within the level): halve from the root
__device__ __forceinline__ void kd_node_range(int n, int level, int j, int &s,
                                              int &e) {
  s = 0;
  e = n;
  for (int b = level - 1; b >= 0; --b) {
    const int m = s + (e - s) / 2;
    if ((j >> b) & 1) s = m; else e = m;
  }
}

// first dimension of largest spread (find_node_split_dim)
__device__ __forceinline__ int kd_split_dim(const float (&lo)[3],
                                            const float (&hi)[3]) {
  int dim = 0;
  double best = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double spread = (double)hi[j] - (double)lo[j];
    if (spread > best) {
      best = spread;
      dim = j;
    }
  }
  return dim;
}

__global__ void kd_init_kernel(const float *__restrict__ pts, int n,
                               Rec *__restrict__ rec, int32_t *status) {
  graph_prio();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    Rec r;
    r.c[0] = pts[3 * (size_t)i];
    r.c[1] = pts[3 * (size_t)i + 1];
    r.c[2] = pts[3 * (size_t)i + 2];
    r.idx = i;
    rec[i] = r;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
}

// One workgroup per node of `level`, nodes longer than KD_LDS_CAP: the first
// introselect iterations run on the global records; as soon as the live range
// fits, it moves to LDS and finishes there.
__global__ __launch_bounds__(KD_NT) void kd_top_kernel(
    Rec *rec, int n, int level, int n_nodes, double *__restrict__ bounds,
    int32_t *__restrict__ status) {
  graph_prio();
  __shared__ float red[6][KD_NW];
  __shared__ int sh[2 * KD_NW];
  __shared__ int32_t lg[KD_BATCH], ls[KD_BATCH];
  extern __shared__ __attribute__((aligned(16))) char kd_dyn[];
  Rec *L = reinterpret_cast<Rec *>(kd_dyn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a workgroup takes the level's nodes blockIdx.x, + gridDim.x, ... (the
  // frame pipeline caps the builder's grids, graph_max_wgs)
  for (int jn = blockIdx.x; jn < (1 << level); jn += gridDim.x) {
  __syncthreads();  // LDS of the previous node is free
  const int node = (1 << level) - 1 + jn;
  if (node >= n_nodes) continue;
  int s, e;
  kd_node_range(n, level, jn, s, e);
  const int len = e - s;
  float lo[3] = {INFINITY, INFINITY, INFINITY};
  float hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = s + threadIdx.x; i < e; i += KD_NT) {
    const kd_v4 r = *reinterpret_cast<const kd_v4 *>(&rec[i]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      lo[j] = fminf(lo[j], r[j]);
      hi[j] = fmaxf(hi[j], r[j]);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      lo[j] = fminf(lo[j], __shfl_xor(lo[j], d));
      hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], d));
    }
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      red[j][wave] = lo[j];
      red[3 + j][wave] = hi[j];
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    lo[j] = red[j][0];
    hi[j] = red[3 + j][0];
#pragma unroll
    for (int w = 1; w < KD_NW; ++w) {
      lo[j] = fminf(lo[j], red[j][w]);
      hi[j] = fmaxf(hi[j], red[3 + j][w]);
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bounds[6 * (size_t)node + j] = (double)lo[j];
      bounds[6 * (size_t)node + 3 + j] = (double)hi[j];
    }
  }
  if (2 * node + 1 >= n_nodes || len < 2) continue;  // leaf
  const int dim = kd_split_dim(lo, hi);
  // ---- std::nth_element(idx + s, idx + s + len/2, idx + e)
  int first = s, last = e;
  const int nth = s + len / 2;
  int depth = 2 * (31 - __clz(len));
  kd_introselect<false, KD_BATCH>(rec, 0, dim, first, last, nth, depth,
                                  KD_LDS_CAP, threadIdx.x, lg, ls, sh, status);
  const int f0 = first, l0 = last;  // live range, now <= KD_LDS_CAP: to LDS
  for (int i = f0 + threadIdx.x; i < l0; i += KD_NT)
    *reinterpret_cast<kd_v4 *>(&L[i - f0]) =
        *reinterpret_cast<const kd_v4 *>(&rec[i]);
  __syncthreads();
  kd_introselect<false, KD_BATCH>(L, f0, dim, first, last, nth, depth,
                                  KD_WAVE_TAIL, threadIdx.x, lg, ls, sh, status);
  if (wave == 0) {
    kd_introselect_wave(L, f0, dim, first, last, nth, depth, lane, lg, ls,
                        status);
    if (lane == 0 && last - first > 1 && last - first <= 3)
      kd_final_sort(L, f0, dim, first, last);
  }
  __syncthreads();
  for (int i = f0 + threadIdx.x; i < l0; i += KD_NT)
    *reinterpret_cast<kd_v4 *>(&rec[i]) =
        *reinterpret_cast<const kd_v4 *>(&L[i - f0]);
  }
}

// One workgroup per node of `level0` (<= KD_LDS_CAP records): the whole
// subtree below it is built in LDS, one WAVE per node, a workgroup barrier
// per level; then idx_array and its inverse are written out.
__global__ __launch_bounds__(KD_NT) void kd_subtree_kernel(
    const Rec *__restrict__ rec, int n, int level0, int n_levels, int n_nodes,
    double *__restrict__ bounds, int32_t *__restrict__ idx_out,
    int32_t *__restrict__ pos_out, int32_t *__restrict__ status) {
  graph_prio();
  __shared__ int32_t lists[KD_NW][2][KD_WBATCH];  // 32 KB >= 2 * KD_BATCH ints
  __shared__ float red[6][KD_NW];
  __shared__ int sh[2 * KD_NW];
  static_assert(KD_NW * 2 * KD_WBATCH >= 2 * KD_BATCH, "list storage");
  static_assert(64 * KD_WCHUNK >= KD_WAVE_NODE && KD_WAVE_NODE >= KD_WAVE_TAIL,
                "wave-mode ranges");
  extern __shared__ __attribute__((aligned(16))) char kd_dyn[];
  Rec *L = reinterpret_cast<Rec *>(kd_dyn);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a workgroup takes the subtrees blockIdx.x, + gridDim.x, ...
  for (int jb = blockIdx.x; jb < (1 << level0); jb += gridDim.x) {
  __syncthreads();  // LDS of the previous subtree is free
  int s0, e0;
  kd_node_range(n, level0, jb, s0, e0);
  for (int i = s0 + threadIdx.x; i < e0; i += KD_NT)
    *reinterpret_cast<kd_v4 *>(&L[i - s0]) =
        *reinterpret_cast<const kd_v4 *>(&rec[i]);
  __syncthreads();
  for (int lv = level0; lv < n_levels; ++lv) {
    const int dl = lv - level0;
    const int cnt = 1 << dl;  // nodes of level lv inside this subtree
    // few long nodes: the whole workgroup takes them one after the other
    // (block mode); many short nodes: one wave each
    const bool block_mode = ((e0 - s0 + cnt - 1) >> dl) > KD_WAVE_NODE;
    for (int j = block_mode ? 0 : wave; j < cnt; j += block_mode ? 1 : KD_NW) {
      const int jl = (jb << dl) + j;  // index within level lv
      const int node = (1 << lv) - 1 + jl;
      if (node >= n_nodes) continue;
      int s, e;
      kd_node_range(n, lv, jl, s, e);
      float lo[3] = {INFINITY, INFINITY, INFINITY};
      float hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      const int t = block_mode ? threadIdx.x : lane;
      const int nt = block_mode ? KD_NT : 64;
      for (int i = s + t; i < e; i += nt) {
        const kd_v4 r = *reinterpret_cast<const kd_v4 *>(&L[i - s0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          lo[q] = fminf(lo[q], r[q]);
          hi[q] = fmaxf(hi[q], r[q]);
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          lo[q] = fminf(lo[q], __shfl_xor(lo[q], d));
          hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], d));
        }
      if (block_mode) {
        if (lane == 0)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            red[q][wave] = lo[q];
            red[3 + q][wave] = hi[q];
          }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          lo[q] = red[q][0];
          hi[q] = red[3 + q][0];
#pragma unroll
          for (int w = 1; w < KD_NW; ++w) {
            lo[q] = fminf(lo[q], red[q][w]);
            hi[q] = fmaxf(hi[q], red[3 + q][w]);
          }
        }
        __syncthreads();  // red[] is reused by the next node
      }
      if (t == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          bounds[6 * (size_t)node + q] = (double)lo[q];
          bounds[6 * (size_t)node + 3 + q] = (double)hi[q];
        }
      }
      const int len = e - s;
      if (2 * node + 1 >= n_nodes || len < 2) continue;  // leaf
      const int dim = kd_split_dim(lo, hi);
      int first = s, last = e;
      const int nth = s + len / 2;
      int depth = 2 * (31 - __clz(len));
      if (block_mode) {
        kd_introselect<false, KD_BATCH>(L, s0, dim, first, last, nth, depth,
                                        KD_WAVE_TAIL, threadIdx.x, &lists[0][0][0],
                                        &lists[0][0][0] + KD_BATCH, sh, status);
        if (wave == 0) {
          kd_introselect_wave(L, s0, dim, first, last, nth, depth, lane,
                              &lists[0][0][0], &lists[0][0][0] + KD_BATCH,
                              status);
          if (lane == 0 && last - first > 1 && last - first <= 3)
            kd_final_sort(L, s0, dim, first, last);
        }
        __syncthreads();
      } else {
        kd_introselect_wave(L, s0, dim, first, last, nth, depth, lane,
                            lists[wave][0], lists[wave][1], status);
        if (lane == 0 && last - first > 1 && last - first <= 3)
          kd_final_sort(L, s0, dim, first, last);
      }
    }
    __syncthreads();
  }
  for (int i = s0 + threadIdx.x; i < e0; i += KD_NT) {
    const int p = L[i - s0].idx;
    idx_out[i] = p;
    pos_out[p] = i;
  }
  }
}

}  // namespace

int kd_build(const float *pts, int64_t n, Arena &a, KdBuild &kd,
             hipStream_t stream);

size_t kd_workspace_bytes(int64_t n) {
  int lv, nodes;
  kd_shape(n, &lv, &nodes);
  const size_t nn = (size_t)(n > 0 ? n : 1);
  return 2 * align_up(nn * 4, 256) + align_up(nn * 16, 256) +
         align_up((size_t)nodes * 6 * 8, 256) + 256 + 2048;
}

// carve + build on `stream`; arrays live in the caller's arena
int kd_build(const float *pts, int64_t n, Arena &a, KdBuild &kd,
             hipStream_t stream) {
  PGNN_REQUIRE(n >= 0, PGNN_E_INVALID, "kdtree: negative point count");
  PGNN_REQUIRE(n <= KD_MAX_POINTS, PGNN_E_UNSUPPORTED,
               "kdtree: more than 524288 points are not supported (the replica "
               "of scikit-learn's KDTree keeps node records in LDS)");
  static_assert(KD_MAX_POINTS == 524288 || KD_NT != 1024,
                "keep the message in step");
  kd_shape(n, &kd.n_levels, &kd.n_nodes);
  const size_t nn = (size_t)(n > 0 ? n : 1);
  kd.idx = a.take<int32_t>(nn);
  kd.pos = a.take<int32_t>(nn);
  Rec *rec = a.take<Rec>(nn);
  kd.bounds = a.take<double>((size_t)kd.n_nodes * 6);
  kd.status = a.take<int32_t>(1);
  PGNN_REQUIRE(kd.idx && kd.pos && rec && kd.bounds && kd.status,
               PGNN_E_WORKSPACE, "kdtree: workspace too small");
  if (n == 0) {
    PGNN_HIP(hipMemsetAsync(kd.status, 0, 4, stream));
    return 0;
  }
  const size_t dyn = (size_t)KD_LDS_CAP * sizeof(Rec);
  static bool attr_set = false;
  if (!attr_set) {  // 80 KB dynamic + 16-32 KB static LDS per workgroup
    PGNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kd_top_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)dyn));
    PGNN_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void *>(kd_subtree_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    attr_set = true;
  }
  hipLaunchKernelGGL(kd_init_kernel, dim3(graph_grid((n + 255) / 256)), dim3(256),
                     graph_lds_pad(), stream, pts, (int)n, rec, kd.status);
  // levels whose nodes can exceed the LDS capacity: one launch each
  int level0 = 0;
  while (level0 < kd.n_levels - 1 &&
         ((n + ((int64_t)1 << level0) - 1) >> level0) > KD_TOP_LEN) {
    hipLaunchKernelGGL(kd_top_kernel, dim3(graph_grid(1 << level0)), dim3(KD_NT), dyn,
                       stream, rec, (int)n, level0, kd.n_nodes, kd.bounds,
                       kd.status);
    ++level0;
  }
  PGNN_REQUIRE(((n + ((int64_t)1 << level0) - 1) >> level0) <= KD_LDS_CAP,
               PGNN_E_INVALID, "kdtree: leaf larger than the LDS capacity");
  hipLaunchKernelGGL(kd_subtree_kernel, dim3(graph_grid(1 << level0)), dim3(KD_NT), dyn,
                     stream, rec, (int)n, level0, kd.n_levels, kd.n_nodes,
                     kd.bounds, kd.idx, kd.pos, kd.status);
  PGNN_HIP(hipGetLastError());
  return 0;
}

}  // namespace pgnn

using namespace pgnn;

extern "C" int pgnn_kdtree_shape(int64_t n_points, int32_t *n_levels,
                                 int32_t *n_nodes) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_points >= 0 && n_levels && n_nodes, PGNN_E_INVALID,
               "kdtree_shape: bad argument");
  int lv, nodes;
  kd_shape(n_points, &lv, &nodes);
  *n_levels = lv;
  *n_nodes = nodes;
  return 0;
  PGNN_GUARD_END
}

extern "C" size_t pgnn_kdtree_workspace_bytes(int64_t n_points) {
  if (n_points < 0) return 0;
  return kd_workspace_bytes(n_points);
}

extern "C" int pgnn_kdtree_replica(const float *points, int64_t n_points,
                                   void *workspace, size_t workspace_bytes,
                                   int32_t *idx_array, double *node_bounds,
                                   int32_t *status, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_points >= 0 && idx_array && node_bounds && status,
               PGNN_E_INVALID, "kdtree_replica: bad argument");
  PGNN_REQUIRE(n_points == 0 || points, PGNN_E_INVALID,
               "kdtree_replica: null points");
  PGNN_REQUIRE(workspace && workspace_bytes >= kd_workspace_bytes(n_points),
               PGNN_E_WORKSPACE, "kdtree_replica: workspace too small");
  Arena a(workspace, workspace_bytes);
  KdBuild kd;
  int rc = kd_build(points, n_points, a, kd, stream);
  if (rc) return rc;
  if (n_points > 0)
    PGNN_HIP(hipMemcpyAsync(idx_array, kd.idx, (size_t)n_points * 4,
                            hipMemcpyDeviceToDevice, stream));
  PGNN_HIP(hipMemcpyAsync(node_bounds, kd.bounds, (size_t)kd.n_nodes * 48,
                          hipMemcpyDeviceToDevice, stream));
  PGNN_HIP(hipMemcpyAsync(status, kd.status, 4, hipMemcpyDeviceToDevice,
                          stream));
  return 0;
  PGNN_GUARD_END
}
