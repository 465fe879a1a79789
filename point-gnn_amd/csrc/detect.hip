// Detection post-processing, the step right after the GNN:
// class-aware box decode/encode (box_encoding.py:231-299), candidate selection
// (run.py:264-290) and rotated-box NMS with median merge and score
// accumulation (nms.py:9-27,64-88,90-300).
//
// The reference walks the score-sorted boxes in a Python loop and calls shapely
// for every polygon pair.  Here:
//   1. a radix sort orders boxes by descending score;
//   2. one thread per box builds its float64 corner geometry exactly the way
//      boxes_3d_to_corners does (float32 cos/sin and half sizes, float64 rest);
//   3. one workgroup per 64 x 64 block of pairs (bounds and labels staged in
//      LDS) evaluates the pairwise test "same class and overlap > threshold"
//      for j > i and stores it as a bit matrix (ballot), so the part that is
//      quadratic runs on the whole chip;
//   4. the reference's loop is sequential only through the keep flags: ONE
//      WAVE replays it on the bit matrix (removed set = row & keep, keep &=
//      ~removed) and writes every surviving row back masked;
//   5. the per-row arithmetic depends only on that removed set and on
//      untouched inputs, so it runs for all survivors in parallel, a workgroup
//      each: per-component median by rank counting, overlap of the merged box
//      with the removed ones, float64 score sum in a fixed-shape tree;
//   6. ordered compaction of the survivors.
// Everything is HBM/latency-bound integer and float64 work; no MFMA.
#include "pgnn_common.h"
#include "sort.h"

namespace pgnn {
namespace {

// the sequential scan is barrier-latency bound: four waves keep a workgroup
// barrier cheap (1024 threads: 1.65 ms for 3.5k candidates; 256: see DESIGN.md)
constexpr int kScanThreads = 256;
constexpr int kListCap = 1024;  // removed-set entries kept in LDS

struct Geom {
  double px[4], pz[4];  // BEV polygon (corners 0-3, x and z)
  double lo[3], hi[3];  // axis-aligned bounds over the 8 corners
  double area;          // |polygon area|
};

// nms.py:9-27.  cos/sin, l/2, w/2 and -h are float32 values (NumPy float32
// scalars); the rotation and translation happen in float64.  appr > 0
// reproduces bboxes_nms' `np.int32(corners * appr_factor)` (nms.py:113-115).
__device__ Geom box_geometry(const float *b, double appr) {
  const float yaw = b[6];
  const double c = (double)(float)cos((double)yaw);
  const double s = (double)(float)sin((double)yaw);
  const double hl = (double)(b[3] / 2.0f), hw = (double)(b[5] / 2.0f);
  const double nh = (double)(-b[4]);
  const double cx[4] = {hl, hl, -hl, -hl};
  const double cz[4] = {hw, -hw, -hw, hw};
  Geom g;
  double y0 = (double)b[1], y1 = nh + (double)b[1];
  if (appr > 0.0) {
    y0 = (double)(int)(y0 * appr);
    y1 = (double)(int)(y1 * appr);
  }
  g.lo[1] = fmin(y0, y1);
  g.hi[1] = fmax(y0, y1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double x = (cx[i] * c + cz[i] * s) + (double)b[0];
    double z = (cx[i] * -s + cz[i] * c) + (double)b[2];
    if (appr > 0.0) {
      x = (double)(int)(x * appr);
      z = (double)(int)(z * appr);
    }
    g.px[i] = x;
    g.pz[i] = z;
  }
  g.lo[0] = fmin(fmin(g.px[0], g.px[1]), fmin(g.px[2], g.px[3]));
  g.hi[0] = fmax(fmax(g.px[0], g.px[1]), fmax(g.px[2], g.px[3]));
  g.lo[2] = fmin(fmin(g.pz[0], g.pz[1]), fmin(g.pz[2], g.pz[3]));
  g.hi[2] = fmax(fmax(g.pz[0], g.pz[1]), fmax(g.pz[2], g.pz[3]));
  double a2 = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    a2 += g.px[i] * g.pz[j] - g.px[j] * g.pz[i];
  }
  g.area = fabs(a2) * 0.5;
  return g;
}

// Area of (quadrilateral a) ∩ (convex quadrilateral b): Sutherland-Hodgman
// clipping of a against the four half-planes of b, shoelace on the result.
// Stands in for shapely's `p1.intersection(p2).area` (nms.py:82).
// Every array index below is a compile-time constant after unrolling (vertex
// pushes are select chains), so the polygon lives in VGPRs: the first version
// indexed dynamically, went through scratch memory and spent ~13 us per call
// in the latency-bound scan kernel.
constexpr int kMaxVerts = 8;  // a quad gains at most one vertex per clip

__device__ __forceinline__ void push_vertex(double (&tx)[kMaxVerts],
                                            double (&tz)[kMaxVerts], int &m,
                                            double x, double z) {
#pragma unroll
  for (int k = 0; k < kMaxVerts; ++k) {
    tx[k] = (m == k) ? x : tx[k];
    tz[k] = (m == k) ? z : tz[k];
  }
  ++m;
}

__device__ double clip_area(const Geom &a, const Geom &b) {
  double sx[kMaxVerts], sz[kMaxVerts];
#pragma unroll
  for (int i = 0; i < kMaxVerts; ++i) {
    sx[i] = a.px[i & 3];
    sz[i] = a.pz[i & 3];
  }
  int n = 4;
  double orient = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    orient += b.px[i] * b.pz[j] - b.px[j] * b.pz[i];
  }
  if (orient == 0.0) return 0.0;
  const double sgn = orient > 0.0 ? 1.0 : -1.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double ex = b.px[e], ez = b.pz[e];
    const double dx = b.px[(e + 1) & 3] - ex, dz = b.pz[(e + 1) & 3] - ez;
    double d[kMaxVerts];
#pragma unroll
    for (int i = 0; i < kMaxVerts; ++i)
      d[i] = sgn * (dx * (sz[i] - ez) - dz * (sx[i] - ex));
    double tx[kMaxVerts], tz[kMaxVerts];
#pragma unroll
    for (int i = 0; i < kMaxVerts; ++i) tx[i] = tz[i] = 0.0;
    int m = 0;
#pragma unroll
    for (int i = 0; i < kMaxVerts; ++i) {
      if (i < n) {
        const bool last = i + 1 == n;
        const int jn = (i + 1) & (kMaxVerts - 1);
        const double jx = last ? sx[0] : sx[jn];
        const double jz = last ? sz[0] : sz[jn];
        const double dj = last ? d[0] : d[jn];
        const double di = d[i];
        if (di >= 0.0) push_vertex(tx, tz, m, sx[i], sz[i]);
        if ((di >= 0.0) != (dj >= 0.0)) {
          const double t = di / (di - dj);
          push_vertex(tx, tz, m, sx[i] + t * (jx - sx[i]),
                      sz[i] + t * (jz - sz[i]));
        }
      }
    }
    n = m < kMaxVerts ? m : kMaxVerts;
#pragma unroll
    for (int i = 0; i < kMaxVerts; ++i) {
      sx[i] = tx[i];
      sz[i] = tz[i];
    }
  }
  if (n < 3) return 0.0;
  double a2 = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxVerts; ++i) {
    if (i < n) {
      const bool last = i + 1 == n;
      const int jn = (i + 1) & (kMaxVerts - 1);
      const double jx = last ? sx[0] : sx[jn];
      const double jz = last ? sz[0] : sz[jn];
      a2 += sx[i] * jz - jx * sz[i];
    }
  }
  return fabs(a2) * 0.5;
}

// overlapped_boxes_3d_fast_poly for one pair (nms.py:64-88): `single` is the
// reference's single_box, `other` one entry of box_list.
__device__ double overlap_3d(const Geom &single, const Geom &other) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (single.hi[k] < other.lo[k] || single.lo[k] > other.hi[k]) return 0.0;
  const double shared_area = clip_area(single, other);
  const double shared_y =
      fmin(other.hi[1], single.hi[1]) - fmax(other.lo[1], single.lo[1]);
  const double inter = shared_y * shared_area;
  const double uni = (other.hi[1] - other.lo[1]) * other.area +
                     (single.hi[1] - single.lo[1]) * single.area;
  return (double)(float)inter / (uni - inter);
}

// ---------------------------------------------------------------- box codec
// class table rows: {l, h, w, yaw_offset, active}; label values index it.
__global__ void box_decode_kernel(const int32_t *__restrict__ labels,
                                  const float *__restrict__ xyz,
                                  const float *__restrict__ enc,
                                  const float *__restrict__ table, int n_table,
                                  int64_t rows, int per_row,
                                  float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row;
  const int col = (int)(idx - r * per_row);
  const float *e = enc + idx * 7;
  float d[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = e[i];
  const int lab = labels[r];
  if (col == 0 && lab >= 0 && lab < n_table && table[5 * lab + 4] != 0.0f) {
    const float l = table[5 * lab], h = table[5 * lab + 1],
                w = table[5 * lab + 2], yo = table[5 * lab + 3];
    d[0] = e[0] * l;
    d[1] = e[1] * h;
    d[2] = e[2] * w;
    d[3] = (float)exp((double)e[3]) * l;
    d[4] = (float)exp((double)e[4]) * h;
    d[5] = (float)exp((double)e[5]) * w;
    d[6] = e[6] * 0.78539816339744830962f;
    if (yo != 0.0f) d[6] = d[6] + yo;
  }
  d[0] = d[0] + xyz[3 * r];
  d[1] = d[1] + xyz[3 * r + 1];
  d[2] = d[2] + xyz[3 * r + 2];
#pragma unroll
  for (int i = 0; i < 7; ++i) out[idx * 7 + i] = d[i];
}

__global__ void box_encode_kernel(const int32_t *__restrict__ labels,
                                  const float *__restrict__ xyz,
                                  const float *__restrict__ boxes,
                                  const float *__restrict__ table, int n_table,
                                  int64_t rows, int per_row,
                                  float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row;
  const int col = (int)(idx - r * per_row);
  const float *b = boxes + idx * 7;
  float d[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = b[i];
  d[0] = b[0] - xyz[3 * r];
  d[1] = b[1] - xyz[3 * r + 1];
  d[2] = b[2] - xyz[3 * r + 2];
  const int lab = labels[r];
  if (col == 0 && lab >= 0 && lab < n_table && table[5 * lab + 4] != 0.0f) {
    const float l = table[5 * lab], h = table[5 * lab + 1],
                w = table[5 * lab + 2], yo = table[5 * lab + 3];
    d[0] = d[0] / l;
    d[1] = d[1] / h;
    d[2] = d[2] / w;
    d[3] = (float)log((double)(b[3] / l));
    d[4] = (float)log((double)(b[4] / h));
    d[5] = (float)log((double)(b[5] / w));
    const float y = yo != 0.0f ? b[6] - yo : b[6];
    d[6] = y / 0.78539816339744830962f;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) out[idx * 7 + i] = d[i];
}

// ---------------------------------------------------------------- candidates
// run.py:266-290: flat index p = k*nc + c is a candidate when 0 < c < nc-1 and
// prob > 1/nc; indices come out ascending like np.nonzero.  One workgroup,
// ordered compaction with a running base.
__global__ __launch_bounds__(kScanThreads) void candidates_kernel(
    const float *__restrict__ probs, int64_t n, int nc, int32_t *out_index,
    int32_t *out_label, int32_t *out_count, int64_t capacity,
    const int32_t *__restrict__ n_vertices_dev) {
  __shared__ int wave_tot[kScanThreads / 64];
  __shared__ int base_s;
  if (n_vertices_dev) {  // capacity form: the rows that exist
    const int64_t live = (int64_t)max(*n_vertices_dev, 0) * nc;
    n = live < n ? live : n;
  }
  const float thr = (float)(1.0 / (double)nc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int64_t p0 = 0; p0 < n; p0 += kScanThreads) {
    const int64_t p = p0 + threadIdx.x;
    bool ok = false;
    int c = 0;
    if (p < n) {
      c = (int)(p % nc);
      ok = c > 0 && c < nc - 1 && probs[p] > thr;
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int before = base_s;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && slot < capacity) {
      out_index[slot] = (int32_t)p;
      out_label[slot] = (c & 1) ? c : c - 1;  // run.py:287-289
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kScanThreads / 64; ++w) t += wave_tot[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_count = base_s;
}

// ---------------------------------------------------------------- NMS
__global__ void sort_keys_kernel(const float *__restrict__ scores, int64_t n,
                                 uint32_t *keys, uint32_t *vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t u = __float_as_uint(scores[i]);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending float order
  keys[i] = ~u;                                     // descending score
  vals[i] = (uint32_t)i;
}

__global__ void gather_sorted_kernel(const uint32_t *__restrict__ order,
                                     const int32_t *__restrict__ labels,
                                     const float *__restrict__ boxes,
                                     const float *__restrict__ scores,
                                     const int32_t *__restrict__ attrs,
                                     int64_t m, double appr, int32_t *s_label,
                                     float *s_box, float *s_score,
                                     int32_t *s_attr, Geom *geom,
                                     int32_t *row_flag, int32_t *active,
                                     int32_t *spill_cursor) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t o = order[i];
  s_label[i] = labels[o];
  s_score[i] = scores[o];
  s_attr[i] = attrs ? attrs[o] : (int32_t)o;
  float b[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    b[k] = boxes[(int64_t)o * 7 + k];
    s_box[i * 7 + k] = b[k];
  }
  geom[i] = box_geometry(b, appr);
  row_flag[i] = 0;
  active[i] = 0;
  if (i == 0) *spill_cursor = 0;
}

// bit j of mask[i*words + j/64] <=> j > i, same class, overlap(i, j) > thr.
// One 256-thread workgroup per 64 x 64 block of pairs: the 64 row and 64
// column bounds/labels are staged in LDS once (the per-row form re-read 3 KB
// per 64 pairs from L2), wave w covers rows 16w..16w+15, lane = column, and
// only pairs that pass the label and bounds test fetch the full geometry.
__global__ __launch_bounds__(256) void overlap_mask_kernel(
    const Geom *__restrict__ geom, const int32_t *__restrict__ label,
    int64_t m, int words, double thr, unsigned long long *__restrict__ mask,
    int32_t *__restrict__ row_flag) {
  __shared__ double rlo[64][3], rhi[64][3], clo[64][3], chi[64][3];
  __shared__ int rlab[64], clab[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // grid-stride over the words*words blocks: a launch may not exceed 2^32
  // threads (larger grids wrap silently)
  const int64_t total = (int64_t)words * words;
  for (int64_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
    const int bi = (int)(blk / words), bj = (int)(blk - (int64_t)bi * words);
    if (bj < bi) {  // below the diagonal: the rows still need their zeros
      if (threadIdx.x < 64 && (int64_t)bi * 64 + threadIdx.x < m)
        mask[((int64_t)bi * 64 + threadIdx.x) * words + bj] = 0ull;
      continue;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const bool col = threadIdx.x >= 64;
      const int t = threadIdx.x & 63;
      const int64_t g = (int64_t)(col ? bj : bi) * 64 + t;
      double(*lo)[3] = col ? clo : rlo;
      double(*hi)[3] = col ? chi : rhi;
      int *lab = col ? clab : rlab;
      if (g < m) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          lo[t][k] = geom[g].lo[k];
          hi[t][k] = geom[g].hi[k];
        }
        lab[t] = label[g];
      } else {
        lab[t] = -0x7fffffff - (col ? 1 : 0);  // never equal to anything
      }
    }
    __syncthreads();
    const int64_t j = (int64_t)bj * 64 + lane;
    for (int rr = 0; rr < 16; ++rr) {
      const int r = 16 * wave + rr;
      const int64_t i = (int64_t)bi * 64 + r;
      if (i >= m) break;  // wave-uniform
      bool hit = false;
      if (j > i && j < m && clab[lane] == rlab[r]) {
        bool apart = false;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          apart |= rhi[r][k] < clo[lane][k] || rlo[r][k] > chi[lane][k];
        if (!apart) {
          const Geom a = geom[i];
          const Geom b = geom[j];
          hit = overlap_3d(a, b) > thr;
        }
      }
      const unsigned long long bal = __ballot(hit);
      if (lane == 0) {
        mask[i * words + bj] = bal;
        if (bal) row_flag[i] = 1;
      }
    }
  }
}

struct ScanArgs {
  int64_t m;
  int words;
  unsigned long long *mask;  // rows of survivors are rewritten masked
  const int32_t *row_flag;
  int32_t *active;       // [m] survivor with a non-empty removed set
  int32_t *spill_cursor;  // bump allocator of the spill buffers
  const Geom *geom;
  const int32_t *s_label;
  float *s_box;     // [m,7], merged boxes written in place
  float *s_score;   // [m], accumulated scores written in place
  const int32_t *s_attr;
  int merge, rescore;
  int32_t *big_list;  // 2m ints  (removed sets larger than kListCap)
  float *big_vals;    // 14m floats
  double *big_dbl;    // 2m doubles
  unsigned long long *keep_ws;  // words: the final keep bits
  int32_t *out_label;
  float *out_box;
  float *out_score;
  int32_t *out_attr;
  int32_t *out_count;
};

constexpr int kKeepLds = 1024;  // 65536 boxes with the keep bits in LDS

__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_tot,
                                                    int *total) {
  // wave-inclusive scan by shuffles, then wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_tot[wave] = x;
  __syncthreads();
  int before = 0, tot = 0;
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wave) before += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return before + x - v;
}

// ---- phase 1 of the scan: which boxes survive, and who removed whom ---------
// The reference loop (nms.py:142-166) is sequential only through the keep
// flags: box i's removed set is row_i & keep, and the boxes it removes are
// never looked at again.  One WAVE replays exactly that on the bit matrix --
// keep words in LDS, no workgroup barriers, the mask row of the next flagged
// candidate requested while the current one is processed -- and writes each
// surviving row back MASKED (its removed set).  The per-row arithmetic (median,
// overlaps, score sum) depends only on that set and on untouched inputs, so it
// runs afterwards for all rows in parallel (nms_merge_kernel).
__global__ __launch_bounds__(64) void nms_resolve_kernel(ScanArgs a) {
  __shared__ unsigned long long keep_lds[kKeepLds];
  const int lane = threadIdx.x;
  unsigned long long *keep = a.words <= kKeepLds ? keep_lds : a.keep_ws;
  for (int w = lane; w < a.words; w += 64) {
    const int64_t left = a.m - (int64_t)w * 64;
    keep[w] = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
  }
  int64_t pre_row = -1;  // row whose first 64 mask words sit in pre_bits
  unsigned long long pre_bits = 0ull;
  int next_flag = lane < a.m ? a.row_flag[lane] : 0;
  for (int64_t i0 = 0; i0 < a.m; i0 += 64) {
    const unsigned long long fb = __ballot(next_flag != 0);
    if (i0 + 64 < a.m)  // the next chunk's flags travel during this chunk
      next_flag = i0 + 64 + lane < a.m ? a.row_flag[i0 + 64 + lane] : 0;
    const int64_t kw = i0 >> 6;
    unsigned long long rem = fb & keep[kw];
    while (rem) {
      const int bit = __builtin_ctzll(rem);
      rem &= rem - 1;
      const int64_t i = i0 + bit;
      bool any = false;
      for (int w = lane; w < a.words; w += 64) {
        const unsigned long long row_bits =
            (w == lane && pre_row == i) ? pre_bits : a.mask[i * a.words + w];
        if (w == lane && rem) {  // speculative: it may get removed meanwhile
          pre_row = i0 + __builtin_ctzll(rem);
          pre_bits = a.mask[pre_row * a.words + w];
        }
        const unsigned long long r = row_bits & keep[w];
        if (r) keep[w] &= ~r;
        a.mask[i * a.words + w] = r;
        any = any || r != 0ull;
      }
      const unsigned long long act = __ballot(any);
      if (lane == 0) a.active[i] = act != 0ull ? 1 : 0;
      // the keep words other lanes just updated (LDS, or global for > 65 536
      // boxes) must be visible before the next candidate is chosen
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      rem &= keep[kw];
    }
  }
  if (keep != a.keep_ws)
    for (int w = lane; w < a.words; w += 64) a.keep_ws[w] = keep[w];
}

// ---- phase 2: one workgroup per surviving row with a non-empty removed set ---
__global__ __launch_bounds__(kScanThreads) void nms_merge_kernel(ScanArgs a) {
  const int64_t i = blockIdx.x;
  if (!a.active[i]) return;
  __shared__ int list_lds[kListCap];
  __shared__ float vals_lds[7 * (kListCap + 1)];
  __shared__ double dbl_lds[kListCap];
  __shared__ int wave_tot[kScanThreads / 64];
  __shared__ float med[7][2];
  __shared__ int seg_base;
  const int tid = threadIdx.x;
  // size of the removed set, then its members in ascending order
  int n = 0;
  for (int w0 = 0; w0 < a.words; w0 += kScanThreads) {
    const int w = w0 + tid;
    int tot;
    block_exclusive_scan(w < a.words ? __popcll(a.mask[i * a.words + w]) : 0,
                         wave_tot, &tot);
    n += tot;
  }
  const bool small = n <= kListCap;
  if (!small) {  // disjoint sets: a private segment of the spill buffers
    if (tid == 0) seg_base = atomicAdd(a.spill_cursor, n + 1);
    __syncthreads();
  }
  int *list = small ? list_lds : a.big_list + seg_base;
  float *vals = small ? vals_lds : a.big_vals + (int64_t)7 * seg_base;
  double *dbl = small ? dbl_lds : a.big_dbl + seg_base;
  int base = 0;
  for (int w0 = 0; w0 < a.words; w0 += kScanThreads) {
    const int w = w0 + tid;
    unsigned long long bits = w < a.words ? a.mask[i * a.words + w] : 0ull;
    int tot;
    int pos = base + block_exclusive_scan(__popcll(bits), wave_tot, &tot);
    while (bits) {
      const int bb = __builtin_ctzll(bits);
      bits &= bits - 1;
      list[pos++] = w * 64 + bb;
    }
    base += tot;
  }
  __syncthreads();
  const int n1 = n + 1;
  float newbox[7];
  if (a.merge) {
    // np.median over [removed..., box i] per component (nms.py:155-157)
    for (int t = tid; t < 7 * n1; t += kScanThreads) {
      const int c = t / n1, u = t - c * n1;
      const int64_t row = u < n ? list[u] : i;
      vals[t] = a.s_box[row * 7 + c];
    }
    __syncthreads();
    const int r_lo = (n1 - 1) >> 1, r_hi = n1 >> 1;
    for (int t = tid; t < 7 * n1; t += kScanThreads) {
      const int c = t / n1, u = t - c * n1;
      const float *v = vals + c * n1;
      const float x = v[u];
      int rank = 0;
      for (int q = 0; q < n1; ++q) {
        const float y = v[q];
        rank += (y < x || (y == x && q < u)) ? 1 : 0;
      }
      if (rank == r_lo) med[c][0] = x;
      if (rank == r_hi) med[c][1] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 7; ++c)
      newbox[c] = (n1 & 1) ? med[c][0] : (med[c][0] + med[c][1]) / 2.0f;
  } else {
#pragma unroll
    for (int c = 0; c < 7; ++c) newbox[c] = a.s_box[i * 7 + c];
  }
  if (a.rescore) {
    // scores[i] += sum(scores[removed] * overlap(mean box, removed))
    const Geom g = box_geometry(newbox, 0.0);
    for (int u = tid; u < n; u += kScanThreads) {
      const int j = list[u];
      const Geom o = a.geom[j];
      dbl[u] = (double)a.s_score[j] * overlap_3d(g, o);
    }
    __syncthreads();
    // fixed-shape tree: deterministic regardless of timing
    int len = n;
    while (len > 1) {
      const int half = (len + 1) >> 1;
      for (int u = tid; u + half < len; u += kScanThreads)
        dbl[u] += dbl[u + half];
      __syncthreads();
      len = half;
    }
    if (tid == 0) a.s_score[i] = (float)((double)a.s_score[i] + dbl[0]);
  }
  // row i is a survivor: no other workgroup reads or writes its box / score
  if (a.merge && tid < 7) a.s_box[i * 7 + tid] = newbox[tid];
}

// ---- phase 3: kept boxes, in score order --------------------------------------
__global__ __launch_bounds__(kScanThreads) void nms_output_kernel(ScanArgs a) {
  __shared__ int wave_tot[kScanThreads / 64];
  const int tid = threadIdx.x;
  int base = 0;
  for (int w0 = 0; w0 < a.words; w0 += kScanThreads) {
    const int w = w0 + tid;
    unsigned long long bits = w < a.words ? a.keep_ws[w] : 0ull;
    int tot;
    int pos = base + block_exclusive_scan(__popcll(bits), wave_tot, &tot);
    while (bits) {
      const int b = __builtin_ctzll(bits);
      bits &= bits - 1;
      const int64_t j = (int64_t)w * 64 + b;
      a.out_label[pos] = a.s_label[j];
      a.out_score[pos] = a.s_score[j];
      a.out_attr[pos] = a.s_attr[j];
#pragma unroll
      for (int c = 0; c < 7; ++c) a.out_box[(int64_t)pos * 7 + c] = a.s_box[j * 7 + c];
      ++pos;
    }
    base += tot;
  }
  if (tid == 0) *a.out_count = base;
}

__global__ void pairwise_overlap_kernel(const float *__restrict__ single,
                                        const float *__restrict__ boxes,
                                        int64_t n, double appr,
                                        double *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const Geom a = box_geometry(single, appr);
  const Geom b = box_geometry(boxes + j * 7, appr);
  out[j] = overlap_3d(a, b);
}

struct NmsLayout {
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b;
  void *sort_scratch;
  size_t sort_bytes;
  int32_t *s_label, *s_attr, *row_flag, *active, *big_list;
  float *s_box, *s_score, *big_vals;
  double *big_dbl;
  Geom *geom;
  unsigned long long *mask, *keep_ws;
  int words;
};

bool carve_nms(Arena &ar, int64_t n, NmsLayout *L) {
  const size_t m = (size_t)(n > 0 ? n : 1);
  L->words = (int)((m + 63) / 64);
  L->sort_bytes = radix_sort_scratch_bytes((int64_t)m);
  L->keys_a = ar.take<uint32_t>(m);
  L->vals_a = ar.take<uint32_t>(m);
  L->keys_b = ar.take<uint32_t>(m);
  L->vals_b = ar.take<uint32_t>(m);
  L->sort_scratch = ar.take<char>(L->sort_bytes);
  L->s_label = ar.take<int32_t>(m);
  L->s_attr = ar.take<int32_t>(m);
  L->row_flag = ar.take<int32_t>(m + 1);  // + the spill cursor
  L->active = ar.take<int32_t>(m);
  L->big_list = ar.take<int32_t>(2 * m);
  L->s_box = ar.take<float>(7 * m);
  L->s_score = ar.take<float>(m);
  L->big_vals = ar.take<float>(14 * m);
  L->big_dbl = ar.take<double>(2 * m);
  L->geom = ar.take<Geom>(m);
  L->keep_ws = ar.take<unsigned long long>((size_t)L->words);
  L->mask = ar.take<unsigned long long>(m * (size_t)L->words);
  return L->mask != nullptr;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" int pgnn_box_decode_f32(const int32_t *cls_labels, const float *xyz,
                                   const float *encoded,
                                   const float *class_table, int32_t n_table,
                                   int64_t n_rows, int32_t boxes_per_row,
                                   float *decoded, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && boxes_per_row > 0 && n_table >= 0,
               PGNN_E_INVALID, "box_decode: bad size");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(cls_labels && xyz && encoded && decoded &&
                   (class_table || n_table == 0),
               PGNN_E_INVALID, "box_decode: null pointer");
  const int64_t total = n_rows * boxes_per_row;
  hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, stream, cls_labels, xyz, encoded,
                     class_table, n_table, n_rows, boxes_per_row, decoded);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_box_encode_f32(const int32_t *cls_labels, const float *xyz,
                                   const float *boxes, const float *class_table,
                                   int32_t n_table, int64_t n_rows,
                                   int32_t boxes_per_row, float *encoded,
                                   void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && boxes_per_row > 0 && n_table >= 0,
               PGNN_E_INVALID, "box_encode: bad size");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(cls_labels && xyz && boxes && encoded &&
                   (class_table || n_table == 0),
               PGNN_E_INVALID, "box_encode: null pointer");
  const int64_t total = n_rows * boxes_per_row;
  hipLaunchKernelGGL(box_encode_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, stream, cls_labels, xyz, boxes, class_table,
                     n_table, n_rows, boxes_per_row, encoded);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_detection_candidates(const float *probs,
                                         int64_t n_vertices,
                                         int32_t num_classes,
                                         int32_t *out_index, int32_t *out_label,
                                         int64_t capacity, int32_t *out_count,
                                         void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices >= 0 && num_classes > 0 && capacity >= 0 && out_count,
               PGNN_E_INVALID, "detection_candidates: bad argument");
  PGNN_REQUIRE(n_vertices * num_classes < (int64_t)1 << 31, PGNN_E_INVALID,
               "detection_candidates: index range exceeds int32");
  if (n_vertices == 0) {
    PGNN_HIP(hipMemsetAsync(out_count, 0, 4, stream));
    return 0;
  }
  PGNN_REQUIRE(probs && (capacity == 0 || (out_index && out_label)),
               PGNN_E_INVALID, "detection_candidates: null pointer");
  hipLaunchKernelGGL(candidates_kernel, dim3(1), dim3(kScanThreads), 0, stream,
                     probs, n_vertices * num_classes, num_classes, out_index,
                     out_label, out_count, capacity, (const int32_t *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_detection_candidates_dyn(const float *probs,
                                             int64_t n_vertices_cap,
                                             const int32_t *n_vertices_dev,
                                             int32_t num_classes,
                                             int32_t *out_index,
                                             int32_t *out_label,
                                             int64_t capacity,
                                             int32_t *out_count,
                                             void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices_cap >= 0 && num_classes > 0 && capacity >= 0 &&
                   out_count && n_vertices_dev,
               PGNN_E_INVALID, "detection_candidates_dyn: bad argument");
  PGNN_REQUIRE(n_vertices_cap * num_classes < (int64_t)1 << 31, PGNN_E_INVALID,
               "detection_candidates_dyn: index range exceeds int32");
  if (n_vertices_cap == 0) {
    PGNN_HIP(hipMemsetAsync(out_count, 0, 4, stream));
    return 0;
  }
  PGNN_REQUIRE(probs && (capacity == 0 || (out_index && out_label)),
               PGNN_E_INVALID, "detection_candidates_dyn: null pointer");
  hipLaunchKernelGGL(candidates_kernel, dim3(1), dim3(kScanThreads), 0, stream,
                     probs, n_vertices_cap * num_classes, num_classes,
                     out_index, out_label, out_count, capacity, n_vertices_dev);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" size_t pgnn_nms_workspace_bytes(int64_t n_boxes) {
  if (n_boxes < 0) return 0;
  Arena ar(nullptr, 0);
  NmsLayout L;
  carve_nms(ar, n_boxes, &L);
  return align_up(ar.used, 256);
}

extern "C" int pgnn_nms_boxes_3d(const int32_t *class_labels,
                                 const float *boxes_3d, const float *scores,
                                 const int32_t *attributes, int64_t n_boxes,
                                 double overlapped_thres, int32_t mode,
                                 float appr_factor, int64_t top_k,
                                 void *workspace, size_t workspace_bytes,
                                 int32_t *out_labels, float *out_boxes,
                                 float *out_scores, int32_t *out_attributes,
                                 int32_t *out_count, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_boxes >= 0 && n_boxes < ((int64_t)1 << 24) && out_count &&
                   mode >= 0 && mode <= 3,
               PGNN_E_INVALID, "nms_boxes_3d: bad argument");
  if (n_boxes == 0) {
    PGNN_HIP(hipMemsetAsync(out_count, 0, 4, stream));
    return 0;
  }
  PGNN_REQUIRE(class_labels && boxes_3d && scores && out_labels && out_boxes &&
                   out_scores && out_attributes,
               PGNN_E_INVALID, "nms_boxes_3d: null pointer");
  Arena ar(workspace, workspace_bytes);
  NmsLayout L;
  PGNN_REQUIRE(carve_nms(ar, n_boxes, &L), PGNN_E_WORKSPACE,
               "nms_boxes_3d: workspace too small "
               "(see pgnn_nms_workspace_bytes)");
  const unsigned nb = (unsigned)((n_boxes + 255) / 256);
  hipLaunchKernelGGL(sort_keys_kernel, dim3(nb), dim3(256), 0, stream, scores,
                     n_boxes, L.keys_a, L.vals_a);
  PGNN_HIP(hipGetLastError());
  uint32_t *keys = nullptr, *order = nullptr;
  int rc = radix_sort_pairs(L.keys_a, L.vals_a, L.keys_b, L.vals_b, n_boxes, 32,
                            L.sort_scratch, L.sort_bytes, &keys, &order,
                            stream);
  if (rc != 0) return rc;
  const int64_t m = (top_k > 0 && top_k < n_boxes) ? top_k : n_boxes;
  const int words = (int)((m + 63) / 64);
  // mode 0 = nms_boxes_3d: integer "pixel" corners (nms.py:113-115)
  const double appr = mode == 0 ? (double)appr_factor : 0.0;
  hipLaunchKernelGGL(gather_sorted_kernel, dim3((unsigned)((m + 255) / 256)),
                     dim3(256), 0, stream, order, class_labels, boxes_3d,
                     scores, attributes, m, appr, L.s_label, L.s_box,
                     L.s_score, L.s_attr, L.geom, L.row_flag, L.active,
                     L.row_flag + n_boxes);
  PGNN_HIP(hipGetLastError());
  const int64_t pair_blocks = (int64_t)words * words;
  hipLaunchKernelGGL(overlap_mask_kernel,
                     dim3((unsigned)(pair_blocks < (1 << 22) ? pair_blocks
                                                             : (1 << 22))),
                     dim3(256), 0, stream, L.geom, L.s_label, m, words,
                     overlapped_thres, L.mask, L.row_flag);
  PGNN_HIP(hipGetLastError());
  ScanArgs sa;
  sa.m = m;
  sa.words = words;
  sa.mask = L.mask;
  sa.row_flag = L.row_flag;
  sa.active = L.active;
  sa.spill_cursor = L.row_flag + n_boxes;
  sa.geom = L.geom;
  sa.s_label = L.s_label;
  sa.s_box = L.s_box;
  sa.s_score = L.s_score;
  sa.s_attr = L.s_attr;
  sa.merge = (mode == 1 || mode == 2) ? 1 : 0;
  sa.rescore = (mode == 1 || mode == 3) ? 1 : 0;
  sa.big_list = L.big_list;
  sa.big_vals = L.big_vals;
  sa.big_dbl = L.big_dbl;
  sa.keep_ws = L.keep_ws;
  sa.out_label = out_labels;
  sa.out_box = out_boxes;
  sa.out_score = out_scores;
  sa.out_attr = out_attributes;
  sa.out_count = out_count;
  hipLaunchKernelGGL(nms_resolve_kernel, dim3(1), dim3(64), 0, stream, sa);
  if (sa.merge || sa.rescore)
    hipLaunchKernelGGL(nms_merge_kernel, dim3((unsigned)m), dim3(kScanThreads),
                       0, stream, sa);
  hipLaunchKernelGGL(nms_output_kernel, dim3(1), dim3(kScanThreads), 0, stream,
                     sa);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_overlapped_boxes_3d(const float *single_box,
                                        const float *boxes_3d, int64_t n_boxes,
                                        float appr_factor, double *overlap,
                                        void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_boxes >= 0, PGNN_E_INVALID, "overlapped_boxes_3d: bad size");
  if (n_boxes == 0) return 0;
  PGNN_REQUIRE(single_box && boxes_3d && overlap, PGNN_E_INVALID,
               "overlapped_boxes_3d: null pointer");
  hipLaunchKernelGGL(pairwise_overlap_kernel,
                     dim3((unsigned)((n_boxes + 255) / 256)), dim3(256), 0,
                     stream, single_box, boxes_3d, n_boxes,
                     (double)appr_factor, overlap);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
