// Training-side kernels: materialising forward
// pieces, their adjoints, weight gradients, loss and SGD.
//
// The inference path never writes an E x C matrix to HBM.  The backward pass
// needs the per-edge activations, so the training forward materialises them
// (E is ~5x smaller under the training graph kwargs: voxel 0.8 m, fan-in cap
// 256) and every adjoint below is a plain, separately testable primitive:
//   H1 = ReLU(P[src] - Q[dst])                  edge_hidden_fwd / _bwd
//   scatter-max gradient with TF's tie rule      scatter_max_bwd (count + route)
//   dW = X^T dZ, db = sum dZ                     weight_grad (MFMA, split over rows)
//   dX = dZ W^T                                  pgnn_mlp_fwd on a transposed pack
//   softmax-CE + Huber                           loss_fwd_bwd   (models.py:170-311)
//   w -= lr (g + l1 sign(w))                     sgd_step       (train.py:375-405)
#include "mlp_engine.h"
#include "sort.h"

namespace {
using namespace pgnn;

// ---------------------------------------------------------------- packing on device
__global__ void pack_fc_device_kernel(const float *__restrict__ w,
                                      const float *__restrict__ b, int k_in,
                                      int n_out, int transpose,
                                      float *__restrict__ packed) {
  // logical layer: y = x W' + b with W' = W ([k_in,n_out]) or W^T (then the
  // layer maps n_out -> k_in and has no bias)
  const int K = transpose ? n_out : k_in, N = transpose ? k_in : n_out;
  const int kq = (K + 15) / 16, nt = (N + 15) / 16;
  const int64_t total = (int64_t)kq * nt * 256;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total + nt * 16; idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx >= total) {
      const int n = (int)(idx - total);
      packed[idx] = (!transpose && b && n < N) ? b[n] : 0.0f;
      continue;
    }
    const int s = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const int64_t qt = idx >> 8;
    const int t = (int)(qt % nt), q = (int)(qt / nt);
    const int k = 16 * q + 4 * (lane >> 4) + s, n = 16 * t + (lane & 15);
    float v = 0.0f;
    if (k < K && n < N)
      v = transpose ? w[(int64_t)n * n_out + k] : w[(int64_t)k * n_out + n];
    packed[idx] = v;
  }
}

// ---------------------------------------------------------------- edge hidden layer
__global__ void edge_hidden_fwd_kernel(const float *__restrict__ P,
                                       const float *__restrict__ Q, int ld4,
                                       const int32_t *__restrict__ edges,
                                       int64_t n_edges, float *__restrict__ H1) {
  const v4f *P4 = reinterpret_cast<const v4f *>(P);
  const v4f *Q4 = reinterpret_cast<const v4f *>(Q);
  v4f *H4 = reinterpret_cast<v4f *>(H1);
  const int64_t total = n_edges * ld4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / ld4;
    const int c = (int)(idx - e * ld4);
    const int s = edges[2 * e], d = edges[2 * e + 1];
    const v4f p = P4[(int64_t)s * ld4 + c], q = Q4[(int64_t)d * ld4 + c];
    v4f h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = p[i] - q[i];
      h[i] = t > 0.0f ? t : 0.0f;
    }
    H4[idx] = h;
  }
}

// dP[src] += g, dQ[dst] -= g with g = dH1 (already masked by H1 > 0)
__global__ void edge_hidden_bwd_kernel(const float *__restrict__ dH1, int ld,
                                       const int32_t *__restrict__ edges,
                                       int64_t n_edges, float *__restrict__ dP,
                                       float *__restrict__ dQ) {
  const int64_t total = n_edges * ld;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / ld;
    const int c = (int)(idx - e * ld);
    const float g = dH1[idx];
    if (g != 0.0f) {
      atomicAdd(&dP[(int64_t)edges[2 * e] * ld + c], g);
      atomicAdd(&dQ[(int64_t)edges[2 * e + 1] * ld + c], -g);
    }
  }
}

// PointSetPooling's edge rows for any feature width (16 columns for the raw
// point features; a pooling level above the first pools the previous level's
// 300-wide features): F[e] = [f(src)[:nfeat] | xyz(src) - xyz(kp(dst)) | 0 ...],
// ld_f columns; thread per (edge, column)
__global__ void pool_features_wide_kernel(
    const float *__restrict__ feat, int64_t ld_feat, int nfeat,
    const float *__restrict__ xyz, const int32_t *__restrict__ kp,
    const int32_t *__restrict__ edges, int64_t n_edges, float *__restrict__ F,
    int ld_f) {
  const int64_t total = n_edges * ld_f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / ld_f;
    const int i = (int)(idx - e * ld_f);
    const int s = edges[2 * e];
    float v = 0.0f;
    if (i < nfeat) {
      v = feat[(int64_t)s * ld_feat + i];
    } else if (i < nfeat + 3) {
      const int k = kp[edges[2 * e + 1]];
      v = xyz[3 * (int64_t)s + (i - nfeat)] - xyz[3 * (int64_t)k + (i - nfeat)];
    }
    F[idx] = v;
  }
}

__global__ void relu_mask_mul_kernel(float *__restrict__ dY,
                                     const float *__restrict__ Y, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x)
    if (!(Y[idx] > 0.0f)) dY[idx] = 0.0f;  // tf ReluGrad: pass where y > 0
}

// ---------------------------------------------------------------- scatter-max gradient
// TF _UnsortedSegmentMinOrMaxGrad: selected = (data == out[seg]); the gradient
// of a segment/channel is divided equally among its selected rows.
__global__ void segmax_count_kernel(const float *__restrict__ data, int64_t ld,
                                    const int32_t *__restrict__ seg, int64_t rows,
                                    int cols, int nseg,
                                    const float *__restrict__ out, int64_t ldo,
                                    int32_t *__restrict__ count,
                                    int relu_mask) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    const int s = seg[r];
    if (s < 0 || s >= nseg) continue;
    // With relu_mask the route kernel gives nothing to entries <= 0, and a
    // maximum <= 0 selects only such entries, so their tie counts are never
    // read: skipping them removes the hot atomics of all-zero (post-ReLU)
    // segments, dozens of adds to one address each.
    const float v = data[r * ld + c];
    if (v == out[(int64_t)s * ldo + c] && (!relu_mask || v > 0.0f))
      atomicAdd(&count[(int64_t)s * cols + c], 1);
  }
}

__global__ void segmax_route_kernel(const float *__restrict__ data, int64_t ld,
                                    const int32_t *__restrict__ seg, int64_t rows,
                                    int cols, int nseg,
                                    const float *__restrict__ out, int64_t ldo,
                                    const float *__restrict__ gout, int64_t ldg,
                                    const int32_t *__restrict__ count,
                                    float *__restrict__ gdata, int64_t ldd,
                                    int relu_mask) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    const int s = seg[r];
    float g = 0.0f;
    if (s >= 0 && s < nseg) {
      const float v = data[r * ld + c];
      if (v == out[(int64_t)s * ldo + c] && (!relu_mask || v > 0.0f))
        g = gout[(int64_t)s * ldg + c] / (float)count[(int64_t)s * cols + c];
    }
    gdata[r * ldd + c] = g;
  }
}

// float4 forms of the two kernels above (cols % 4 == 0, 16-byte aligned rows):
// a thread owns 4 consecutive columns of one row; the scalar forms read the
// [E, C] matrix at ~1.6 TB/s, these stream it.
__global__ void segmax_count4_kernel(const float *__restrict__ data, int64_t ld,
                                     const int32_t *__restrict__ seg,
                                     int64_t rows, int cols4, int nseg,
                                     const float *__restrict__ out, int64_t ldo,
                                     int32_t *__restrict__ count,
                                     int relu_mask) {
  const int64_t total = rows * cols4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols4;
    const int c = 4 * (int)(idx - r * cols4);
    const int s = seg[r];
    if (s < 0 || s >= nseg) continue;
    const v4f d = *reinterpret_cast<const v4f *>(data + r * ld + c);
    const v4f o = *reinterpret_cast<const v4f *>(out + (int64_t)s * ldo + c);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (d[i] == o[i] && (!relu_mask || d[i] > 0.0f))
        atomicAdd(&count[(int64_t)s * (4 * cols4) + c + i], 1);
  }
}

__global__ void segmax_route4_kernel(const float *__restrict__ data, int64_t ld,
                                     const int32_t *__restrict__ seg,
                                     int64_t rows, int cols4, int nseg,
                                     const float *__restrict__ out, int64_t ldo,
                                     const float *__restrict__ gout, int64_t ldg,
                                     const int32_t *__restrict__ count,
                                     float *__restrict__ gdata, int64_t ldd,
                                     int relu_mask) {
  const int64_t total = rows * cols4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols4;
    const int c = 4 * (int)(idx - r * cols4);
    const int s = seg[r];
    v4f g = (v4f){0.f, 0.f, 0.f, 0.f};
    if (s >= 0 && s < nseg) {
      const v4f d = *reinterpret_cast<const v4f *>(data + r * ld + c);
      const v4f o = *reinterpret_cast<const v4f *>(out + (int64_t)s * ldo + c);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (d[i] == o[i] && (!relu_mask || d[i] > 0.0f))
          g[i] = gout[(int64_t)s * ldg + c + i] /
                 (float)count[(int64_t)s * (4 * cols4) + c + i];
    }
    *reinterpret_cast<v4f *>(gdata + r * ldd + c) = g;
  }
}

// ---------------------------------------------------------------- packing, all layers at once
// One launch re-packs every layer after an SGD step (the per-layer entry took
// 155 launches of ~2 us each, ~0.8 ms of host-paced time per step).  `jobs`
// is a device table built once (the flat parameter buffer never moves).
struct PackJob {
  const float *w;   // [k_in, n_out] row-major inside the flat buffer
  const float *b;   // [n_out] or null
  float *dst;
  int32_t k_in, n_out;
  int32_t kind;     // 0 fragment image, 1 fragment image of W^T (no bias),
                    // 2 plain W^T [n_out rows][ld = 16*ceil(k_in/16)], zero pad
                    // 3 plain W   [k_in rows][ld = 16*ceil(n_out/16)], zero pad
                    // 4 block copy  dst[r*ld + c]  = w[r*n_out + c]
                    // 5 block add   dst[r*n_out + c] += w[r*ld + c]
                    //   (r < k_in, c < n_out; 4 assembles fused matrices from
                    //   parameter blocks, 5 hands their gradients back)
                    // 6 dst[i*ld + i] = 1, i < k_in
  int32_t first_block;  // prefix sum of the jobs' block counts
  int32_t ld;           // kinds 4-6: row stride of the strided side
  int32_t reserved;
};
static_assert(sizeof(PackJob) == 48, "PackJob layout (Python mirrors it)");

__global__ __launch_bounds__(256) void pack_many_kernel(
    const PackJob *__restrict__ jobs, int n_jobs) {
  // binary search of the job this block belongs to
  int lo = 0, hi = n_jobs - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  const PackJob j = jobs[lo];
  const int64_t base = (int64_t)(blk - j.first_block) * 256 + threadIdx.x;
  if (j.kind >= 4) {
    const int64_t total = j.kind == 6 ? j.k_in : (int64_t)j.k_in * j.n_out;
    if (base < total) {
      if (j.kind == 6) {
        j.dst[base * j.ld + base] = 1.0f;
      } else {
        const int r = (int)(base / j.n_out), c = (int)(base - (int64_t)r * j.n_out);
        if (j.kind == 4)
          j.dst[(int64_t)r * j.ld + c] = j.w[base];
        else
          j.dst[base] += j.w[(int64_t)r * j.ld + c];
      }
    }
    return;
  }
  if (j.kind == 3) {
    const int ld = (j.n_out + 15) / 16 * 16;
    const int64_t total = (int64_t)j.k_in * ld;
    if (base < total) {
      const int k = (int)(base / ld), n = (int)(base - (int64_t)k * ld);
      j.dst[base] = n < j.n_out ? j.w[(int64_t)k * j.n_out + n] : 0.0f;
    }
    return;
  }
  if (j.kind == 2) {
    const int ld = (j.k_in + 15) / 16 * 16;
    const int64_t total = (int64_t)j.n_out * ld;
    if (base < total) {
      const int n = (int)(base / ld), k = (int)(base - (int64_t)n * ld);
      j.dst[base] = k < j.k_in ? j.w[(int64_t)k * j.n_out + n] : 0.0f;
    }
    return;
  }
  const int transpose = j.kind;
  const int K = transpose ? j.n_out : j.k_in, N = transpose ? j.k_in : j.n_out;
  const int kq = (K + 15) / 16, nt = (N + 15) / 16;
  const int64_t total = (int64_t)kq * nt * 256;
  const int64_t idx = base;
  if (idx >= total + nt * 16) return;
  if (idx >= total) {
    const int n = (int)(idx - total);
    j.dst[idx] = (!transpose && j.b && n < N) ? j.b[n] : 0.0f;
    return;
  }
  const int sidx = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
  const int64_t qt = idx >> 8;
  const int t = (int)(qt % nt), q = (int)(qt / nt);
  const int k = 16 * q + 4 * (lane >> 4) + sidx, n = 16 * t + (lane & 15);
  float v = 0.0f;
  if (k < K && n < N)
    v = transpose ? j.w[(int64_t)n * j.n_out + k] : j.w[(int64_t)k * j.n_out + n];
  j.dst[idx] = v;
}

// ---------------------------------------------------------------- sparse scatter-max adjoint
// y = scatter_max(Y), Y = ReLU(X W + b)  (gnn.py:269-277 / 357-365).  The
// gradient of the max reaches ONE row per (segment, column) (TF's tie rule
// aside), so dZ = dY * [Y > 0] has ~K*C non-zeros among E*C entries (2-7 per
// row).  The dense adjoint -- dX = dZ W^T and dW = X^T dZ, two E-row GEMMs per
// layer, the largest kernels of the dense training step -- collapses to
//   count_win : winners and tie counts per (segment, column)
//   route     : dX[e,:] = sum over the row's winning columns g * W^T[c,:]
//   wgrad     : dW[:,c] = sum over segments g * X[winner(s,c),:]  (row gathers)
// with g = dY[s,c] / ties(s,c).  Ties between POSITIVE maxima (duplicate
// points) are rare; they are routed exactly (route sees every tied row; a
// separate pass adds their weight-gradient terms and runs only when the
// count pass raised the tie flag).
#ifndef PGNN_CW_ITEMS
#define PGNN_CW_ITEMS 4
#endif
constexpr int kCwItems = PGNN_CW_ITEMS;
__global__ void segmax_count_win4_kernel(
    const float *__restrict__ data, int64_t ld, const int32_t *__restrict__ seg,
    int64_t rows, int cols4, int nseg, const float *__restrict__ out,
    int64_t ldo, int32_t *__restrict__ cnt, int32_t *__restrict__ win,
    int ldc, int32_t *__restrict__ tie_count, int32_t *__restrict__ tie_list,
    int tie_cap) {
  const int64_t total = rows * cols4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // kCwItems items per trip: all segment ids, then all (row, maxima) pairs are
  // in flight together (one item per trip was two dependent round trips per
  // trip: 55 us; two: 52; four: see DESIGN 5)
  constexpr int U = kCwItems;
  for (int64_t idx0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx0 < total; idx0 += U * stride) {
    int64_t r[U];
    int c[U], s[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t idx = idx0 + u * stride;
      ok[u] = idx < total;
      const int64_t ic = ok[u] ? idx : idx0;  // clamped: an unconditional load
      r[u] = ic / cols4;
      c[u] = 4 * (int)(ic - r[u] * cols4);
      s[u] = seg[r[u]];
    }
    v4f d[U], o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = ok[u] && s[u] >= 0 && s[u] < nseg;
      const int sc = ok[u] ? s[u] : 0;
      d[u] = *reinterpret_cast<const v4f *>(data + r[u] * ld + c[u]);
      o[u] = *reinterpret_cast<const v4f *>(out + (int64_t)sc * ldo + c[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (d[u][i] == o[u][i] && d[u][i] > 0.0f) {
          const int slot = atomicAdd(&cnt[(int64_t)s[u] * ldc + c[u] + i], 1);
          if (slot == 0) {
            win[(int64_t)s[u] * ldc + c[u] + i] = (int32_t)r[u];
          } else {  // a further row holding the same positive maximum
            const int p = atomicAdd(tie_count, 1);
            if (p < tie_cap) {
              tie_list[2 * p] = (int32_t)r[u];
              tie_list[2 * p + 1] = c[u] + i;
            }
          }
        }
    }
  }
}

// one wave per row: lanes own columns c = lane + 64 j of Y and features
// k = lane + 64 i of X / dX.  Branch-free on purpose: every load of a phase is
// issued before the first use (clamped addresses + selects instead of
// predicated loads -- with `if (c < cols) load` the compiler serialised the
// loads behind exec-mask branches and a row cost ~30 us of latency).
template <int J /* ceil(cols/64) */, int I /* ceil(dx_cols/64) */>
__global__ __launch_bounds__(256) void segmax_route_sparse_kernel(
    const float *__restrict__ data, int64_t ld, const int32_t *__restrict__ seg,
    int64_t rows, int cols, int nseg, const float *__restrict__ out, int64_t ldo,
    const float *__restrict__ gout, int64_t ldg, const int32_t *__restrict__ cnt,
    int ldc, const float *__restrict__ WT, int64_t ldwt, int k_in,
    const float *__restrict__ X, int64_t ldx, int mask_x,
    float *__restrict__ dX, int64_t lddx, int wdx) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int kc[I];  // clamped feature index of this lane (loads), validity by k < k_in
#pragma unroll
  for (int i = 0; i < I; ++i) kc[i] = min(lane + 64 * i, k_in - 1);
  for (int64_t r = wave; r < rows; r += n_waves) {
    const int s_raw = seg[r];
    const bool s_ok = s_raw >= 0 && s_raw < nseg;
    const int s = s_ok ? s_raw : 0;
    float d[J], o[J], go[J], xv[I];
    int cn[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int cc = min(lane + 64 * j, cols - 1);
      d[j] = data[r * ld + cc];
      o[j] = out[(int64_t)s * ldo + cc];
      go[j] = gout[(int64_t)s * ldg + cc];
      cn[j] = cnt[(int64_t)s * ldc + cc];
    }
#pragma unroll
    for (int i = 0; i < I; ++i) xv[i] = X[r * ldx + kc[i]];
    float g[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool w = s_ok && (lane + 64 * j) < cols && d[j] > 0.0f && d[j] == o[j];
      g[j] = w ? go[j] / (float)max(cn[j], 1) : 0.0f;
    }
    float acc[I];
#pragma unroll
    for (int i = 0; i < I; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      // a winner whose incoming gradient is exactly 0 contributes nothing
      unsigned long long m = __ballot(g[j] != 0.0f);
      while (m) {  // two winners per round: their W^T rows load together
        const int l0 = __builtin_ctzll(m);
        m &= m - 1;
        const int l1 = m ? __builtin_ctzll(m) : l0;
        const float g0 = __shfl(g[j], l0);
        const float g1 = m ? __shfl(g[j], l1) : 0.0f;
        m &= m - 1;
        const float *w0 = WT + (int64_t)(l0 + 64 * j) * ldwt;
        const float *w1 = WT + (int64_t)(l1 + 64 * j) * ldwt;
        float a0[I], a1[I];
#pragma unroll
        for (int i = 0; i < I; ++i) {
          a0[i] = w0[kc[i]];
          a1[i] = w1[kc[i]];
        }
#pragma unroll
        for (int i = 0; i < I; ++i) {
          acc[i] += g0 * a0[i];
          acc[i] += g1 * a1[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < I; ++i) {
      const int k = lane + 64 * i;
      const bool live = k < k_in && (!mask_x || xv[i] > 0.0f);
      if (k < wdx) dX[r * lddx + k] = live ? acc[i] : 0.0f;
    }
  }
}

// Where the layer's input rows X come from: a materialised [rows, ldx] matrix,
// or -- the edge stage after the fused training forward, which never writes
// H1 -- recomputed as ReLU(P[src] - Q[dst]) from the per-vertex tables.
struct XSrc {
  const float *X;
  int64_t ldx;
  const float *P, *Q;
  int64_t ldpq;
  const int32_t *edges;
};
struct XRow {  // row r resolved: x[k] = a[k] - (b ? b[k] : 0), ReLU when b
  const float *a, *b;
};
__device__ __forceinline__ XRow xs_row(const XSrc &x, int64_t r) {
  XRow o;
  if (x.X) {
    o.a = x.X + r * x.ldx;
    o.b = nullptr;
  } else {
    o.a = x.P + (int64_t)x.edges[2 * r] * x.ldpq;
    o.b = x.Q + (int64_t)x.edges[2 * r + 1] * x.ldpq;
  }
  return o;
}
__device__ __forceinline__ float xr_at(const XRow &r, int k) {
  if (!r.b) return r.a[k];
  const float t = r.a[k] - r.b[k];
  return t > 0.0f ? t : 0.0f;
}

// The edge stage's form of the same routing (gnn.py:348-356 adjoint): X = H1
// = ReLU(P[src] - Q[dst]) feeds the layer, so the routed gradient goes
// straight to dP[src] += g, dQ[dst] -= g (pgnn_edge_hidden_bwd) and the E x C
// matrix dH1 is never written.  A wave owns kChunk consecutive rows: the list
// is grouped by dst, so the dQ sum of a run stays in registers and is flushed
// with one row of atomics per run instead of one per edge; dP rows are atomic
// adds (src is unordered).
//
// Round 5: what bounds it is L2 -> L1 traffic, not arithmetic (0.36 GFLOP per
// launch) and not latency: every winner (segment, column) pulls one row of
// W^T (1.2 KB) from L2 -- K C rows = 730 MB per launch at the training step's
// shape, beside 128 MB of Y and ~600 MB of per-row P / Q / out / grad / count
// slices: ~1.5 GB in 160 us = 9 TB/s of gathered 1.2 KB rows.  The traffic is
// inherent to the routed form (each column's gradient goes to ANOTHER row with
// its own gate: nothing accumulates across columns, so there is no GEMM to tile
// W^T for); the dense adjoint it replaces is 18x the FLOPs.  Tried: the chunk's
// index pairs in one load, row r + 1's slices requested before row r is worked
// on, a row's winners of ALL column groups compacted into a per-wave list in
// LDS and taken kWinBatch at a time (one round trip for up to eight W^T rows
// instead of one per pair and column group): 164.8 -> 159.8 us
// (tools/sessions/r05_s10.sh) -- kept, for what it is.
constexpr int kScatterChunk = 16;
#ifndef PGNN_WIN_BATCH
#define PGNN_WIN_BATCH 8
#endif
constexpr int kWinBatch = PGNN_WIN_BATCH;
template <int J, int I>
__global__ __launch_bounds__(256) void segmax_route_scatter_kernel(
    const float *__restrict__ data, int64_t ld, const int32_t *__restrict__ edges,
    int64_t rows, int cols, int nseg, const float *__restrict__ out, int64_t ldo,
    const float *__restrict__ gout, int64_t ldg, const int32_t *__restrict__ cnt,
    int ldc, const float *__restrict__ WT, int64_t ldwt, int k_in, XSrc xs,
    float *__restrict__ dP, float *__restrict__ dQ, int64_t ldpq) {
  __shared__ float s_g[4][64 * J];
  __shared__ int s_c[4][64 * J];
  const int lane = threadIdx.x & 63;
  float *wg = s_g[threadIdx.x >> 6];
  int *wc = s_c[threadIdx.x >> 6];
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int kc[I], cj[J];
#pragma unroll
  for (int i = 0; i < I; ++i) kc[i] = min(lane + 64 * i, k_in - 1);
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = min(lane + 64 * j, cols - 1);
  struct RowIn {
    float d[J], o[J], go[J], xa[I], xb[I];
    int cn[J];
    int src, s, ok;
  };
  const int64_t n_chunks = (rows + kScatterChunk - 1) / kScatterChunk;
  for (int64_t ch = wave; ch < n_chunks; ch += n_waves) {
    const int64_t r0 = ch * kScatterChunk;
    const int64_t r1 = min(r0 + (int64_t)kScatterChunk, rows);
    // lane i: (src, dst) of row r0 + i
    int my_src = 0, my_dst = -1;
    if (lane < kScatterChunk && r0 + lane < r1) {
      my_src = edges[2 * (r0 + lane)];
      my_dst = edges[2 * (r0 + lane) + 1];
    }
    auto fetch = [&](int64_t r, RowIn &in) {
      const int i = (int)(r - r0);
      const int src = __builtin_amdgcn_readlane(my_src, i);
      const int s_raw = __builtin_amdgcn_readlane(my_dst, i);
      in.ok = (s_raw >= 0 && s_raw < nseg && src >= 0 && src < nseg) ? 1 : 0;
      in.src = src;
      in.s = in.ok ? s_raw : 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        in.d[j] = data[r * ld + cj[j]];
        in.o[j] = out[(int64_t)in.s * ldo + cj[j]];
        in.go[j] = gout[(int64_t)in.s * ldg + cj[j]];
        in.cn[j] = cnt[(int64_t)in.s * ldc + cj[j]];
      }
      if (xs.X) {
        const float *xa = xs.X + r * xs.ldx;
#pragma unroll
        for (int i2 = 0; i2 < I; ++i2) in.xa[i2] = xa[kc[i2]], in.xb[i2] = 0.0f;
      } else {
        const float *xa = xs.P + (int64_t)(in.ok ? src : 0) * xs.ldpq;
        const float *xb = xs.Q + (int64_t)in.s * xs.ldpq;
#pragma unroll
        for (int i2 = 0; i2 < I; ++i2) in.xa[i2] = xa[kc[i2]], in.xb[i2] = xb[kc[i2]];
      }
    };
    float qacc[I];
#pragma unroll
    for (int i = 0; i < I; ++i) qacc[i] = 0.0f;
    int q_dst = -1;
    RowIn cur;
    fetch(r0, cur);
    for (int64_t r = r0; r < r1; ++r) {
      RowIn nxt = cur;
      if (r + 1 < r1) fetch(r + 1, nxt);  // in flight while this row is worked on
      // the row's winning columns, all groups, compacted: (column, g)
      int n_win = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const bool w = cur.ok && (lane + 64 * j) < cols && cur.d[j] > 0.0f &&
                       cur.d[j] == cur.o[j];
        const float g = w ? cur.go[j] / (float)max(cur.cn[j], 1) : 0.0f;
        const unsigned long long m = __ballot(g != 0.0f);
        if (g != 0.0f) {
          const int pos = n_win + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          wg[pos] = g;
          wc[pos] = lane + 64 * j;
        }
        n_win += __builtin_popcountll(m);
      }
      // (one wave: its LDS operations execute in order; the compiler must not
      // move the reads below above the writes)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float acc[I];
#pragma unroll
      for (int i = 0; i < I; ++i) acc[i] = 0.0f;
      for (int b = 0; b < n_win; b += kWinBatch) {  // wave-uniform
        float gv[kWinBatch], a[kWinBatch][I];
#pragma unroll
        for (int u = 0; u < kWinBatch; ++u) {
          const int idx = b + u < n_win ? b + u : n_win - 1;
          gv[u] = b + u < n_win ? wg[idx] : 0.0f;
          const float *wr = WT + (int64_t)wc[idx] * ldwt;
#pragma unroll
          for (int i = 0; i < I; ++i) a[u][i] = wr[kc[i]];
        }
#pragma unroll
        for (int u = 0; u < kWinBatch; ++u)
#pragma unroll
          for (int i = 0; i < I; ++i) acc[i] += gv[u] * a[u][i];
      }
      asm volatile("" ::: "memory");  // the list is rewritten by the next row
      if (cur.s != q_dst) {  // wave-uniform: the dst run ended
        if (q_dst >= 0) {
#pragma unroll
          for (int i = 0; i < I; ++i) {
            const int k = lane + 64 * i;
            if (k < k_in && qacc[i] != 0.0f)
              atomicAdd(&dQ[(int64_t)q_dst * ldpq + k], -qacc[i]);
            qacc[i] = 0.0f;
          }
        }
        q_dst = cur.s;
      }
      if (n_win > 0) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < I; ++i) {
          const int k = lane + 64 * i;
          // x = ReLU(a - b) (b = 0 for materialised rows): the gate is x > 0
          const float x = cur.xa[i] - cur.xb[i];
          const float v = (k < k_in && x > 0.0f) ? acc[i] : 0.0f;
          if (v != 0.0f) {
            atomicAdd(&dP[(int64_t)cur.src * ldpq + k], v);
            qacc[i] += v;
          }
        }
      }
      cur = nxt;
    }
    if (q_dst >= 0) {
#pragma unroll
      for (int i = 0; i < I; ++i) {
        const int k = lane + 64 * i;
        if (k < k_in && qacc[i] != 0.0f)
          atomicAdd(&dQ[(int64_t)q_dst * ldpq + k], -qacc[i]);
      }
    }
  }
}

// grid (column, slice): dW^T partial [slice][c][k] and db partial [slice][c]
// from the recorded winner of every (segment, column); the other rows of a
// tie are left to the tie passes
template <int I /* ceil(kin_p/64) */>
__global__ __launch_bounds__(64) void segmax_wgrad_gather_kernel(
    const float *__restrict__ gout, int64_t ldg, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ win, int ldc, int nseg, int seg_per_slice,
    XSrc xs, int k_in, int kin_p, float *__restrict__ partial,
    float *__restrict__ partial_b, int cols) {
  const int lane = threadIdx.x;
  const int c = blockIdx.x, slice = blockIdx.y;
  const int s0 = slice * seg_per_slice;
  int s1 = s0 + seg_per_slice;
  if (s1 > nseg) s1 = nseg;
  float acc[I];
#pragma unroll
  for (int i = 0; i < I; ++i) acc[i] = 0.0f;
  float bsum = 0.0f;
  for (int sb = s0; sb < s1; sb += 64) {
    const int s = sb + lane;
    float g = 0.0f;
    int e = 0;
    if (s < s1) {
      // the recorded (first) winner takes its share here; further rows tied
      // with it are on the tie list
      const int n = cnt[(int64_t)s * ldc + c];
      if (n >= 1) {
        g = gout[(int64_t)s * ldg + c] / (float)n;
        e = win[(int64_t)s * ldc + c];
      }
    }
    bsum += g;
    const int nl = min(64, s1 - sb);
    // rows are independent gathers: four in flight per wave
    for (int l0 = 0; l0 < nl; l0 += 4) {
      float gv[4];
      XRow xr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gv[u] = __shfl(g, (l0 + u) & 63);
        xr[u] = xs_row(xs, (int64_t)__shfl(e, (l0 + u) & 63));
      }
      float xv[4][I];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < I; ++i) {
          const int k = lane + 64 * i;
          xv[u][i] = xr_at(xr[u], min(k, k_in - 1));
        }
#pragma unroll
      for (int i = 0; i < I; ++i)
        if (lane + 64 * i >= k_in)
#pragma unroll
          for (int u = 0; u < 4; ++u) xv[u][i] = 0.0f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < I; ++i) acc[i] += gv[u] * xv[u][i];
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) bsum += __shfl_xor(bsum, d);
  float *po = partial + ((int64_t)slice * cols + c) * kin_p;
#pragma unroll
  for (int i = 0; i < I; ++i) {
    const int k = lane + 64 * i;
    if (k < kin_p) po[k] = acc[i];
  }
  if (lane == 0) partial_b[(int64_t)slice * cols + c] = bsum;
}

// dW[k, c] += sum_slices partial[slice][c][k]; db[c] += sum_slices pb[slice][c].
// Thread per (c, k) with k fastest: partial's rows are [c][k], so a wave reads
// 64 consecutive floats per slice (with c fastest every lane touched a line of
// its own), four slices in flight; the one store per output is the strided side.
__global__ void segmax_wgrad_reduce_kernel(const float *__restrict__ partial,
                                           const float *__restrict__ partial_b,
                                           int slices, int cols, int k_in,
                                           int kin_p, float *__restrict__ dW,
                                           float *__restrict__ db) {
  const int kk = k_in + 1;
  const int64_t total = (int64_t)kk * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / kk), k = (int)(idx - (int64_t)c * kk);
    const bool bias = k == k_in;
    if (bias && !db) continue;
    const float *__restrict__ src =
        bias ? partial_b + c : partial + (int64_t)c * kin_p + k;
    const int64_t step = bias ? (int64_t)cols : (int64_t)cols * kin_p;
    float s = 0.0f;
    int sl = 0;
    for (; sl + 3 < slices; sl += 4) {
      const float a0 = src[(int64_t)sl * step];
      const float a1 = src[(int64_t)(sl + 1) * step];
      const float a2 = src[(int64_t)(sl + 2) * step];
      const float a3 = src[(int64_t)(sl + 3) * step];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; sl < slices; ++sl) s += src[(int64_t)sl * step];
    if (bias) db[c] += s;
    else dW[(int64_t)k * cols + c] += s;
  }
}

// weight-gradient terms of the further rows of tied positive maxima (rare:
// duplicate points; float atomics).  One launch: while the tie list holds them
// all, one wave per list entry (row r, column c); when it overflowed, a full
// scan (every tied row except the recorded winner).
__global__ __launch_bounds__(256) void segmax_wgrad_ties_kernel(
    const int32_t *__restrict__ tie_count, const int32_t *__restrict__ tie_list,
    int tie_cap, const float *__restrict__ data, int64_t ld,
    const int32_t *__restrict__ seg, int64_t rows, int cols, int nseg,
    const float *__restrict__ out, int64_t ldo, const float *__restrict__ gout,
    int64_t ldg, const int32_t *__restrict__ cnt, const int32_t *__restrict__ win,
    int ldc, XSrc xs, int k_in, float *__restrict__ dW, float *__restrict__ db) {
  const int n_ties = *tie_count;
  if (n_ties <= tie_cap) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int t = wave; t < n_ties; t += n_waves) {
      const int r = tie_list[2 * t], c = tie_list[2 * t + 1];
      const int s = seg[r];
      const float g =
          gout[(int64_t)s * ldg + c] / (float)cnt[(int64_t)s * ldc + c];
      const XRow xr = xs_row(xs, r);
      for (int k = lane; k < k_in; k += 64)
        atomicAdd(&dW[(int64_t)k * cols + c], g * xr_at(xr, k));
      if (db && lane == 0) atomicAdd(&db[c], g);
    }
    return;
  }
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    const int s = seg[r];
    if (s < 0 || s >= nseg) continue;
    const float d = data[r * ld + c];
    if (!(d > 0.0f) || d != out[(int64_t)s * ldo + c]) continue;
    const int n = cnt[(int64_t)s * ldc + c];
    if (n <= 1 || win[(int64_t)s * ldc + c] == (int32_t)r) continue;
    const float g = gout[(int64_t)s * ldg + c] / (float)n;
    const XRow xr = xs_row(xs, r);
    for (int k = 0; k < k_in; ++k)
      atomicAdd(&dW[(int64_t)k * cols + c], g * xr_at(xr, k));
    if (db) atomicAdd(&db[c], g);
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[i][j] = sum_r X[r][i] dZ[r][j], db[j] = sum_r dZ[r][j] (a constant-one
// input column at index k_in produces db as an extra row of the same GEMM).
// Grid (in-blocks of 64 inputs, row slices); each workgroup accumulates its
// [64 x <=320] block over the slice with MFMA (the reduction index is the data
// row), writes one partial; a second kernel sums the slices in a fixed order
// (deterministic, no float atomics).
constexpr int kWgRows = 32;  // data rows staged per step

template <int NT>
__device__ __forceinline__ void weight_grad_body(
    const float *__restrict__ X, int64_t ldx, int k_in,
    const float *__restrict__ dZ, int64_t ldz, int n_out, int64_t rows,
    int64_t rows_per_slice, int nt, float *__restrict__ partial, int ib,
    int slice, int n_in_blocks,
    float *__restrict__ bias_partial = nullptr /* [slices][16 nt]: when given,
        db is summed here by input block 0 from the staged dZ tile instead of
        riding as input column k_in -- for k_in % 64 == 0 that column would
        be an input block of its own re-reading all of dZ */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ldxs = 68, ldzs = 16 * nt + 4;
  float *Xs = reinterpret_cast<float *>(smem);
  float *Zs = Xs + kWgRows * ldxs;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int in0 = 64 * ib;
  const int64_t r_begin = (int64_t)slice * rows_per_slice;
  int64_t r_end = r_begin + rows_per_slice;
  if (r_end > rows) r_end = rows;
  v4f acc[4][NT];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[m][j] = (v4f){0.f, 0.f, 0.f, 0.f};
  int toff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int t = wave + 4 * j;
    if (t > nt - 1) t = nt - 1;
    toff[j] = 16 * t + (lane & 15);
  }
  float bsum[2] = {0.0f, 0.0f};
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kWgRows) {
    __syncthreads();
    // Staging.  Every load is UNCONDITIONAL (clamped row / column) and the
    // validity select happens on the loaded value: a load under an `if` (or a
    // select the compiler turns into one) is waited for on the spot, and the
    // tile's 40-odd loads per thread then go out one round trip at a time.
    {  // X tile: thread -> input column (t & 63), rows (t >> 6) + 4 i
      const int c = threadIdx.x & 63, col = in0 + c;
      const int colc = col < k_in ? col : 0;
      float xv[kWgRows / 4];
#pragma unroll
      for (int i = 0; i < kWgRows / 4; ++i) {
        const int64_t row = r0 + (threadIdx.x >> 6) + 4 * i;
        xv[i] = X[(row < r_end ? row : r_end - 1) * ldx + colc];
      }
#pragma unroll
      for (int i = 0; i < kWgRows / 4; ++i) {
        const int r = (threadIdx.x >> 6) + 4 * i;
        const bool ok = r0 + r < r_end;
        // column k_in is the bias "input": a constant one
        Xs[r * ldxs + c] =
            !ok ? 0.0f
                : (col < k_in ? xv[i]
                              : ((col == k_in && !bias_partial) ? 1.0f : 0.0f));
      }
    }
    // dZ tile: thread (t >> 6, t & 63) takes rows (t >> 6) + 4 i and columns
    // (t & 63) + 64 j -- 64 consecutive floats of a row per wave, every thread
    // the same number of independent loads
    const int zc = 16 * nt;
    {
      const int cl = threadIdx.x & 63, rs = threadIdx.x >> 6;
      for (int c = cl; c < zc; c += 64) {
        const bool in_range = c < n_out;
        const int cc = in_range ? c : 0;
        float zv[kWgRows / 4];
#pragma unroll
        for (int i = 0; i < kWgRows / 4; ++i) {
          const int64_t row = r0 + rs + 4 * i;
          zv[i] = dZ[(row < r_end ? row : r_end - 1) * ldz + cc];
        }
#pragma unroll
        for (int i = 0; i < kWgRows / 4; ++i) {
          const int r = rs + 4 * i;
          Zs[r * ldzs + c] = (in_range && r0 + r < r_end) ? zv[i] : 0.0f;
        }
      }
    }
    __syncthreads();
    if (bias_partial && ib == 0) {
      // column sums of the staged dZ tile (rows past the end are zero)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = threadIdx.x + 256 * h;
        if (c < 16 * nt) {
          float sum = 0.0f;
#pragma unroll 8
          for (int r = 0; r < kWgRows; ++r) sum += Zs[r * ldzs + c];
          bsum[h] += sum;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kWgRows / 16; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = 16 * q + 4 * (lane >> 4) + s;
        float a[4], b[NT];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = Xs[r * ldxs + 16 * m + (lane & 15)];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = Zs[r * ldzs + toff[j]];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[j],
                                                             acc[m][j], 0, 0, 0);
      }
    }
  }
  if (bias_partial && ib == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = threadIdx.x + 256 * h;
      if (c < 16 * nt) bias_partial[(int64_t)slice * 16 * nt + c] = bsum[h];
    }
  }
  // partial[slice][in][out], in padded to 64*gridDim.x, out padded to 16*nt
  const int64_t kin_p = 64 * (int64_t)n_in_blocks, nout_p = 16 * nt;
  float *po = partial + (int64_t)slice * kin_p * nout_p;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = wave + 4 * j;
    if (t < nt) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = in0 + 16 * m + 4 * (lane >> 4) + r;
          po[(int64_t)i * nout_p + 16 * t + (lane & 15)] = acc[m][j][r];
        }
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void weight_grad_kernel(
    const float *__restrict__ X, int64_t ldx, int k_in,
    const float *__restrict__ dZ, int64_t ldz, int n_out, int64_t rows,
    int64_t rows_per_slice, int nt, float *__restrict__ partial,
    float *__restrict__ bias_partial) {
  weight_grad_body<NT>(X, ldx, k_in, dZ, ldz, n_out, rows, rows_per_slice, nt,
                       partial, (int)blockIdx.x, (int)blockIdx.y,
                       (int)gridDim.x, bias_partial);
}

// ---- many small weight gradients in ONE launch -------------------------------
// The K-row layers of a training step (update / offset / P layers of every
// iteration, the pooling output layer, the heads: ~25 GEMMs of
// [300 x 2000] x [2000 x 300]) each fill the chip for ~25 us plus a ~12 us
// reduce when launched one by one -- 0.7 ms of a 4.7 ms step in latency-bound
// launches.  None of them is on the backward's critical path (only dX is), so
// the trainer records them and runs them together at the end: one launch whose
// workgroups are (job, input block, row slice) triples, one reduce launch.
constexpr int kWgManyJobs = 24;  // jobs per launch (kernel-argument size)
struct WgJobDev {
  const float *X, *dZ;
  float *dW, *db;
  int64_t ldx, ldz, rows, rps, part_off /* floats */, out0 /* first output */;
  int k_in, n_out, nt, in_blocks, slices, wg0, accumulate, pad;
};
struct WgJobsDev {
  int n, total_wgs;
  int64_t total_out;
  WgJobDev j[kWgManyJobs];
};

__global__ __launch_bounds__(256) void weight_grad_many_kernel(
    WgJobsDev js, float *__restrict__ partial) {
  int ji = 0;
  while (ji + 1 < js.n && (int)blockIdx.x >= js.j[ji + 1].wg0) ++ji;
  const WgJobDev &J = js.j[ji];
  const int local = (int)blockIdx.x - J.wg0;
  const int ib = local % J.in_blocks, slice = local / J.in_blocks;
  weight_grad_body<5>(J.X, J.ldx, J.k_in, J.dZ, J.ldz, J.n_out, J.rows, J.rps,
                      J.nt, partial + J.part_off, ib, slice, J.in_blocks);
}

// the reduce of weight_grad_reduce_kernel over the outputs of all jobs
__global__ __launch_bounds__(256) void weight_grad_reduce_many_kernel(
    WgJobsDev js, const float *__restrict__ partial) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < js.total_out;
       base += (int64_t)gridDim.x * 64) {
    const int64_t gidx = base + o;
    int ji = 0;
    while (ji + 1 < js.n && gidx >= js.j[ji + 1].out0) ++ji;
    const WgJobDev &J = js.j[ji];
    const int64_t idx = gidx - J.out0;
    const int64_t total = (int64_t)(J.k_in + 1) * J.n_out;
    const bool ok = gidx < js.total_out && idx < total;
    const int64_t i = ok ? idx / J.n_out : 0;
    const int j = ok ? (int)(idx - i * J.n_out) : 0;
    float s = 0.0f;
    if (ok) {
      const int64_t kin_p = 64 * (int64_t)J.in_blocks;
      const int nout_p = 16 * J.nt;
      const float *pj = partial + J.part_off;
      for (int sl = g; sl < J.slices; sl += 4)
        s += pj[((int64_t)sl * kin_p + i) * nout_p + j];
    }
    part[g][o] = s;
    __syncthreads();
    if (g == 0 && ok) {
      s = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
      if (i < J.k_in) {
        float *w = J.dW + i * J.n_out + j;
        *w = J.accumulate ? *w + s : s;
      } else if (J.db) {
        J.db[j] = J.accumulate ? J.db[j] + s : s;
      }
    }
    __syncthreads();
  }
}

// 64 outputs x G slice groups per workgroup: group g adds slices g, g+G, ...
// (four independent loads in flight per thread); the G partial sums meet in
// LDS and are added in a fixed order, so the result does not depend on timing.
// (A thread per output walking all slices serially ran at 1.8 TB/s: too few
// threads for ~150 slices.  G = 4 was still a chain of 192 dependent adds per
// thread for the 768 slices of pool_narrow_bwd_kernel -- three launches of
// ~50 us for 33 MB: G = 16 there.)
template <int G>
__global__ __launch_bounds__(64 * G) void weight_grad_reduce_kernel(
    const float *__restrict__ partial, int slices, int64_t kin_p, int nout_p,
    int k_in, int n_out, int64_t ld_dw, float *__restrict__ dW,
    float *__restrict__ db, int accumulate,
    const float *__restrict__ bias_partial /* nullable, see the kernel */) {
  __shared__ float part[G][64];
  const int64_t total = (int64_t)(k_in + 1) * n_out;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total;
       base += (int64_t)gridDim.x * 64) {
    const int64_t idx = base + o;
    const int64_t i = idx / n_out;
    const int j = (int)(idx - i * n_out);
    float s = 0.0f;
    if (idx < total) {
      const bool bias = bias_partial && i == k_in;
      const float *__restrict__ src =
          bias ? bias_partial + j : partial + i * nout_p + j;
      const int64_t step = bias ? (int64_t)nout_p : kin_p * nout_p;
      int sl = g;
      for (; sl + 3 * G < slices; sl += 4 * G) {
        const float a0 = src[(int64_t)sl * step];
        const float a1 = src[(int64_t)(sl + G) * step];
        const float a2 = src[(int64_t)(sl + 2 * G) * step];
        const float a3 = src[(int64_t)(sl + 3 * G) * step];
        s = (((s + a0) + a1) + a2) + a3;
      }
      for (; sl < slices; sl += G) s += src[(int64_t)sl * step];
    }
    part[g][o] = s;
    __syncthreads();
    if (g == 0 && idx < total) {
      s = part[0][o];
#pragma unroll
      for (int q = 1; q < G; ++q) s += part[q][o];
      if (i < k_in) {
        float *w = dW + i * ld_dw + j;
        *w = accumulate ? *w + s : s;
      } else if (db) {
        db[j] = accumulate ? db[j] + s : s;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- loss
// models.py:212-255 with loc_loss_type 'huber_loss' and cls_loss_type 'softmax'
// (cls_kind 0), 'focal_softmax' (1) or 'focal_sigmoid' (2).
// Per vertex: ce = logsumexp(z) - z[label]; loc = mean_7 huber(pred[label] - gt)
// * valid.  sums[0] += ce, sums[1] += loc, sums[2] += 1, sums[3] += valid.
// Gradients of (cls_scale * sum ce + loc_scale * sum loc) are written out.
// (1 - p_t)^gamma and its derivative gamma (1 - p_t)^(gamma - 1) at om = 1 - p_t
// >= 0.  TF's pow(0, 0) is 1: gamma == 0 turns the focal losses into plain CE /
// sigmoid CE (factor 1, derivative 0).  For 0 < gamma < 1 the derivative is
// unbounded as om -> 0; om is held at 1e-6 there (a vertex classified that
// surely carries no loss either way).
__device__ __forceinline__ float focal_mod(float om, float gamma) {
  if (gamma == 0.0f) return 1.0f;
  return om > 0.0f ? powf(om, gamma) : 0.0f;
}
__device__ __forceinline__ float focal_dmod(float om, float gamma) {
  if (gamma == 0.0f) return 0.0f;
  if (gamma < 1.0f) return gamma * powf(fmaxf(om, 1e-6f), gamma - 1.0f);
  return om > 0.0f ? gamma * powf(om, gamma - 1.0f) : 0.0f;
}

__global__ void loss_kernel(const float *__restrict__ logits, int64_t ldl,
                            const int32_t *__restrict__ labels,
                            const float *__restrict__ pred, int box_len,
                            const float *__restrict__ gt,
                            const float *__restrict__ valid, int64_t n, int nc,
                            float cls_scale, float loc_scale,
                            double *__restrict__ sums,
                            float *__restrict__ dlogits,
                            float *__restrict__ dpred,
                            const double *__restrict__ counts, double cls_w,
                            double loc_w, int cls_kind, float alpha, float gamma,
                            const float *__restrict__ class_loc_w,
                            const float *__restrict__ sel_cls,
                            const float *__restrict__ sel_loc, float ce_weight,
                            float *__restrict__ point_cls,
                            float *__restrict__ point_loc) {
  if (counts) {
    // the GLOBAL endpoint counts are on the device (all-reduced there): the
    // scales are formed here as the host form does (double quotient of the
    // loss weight and the count, rounded to float once)
    const double nt = counts[0], nv = counts[1];
    cls_scale = nt > 0.0 ? (float)(cls_w * (double)ce_weight / nt) : 0.0f;
    loc_scale = nv > 0.0 ? (float)(loc_w / nv) : 0.0f;  // div_no_nan
  }
  double s_ce = 0.0, s_loc = 0.0, s_n = 0.0, s_v = 0.0;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    const float *z = logits + v * ldl;
    const int lab = labels[v];
    float zmax = z[0];
    for (int c = 1; c < nc; ++c) zmax = fmaxf(zmax, z[c]);
    float se = 0.0f;
    for (int c = 0; c < nc; ++c) se += expf(z[c] - zmax);
    const float lse = logf(se) + zmax;
    // top-k selections (models.py:222-228, 266-291): a vertex outside the
    // selection contributes nothing -- no loss, no gradient
    const float pick_c = sel_cls ? sel_cls[v] : 1.0f;
    const float pick_l = sel_loc ? sel_loc[v] : 1.0f;
    if (cls_kind == 0) {
      const float ce = lse - z[lab];
      if (point_cls) point_cls[v] = ce;
      s_ce += (double)(pick_c * ce_weight * ce);
      if (dlogits)
        for (int c = 0; c < nc; ++c)
          dlogits[v * nc + c] = pick_c * cls_scale *
              (expf(z[c] - lse) - (c == lab ? 1.0f : 0.0f));
    } else if (cls_kind == 1) {
      // focal_loss_softmax (models/loss.py:31-48): L = (1 - p_y)^gamma * CE
      const float ce = lse - z[lab];
      const float py = expf(z[lab] - lse);
      const float om = fmaxf(1.0f - py, 0.0f);
      const float mod = focal_mod(om, gamma);
      s_ce += (double)(mod * ce);
      if (point_cls) point_cls[v] = mod * ce;
      if (dlogits) {
        // dL/dp_y = -gamma (1-p_y)^(gamma-1) CE - (1-p_y)^gamma / p_y;
        // dp_y/dz_c = p_y ([c == y] - p_c)
        const float dm = focal_dmod(om, gamma);
        const float a = -(dm * ce * py + mod);
        for (int c = 0; c < nc; ++c)
          dlogits[v * nc + c] = cls_scale * a *
                                ((c == lab ? 1.0f : 0.0f) - expf(z[c] - lse));
      }
    } else {
      // focal_loss_sigmoid (models/loss.py:5-29), mean over the classes too
      // (models.py:229 reduce_mean over [N, nc])
      float sum = 0.0f;
      for (int c = 0; c < nc; ++c) {
        const float zc = z[c];
        const bool t = c == lab;
        const float p = 1.0f / (1.0f + expf(-zc));
        const float xent = fmaxf(zc, 0.0f) - (t ? zc : 0.0f) +
                           log1pf(expf(-fabsf(zc)));
        const float om = t ? 1.0f - p : p;            // 1 - p_t
        const float mod = focal_mod(om, gamma);
        const float aw = t ? alpha : 1.0f - alpha;
        sum += mod * aw * xent;
        if (dlogits) {
          // d(1 - p_t)/dz = -(2t - 1) p (1 - p)
          const float dm = focal_dmod(om, gamma) *
                           (t ? -1.0f : 1.0f) * p * (1.0f - p);
          dlogits[v * nc + c] = cls_scale / (float)nc * aw *
                                (dm * xent + mod * (p - (t ? 1.0f : 0.0f)));
        }
      }
      s_ce += (double)(sum / (float)nc);
      if (point_cls) point_cls[v] = sum / (float)nc;
    }
    const float w = pick_l * valid[v] * (class_loc_w ? class_loc_w[lab] : 1.0f);
    const float *p = pred + (v * nc + lab) * box_len;
    const float *g = gt + v * box_len;
    float acc = 0.0f;
    if (dpred)
      for (int i = 0; i < nc * box_len; ++i) dpred[v * nc * box_len + i] = 0.0f;
    for (int k = 0; k < box_len; ++k) {
      const float err = p[k] - g[k];
      const float ae = fabsf(err);
      const float quad = fminf(ae, 1.0f);
      const float lin = ae - quad;
      acc += (0.5f * quad * quad + lin) * w;
      if (dpred) {
        // d/derr: err where |err| <= 1, sign(err) beyond (tf.losses.huber_loss)
        const float d = ae <= 1.0f ? err : (err > 0.0f ? 1.0f : -1.0f);
        dpred[(v * nc + lab) * box_len + k] = loc_scale * w * d / (float)box_len;
      }
    }
    s_loc += (double)(acc / (float)box_len);
    if (point_loc) point_loc[v] = acc / (float)box_len;
    s_n += 1.0;
    s_v += (double)(pick_l * valid[v]);
  }
  // block reduction (fp64 atomics at the end: 4 per wave)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s_ce += __shfl_xor(s_ce, d);
    s_loc += __shfl_xor(s_loc, d);
    s_n += __shfl_xor(s_n, d);
    s_v += __shfl_xor(s_v, d);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sums[0], s_ce);
    atomicAdd(&sums[1], s_loc);
    atomicAdd(&sums[2], s_n);
    atomicAdd(&sums[3], s_v);
  }
}

__global__ void sgd_kernel(float *__restrict__ w, const float *__restrict__ g,
                           const float *__restrict__ is_weight, int64_t n,
                           float lr, float grad_scale, float l1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float wi = w[i];
    float gi = g[i] * grad_scale;
    if (is_weight[i] != 0.0f)  // l1_regularizer: d|w| = sign(w) (0 at 0)
      gi += l1 * (wi > 0.0f ? 1.0f : (wi < 0.0f ? -1.0f : 0.0f));
    w[i] = wi - lr * gi;
  }
}

// The other optimizers of train.py:380-391 on the flat parameter buffer, with
// TF 1.x's update rules (training_ops.cc: ApplyMomentum without Nesterov,
// ApplyRMSProp (not centered), ApplyAdam): the gradient is the one sgd_kernel
// applies (scaled batch gradient + the L1 regulariser's sign term), the slot
// buffers have the parameters' layout.
//   kind 1 momentum: a = h0 a + g;                  w -= lr a
//   kind 2 rmsprop : ms = h1 ms + (1 - h1) g^2;
//                    mom = h0 mom + lr g / sqrt(ms + h2);   w -= mom
//   kind 3 adam    : m = h0 m + (1 - h0) g;  v = h1 v + (1 - h1) g^2;
//                    w -= lr m / (sqrt(v) + h2)     (lr: bias-corrected by the caller)
__global__ void optimizer_kernel(int kind, float *__restrict__ w,
                                 const float *__restrict__ g,
                                 const float *__restrict__ is_weight,
                                 float *__restrict__ s0, float *__restrict__ s1,
                                 int64_t n, float lr, float grad_scale, float l1,
                                 float h0, float h1, float h2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float wi = w[i];
    float gi = g[i] * grad_scale;
    if (is_weight[i] != 0.0f)
      gi += l1 * (wi > 0.0f ? 1.0f : (wi < 0.0f ? -1.0f : 0.0f));
    if (kind == 1) {
      const float a = h0 * s0[i] + gi;
      s0[i] = a;
      w[i] = wi - lr * a;
    } else if (kind == 2) {
      const float ms = h1 * s0[i] + (1.0f - h1) * gi * gi;
      const float mom = h0 * s1[i] + lr * gi / sqrtf(ms + h2);
      s0[i] = ms;
      s1[i] = mom;
      w[i] = wi - mom;
    } else {
      const float m = h0 * s0[i] + (1.0f - h0) * gi;
      const float v = h1 * s1[i] + (1.0f - h1) * gi * gi;
      s0[i] = m;
      s1[i] = v;
      w[i] = wi - lr * m / (sqrtf(v) + h2);
    }
  }
}

__global__ __launch_bounds__(1024) void l1_norm_kernel(
    const float *__restrict__ w, const float *__restrict__ is_weight, int64_t n,
    double *__restrict__ out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (is_weight[i] != 0.0f) s += (double)fabsf(w[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  // one fp64 atomic per 16-wave workgroup, <= 128 of them (one per wave
  // serialised 23k atomics on a single address: 56 us for 1.5 M parameters;
  // one per 4-wave workgroup, 1024 of them, still 18 us)
  __shared__ double part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += part[q];
    atomicAdd(out, t);
  }
}

// two 16-byte-aligned regions zeroed by ONE launch (n0, n1 in 16-byte words)
__global__ void zero2_kernel(uint4 *__restrict__ a, int64_t n0,
                             uint4 *__restrict__ b, int64_t n1) {
  const uint4 z = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n0) a[i] = z;
    else b[i - n0] = z;
  }
}

// dP and dQ zeroed by one fill when the caller laid them out back to back
// (the trainer does): a fill is a launch, and the step is a chain of them.
inline hipError_t zero_pair(float *a, float *b, size_t n, hipStream_t stream) {
  if (b == a + n) return hipMemsetAsync(a, 0, 2 * n * 4, stream);
  hipError_t e = hipMemsetAsync(a, 0, n * 4, stream);
  return e != hipSuccess ? e : hipMemsetAsync(b, 0, n * 4, stream);
}

inline unsigned grid_for(int64_t total, int cap = 4096) {
  int64_t b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int pgnn_pack_fc_device(const float *w, const float *b, int32_t k_in,
                                   int32_t n_out, int32_t transpose,
                                   float *packed, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(w && packed && k_in > 0 && n_out > 0, PGNN_E_INVALID,
               "pack_fc_device: bad argument");
  const size_t total = transpose ? pgnn_packed_fc_floats(n_out, k_in)
                                 : pgnn_packed_fc_floats(k_in, n_out);
  hipLaunchKernelGGL(pack_fc_device_kernel, dim3(grid_for((int64_t)total)),
                     dim3(256), 0, (hipStream_t)stream_, w, b, k_in, n_out,
                     transpose, packed);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_pack_fc_many(const void *jobs_device, int32_t n_jobs,
                                 int32_t total_blocks, void *stream_) {
  PGNN_GUARD_BEGIN
  if (n_jobs <= 0 || total_blocks <= 0) return 0;
  PGNN_REQUIRE(jobs_device, PGNN_E_INVALID, "pack_fc_many: null job table");
  hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0,
                     (hipStream_t)stream_,
                     reinterpret_cast<const PackJob *>(jobs_device), n_jobs);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

namespace {
constexpr int kTieCap = 65536;
struct SegFcWs {
  int32_t *cnt, *win, *tie, *tie_list;
  float *partial, *partial_b;
  int ldc, slices, seg_per_slice, kin_p;
};
int segfc_slices(int32_t num_segments, int32_t n_cols) {
  // (columns x slices) waves: enough to fill the chip a few times over
  int s = (8 * 256 * 4 + n_cols - 1) / n_cols;
  const int max_s = (num_segments + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}
size_t segfc_carve(void *ws, int64_t n_rows, int32_t n_cols,
                   int32_t num_segments, int32_t k_in, SegFcWs *o) {
  const int ldc = (n_cols + 3) / 4 * 4;
  const int slices = segfc_slices(num_segments, n_cols);
  const int kin_p = (k_in + 15) / 16 * 16;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) / 256 * 256;
    return at;
  };
  const size_t a_cnt = take((size_t)num_segments * ldc * 4 + 256);  // + tie flag
  const size_t a_win = take((size_t)num_segments * ldc * 4);
  const size_t a_par = take((size_t)slices * n_cols * kin_p * 4);
  const size_t a_pb = take((size_t)slices * n_cols * 4);
  const size_t a_tl = take((size_t)kTieCap * 2 * 4);
  if (o) {
    char *b = (char *)ws;
    o->cnt = (int32_t *)(b + a_cnt);
    o->tie = o->cnt + (size_t)num_segments * ldc;  // zeroed with the counts
    o->win = (int32_t *)(b + a_win);
    o->partial = (float *)(b + a_par);
    o->partial_b = (float *)(b + a_pb);
    o->tie_list = (int32_t *)(b + a_tl);
    o->ldc = ldc;
    o->slices = slices;
    o->seg_per_slice = (num_segments + slices - 1) / slices;
    o->kin_p = kin_p;
  }
  (void)n_rows;
  return off + 256;
}
}  // namespace

extern "C" size_t pgnn_segmax_fc_bwd_workspace_bytes(int64_t n_rows,
                                                     int32_t n_cols,
                                                     int32_t num_segments,
                                                     int32_t k_in) {
  if (n_rows < 0 || n_cols <= 0 || num_segments < 0 || k_in <= 0) return 0;
  return segfc_carve(nullptr, n_rows, n_cols, num_segments, k_in, nullptr);
}

namespace {
// shared body: edges != null selects the edge-stage form (dP / dQ scatter
// instead of a materialised dX)
int segfc_bwd_impl(const float *Y, int64_t ld_y, const int32_t *seg_ids,
                   int64_t n_rows, int32_t n_cols, int32_t num_segments,
                   const float *out, int64_t ld_out, const float *grad_out,
                   int64_t ld_go, const float *X, int64_t ld_x, int32_t k_in,
                   const float *WT, int64_t ld_wt, float *dX, int64_t ld_dx,
                   int32_t dx_cols, int32_t mask_x, const int32_t *edges,
                   float *dP, float *dQ, int64_t ld_pq, float *dW, float *db,
                   void *workspace, size_t workspace_bytes, hipStream_t stream,
                   const float *P = nullptr, const float *Q = nullptr) {
  XSrc xs = {X, ld_x, P, Q, ld_pq, edges};
  PGNN_REQUIRE(n_rows >= 0 && n_cols > 0 && n_cols <= 512 && num_segments >= 0 &&
                   k_in > 0 && k_in <= 512,
               PGNN_E_INVALID, "segmax_fc_bwd: bad sizes");
  bool pair_pending = false;  // dP / dQ zeroed together with the counts below
  if (edges && num_segments > 0) {
    PGNN_REQUIRE(dP && dQ && ld_pq >= k_in, PGNN_E_INVALID,
                 "edge_segmax_fc_bwd: bad dP / dQ");
    if (n_rows == 0 || dQ != dP + (size_t)num_segments * ld_pq)
      PGNN_HIP(zero_pair(dP, dQ, (size_t)num_segments * ld_pq, stream));
    else
      pair_pending = true;
  }
  if (n_rows == 0 || num_segments == 0) return 0;
  PGNN_REQUIRE(Y && seg_ids && out && grad_out && WT && dW &&
                   (X || (P && Q && edges)),
               PGNN_E_INVALID, "segmax_fc_bwd: null pointer");
  PGNN_REQUIRE(ld_y % 4 == 0 && ld_out % 4 == 0 && (uintptr_t)Y % 16 == 0 &&
                   (uintptr_t)out % 16 == 0 && ld_y >= (n_cols + 3) / 4 * 4 &&
                   ld_out >= (n_cols + 3) / 4 * 4,
               PGNN_E_INVALID,
               "segmax_fc_bwd: Y / out rows must be 16-byte aligned and padded "
               "to a multiple of 4 columns");
  PGNN_REQUIRE((!X || ld_x >= k_in) && ld_wt >= k_in && ld_go >= n_cols,
               PGNN_E_INVALID, "segmax_fc_bwd: bad leading dimension");
  PGNN_REQUIRE(!dX || (dx_cols > 0 && dx_cols <= 512 && ld_dx >= dx_cols),
               PGNN_E_INVALID, "segmax_fc_bwd: bad dX shape");
  PGNN_REQUIRE(workspace && workspace_bytes >= pgnn_segmax_fc_bwd_workspace_bytes(
                                                   n_rows, n_cols, num_segments,
                                                   k_in),
               PGNN_E_WORKSPACE, "segmax_fc_bwd: workspace too small");
  SegFcWs w;
  segfc_carve(workspace, n_rows, n_cols, num_segments, k_in, &w);
  // (counts + the tie flag behind them; a whole number of 16-byte words -- the
  // region has 256 spare bytes)
  if (pair_pending) {
    // ... and the back-to-back dP | dQ in the SAME launch: a fill is a launch,
    // and the step is a chain of them
    const int64_t w0 = ((int64_t)num_segments * w.ldc * 4 + 16) / 16;
    const int64_t w1 = (int64_t)num_segments * ld_pq * 2 * 4 / 16;
    PGNN_REQUIRE(((uintptr_t)w.cnt % 16 == 0) && ((uintptr_t)dP % 16 == 0) &&
                     ((int64_t)num_segments * ld_pq * 8 % 16 == 0),
                 PGNN_E_INVALID, "edge_segmax_fc_bwd: dP must be 16-byte aligned");
    hipLaunchKernelGGL(zero2_kernel, dim3(grid_for(w0 + w1, 2048)), dim3(256), 0,
                       stream, (uint4 *)w.cnt, w0, (uint4 *)dP, w1);
  } else {
    PGNN_HIP(hipMemsetAsync(w.cnt, 0, (size_t)num_segments * w.ldc * 4 + 16,
                            stream));
  }
  const int cols4 = (n_cols + 3) / 4;
  hipLaunchKernelGGL(segmax_count_win4_kernel,
                     dim3(grid_for(n_rows * cols4, 8192)), dim3(256), 0, stream, Y,
                     ld_y, seg_ids, n_rows, cols4, num_segments, out, ld_out,
                     w.cnt, w.win, w.ldc, w.tie, w.tie_list, kTieCap);
  if (edges) {
    const int J = (n_cols + 63) / 64, I = (k_in + 63) / 64;
    const int64_t chunks = (n_rows + kScatterChunk - 1) / kScatterChunk;
    const unsigned blocks =
        (unsigned)((chunks + 3) / 4 < 8192 ? (chunks + 3) / 4 : 8192);
#define PGNN_SCAT(JV, IV)                                                      \
  hipLaunchKernelGGL((segmax_route_scatter_kernel<JV, IV>), dim3(blocks),      \
                     dim3(256), 0, stream, Y, ld_y, edges, n_rows, n_cols,     \
                     num_segments, out, ld_out, grad_out, ld_go, w.cnt, w.ldc,  \
                     WT, ld_wt, k_in, xs, dP, dQ, ld_pq)
    if (J == 5 && I == 5) PGNN_SCAT(5, 5);
    else if (J == 4 && I == 4) PGNN_SCAT(4, 4);
    else PGNN_SCAT(8, 8);
#undef PGNN_SCAT
  } else if (dX) {
    const int J = (n_cols + 63) / 64, I = (dx_cols + 63) / 64;
    const unsigned blocks = (unsigned)((n_rows + 3) / 4 < 8192 ? (n_rows + 3) / 4
                                                             : 8192);
#define PGNN_ROUTE(JV, IV)                                                     \
  hipLaunchKernelGGL((segmax_route_sparse_kernel<JV, IV>), dim3(blocks),       \
                     dim3(256), 0, stream, Y, ld_y, seg_ids, n_rows, n_cols,   \
                     num_segments, out, ld_out, grad_out, ld_go, w.cnt, w.ldc,  \
                     WT, ld_wt, k_in, X, ld_x, mask_x, dX, ld_dx, dx_cols)
    if (J == 5 && I == 5) PGNN_ROUTE(5, 5);
    else if (J == 5 && I == 2) PGNN_ROUTE(5, 2);
    else if (J == 4 && I == 4) PGNN_ROUTE(4, 4);
    else if (J == 8 && I == 4) PGNN_ROUTE(8, 4);
    else PGNN_ROUTE(8, 8);
#undef PGNN_ROUTE
  }
  {
    dim3 grid((unsigned)n_cols, (unsigned)w.slices);
    const int I = (w.kin_p + 63) / 64;
#define PGNN_WG(IV)                                                            \
  hipLaunchKernelGGL((segmax_wgrad_gather_kernel<IV>), grid, dim3(64), 0,      \
                     stream, grad_out, ld_go, w.cnt, w.win, w.ldc, num_segments, \
                     w.seg_per_slice, xs, k_in, w.kin_p, w.partial,             \
                     w.partial_b, n_cols)
    if (I <= 2) PGNN_WG(2);
    else if (I <= 4) PGNN_WG(4);
    else if (I == 5) PGNN_WG(5);
    else PGNN_WG(8);
#undef PGNN_WG
    hipLaunchKernelGGL(segmax_wgrad_reduce_kernel,
                       dim3(grid_for((int64_t)(k_in + 1) * n_cols)), dim3(256), 0,
                       stream, w.partial, w.partial_b, w.slices, n_cols, k_in,
                       w.kin_p, dW, db);
    hipLaunchKernelGGL(segmax_wgrad_ties_kernel,
                       dim3(grid_for(n_rows * n_cols, 2048)), dim3(256), 0,
                       stream, w.tie, w.tie_list, kTieCap, Y, ld_y, seg_ids,
                       n_rows, n_cols, num_segments, out, ld_out, grad_out, ld_go,
                       w.cnt, w.win, w.ldc, xs, k_in, dW, db);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_segmax_fc_bwd_f32(
    const float *Y, int64_t ld_y, const int32_t *seg_ids, int64_t n_rows,
    int32_t n_cols, int32_t num_segments, const float *out, int64_t ld_out,
    const float *grad_out, int64_t ld_go, const float *X, int64_t ld_x,
    int32_t k_in, const float *WT, int64_t ld_wt, float *dX, int64_t ld_dx,
    int32_t dx_cols, int32_t mask_x, float *dW, float *db, void *workspace,
    size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  return segfc_bwd_impl(Y, ld_y, seg_ids, n_rows, n_cols, num_segments, out,
                        ld_out, grad_out, ld_go, X, ld_x, k_in, WT, ld_wt, dX,
                        ld_dx, dx_cols, mask_x, nullptr, nullptr, nullptr, 0, dW,
                        db, workspace, workspace_bytes, (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_edge_segmax_fc_bwd_f32(
    const float *Y, int64_t ld_y, const int32_t *edges, const int32_t *dst_ids,
    int64_t n_edges, int32_t n_cols, int32_t num_vertices, const float *out,
    int64_t ld_out, const float *grad_out, int64_t ld_go, const float *H1,
    int64_t ld_h1, const float *P, const float *Q, int32_t k_in, const float *WT,
    int64_t ld_wt, float *dP, float *dQ, int64_t ld_pq, float *dW, float *db,
    void *workspace, size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(edges || n_edges == 0, PGNN_E_INVALID,
               "edge_segmax_fc_bwd: null edges");
  PGNN_REQUIRE(H1 || (P && Q) || n_edges == 0, PGNN_E_INVALID,
               "edge_segmax_fc_bwd: H1 rows or the (P, Q) tables are needed");
  // a non-null marker keeps the edge form selected for n_edges == 0 as well
  static const int32_t kNoEdges[2] = {0, 0};
  return segfc_bwd_impl(Y, ld_y, dst_ids, n_edges, n_cols, num_vertices, out,
                        ld_out, grad_out, ld_go, H1, ld_h1, k_in, WT, ld_wt,
                        nullptr, 0, 0, 1, edges ? edges : kNoEdges, dP, dQ, ld_pq,
                        dW, db, workspace, workspace_bytes, (hipStream_t)stream_,
                        H1 ? nullptr : P, H1 ? nullptr : Q);
  PGNN_GUARD_END
}

extern "C" int pgnn_edge_hidden_fwd(const float *P, const float *Q,
                                    int64_t ld_pq, const int32_t *edges,
                                    int64_t n_edges, float *H1, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(ld_pq > 0 && ld_pq % 4 == 0 && n_edges >= 0, PGNN_E_INVALID,
               "edge_hidden_fwd: ld must be a positive multiple of 4");
  if (n_edges == 0) return 0;
  PGNN_REQUIRE(P && Q && edges && H1, PGNN_E_INVALID,
               "edge_hidden_fwd: null pointer");
  hipLaunchKernelGGL(edge_hidden_fwd_kernel,
                     dim3(grid_for(n_edges * (ld_pq / 4), 8192)), dim3(256), 0,
                     (hipStream_t)stream_, P, Q, (int)(ld_pq / 4), edges,
                     n_edges, H1);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_edge_hidden_bwd(const float *dH1, int64_t ld,
                                    const int32_t *edges, int64_t n_edges,
                                    int64_t n_vertices, float *dP, float *dQ,
                                    void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(ld > 0 && n_edges >= 0 && n_vertices >= 0 && dP && dQ,
               PGNN_E_INVALID, "edge_hidden_bwd: bad argument");
  PGNN_HIP(zero_pair(dP, dQ, (size_t)n_vertices * ld, stream));
  if (n_edges == 0) return 0;
  PGNN_REQUIRE(dH1 && edges, PGNN_E_INVALID, "edge_hidden_bwd: null pointer");
  hipLaunchKernelGGL(edge_hidden_bwd_kernel, dim3(grid_for(n_edges * ld, 8192)),
                     dim3(256), 0, stream, dH1, (int)ld, edges, n_edges, dP, dQ);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_pool_features_fwd(const float *point_features,
                                      int32_t n_feat, const float *point_xyz,
                                      const int32_t *keypoint_indices,
                                      const int32_t *edges, int64_t n_edges,
                                      float *F, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_feat >= 0 && n_feat <= 13 && n_edges >= 0, PGNN_E_INVALID,
               "pool_features: bad sizes");
  if (n_edges == 0) return 0;
  PGNN_REQUIRE(point_xyz && keypoint_indices && edges && F, PGNN_E_INVALID,
               "pool_features: null pointer");
  // (thread per (edge, column): coalesced 64-byte rows; a thread per edge
  // writing its 16 floats ran at 0.7 TB/s)
  hipLaunchKernelGGL(pool_features_wide_kernel, dim3(grid_for(n_edges * 16, 8192)),
                     dim3(256), 0, (hipStream_t)stream_, point_features,
                     (int64_t)n_feat, n_feat, point_xyz, keypoint_indices, edges,
                     n_edges, F, 16);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_pool_features_wide_fwd(
    const float *point_features, int64_t ld_features, int32_t n_feat,
    const float *point_xyz, const int32_t *keypoint_indices,
    const int32_t *edges, int64_t n_edges, float *F, int64_t ld_f,
    void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(n_feat >= 0 && n_feat <= 4093 && ld_features >= n_feat &&
                   ld_f >= n_feat + 3 && ld_f <= 4096 && n_edges >= 0,
               PGNN_E_INVALID, "pool_features_wide: bad sizes");
  if (n_edges == 0) return 0;
  PGNN_REQUIRE((point_features || n_feat == 0) && point_xyz &&
                   keypoint_indices && edges && F,
               PGNN_E_INVALID, "pool_features_wide: null pointer");
  hipLaunchKernelGGL(pool_features_wide_kernel,
                     dim3(grid_for(n_edges * ld_f, 8192)), dim3(256), 0,
                     (hipStream_t)stream_, point_features, ld_features, n_feat,
                     point_xyz, keypoint_indices, edges, n_edges, F, (int)ld_f);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

// ---- batch_data (train.py:135-171): frames merged into one disjoint graph ------
// Every array of the batch is a concatenation of the frames' arrays; index
// arrays (keypoint indices, edge rows) move up by the points / centres of the
// frames before them.  One launch for the whole batch instead of a concatenate
// and an add per array and frame: job j copies n_words 4-byte words, adding
// add[w & 1] to int32 words (add = {0, 0}: plain copy).
namespace {
constexpr int kMergeJobs = 48;
struct MergeJobsDev {
  pgnn_merge_job j[kMergeJobs];
  int n;
};
__global__ void merge_rows_kernel(MergeJobsDev js) {
  const pgnn_merge_job &job = js.j[blockIdx.y];
  const int32_t *src = (const int32_t *)job.src;
  int32_t *dst = (int32_t *)job.dst;
  const int32_t a0 = job.add0, a1 = job.add1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < job.n_words; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i] + ((i & 1) ? a1 : a0);
}
}  // namespace

extern "C" int pgnn_merge_rows(const pgnn_merge_job *jobs_host, int32_t n_jobs,
                               void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs_host), PGNN_E_INVALID,
               "merge_rows: bad argument");
  for (int first = 0; first < n_jobs; first += kMergeJobs) {
    MergeJobsDev js;
    js.n = 0;
    int64_t longest = 0;
    for (int i = first; i < n_jobs && js.n < kMergeJobs; ++i) {
      const pgnn_merge_job &j = jobs_host[i];
      PGNN_REQUIRE(j.n_words >= 0 && (j.n_words == 0 || (j.src && j.dst)),
                   PGNN_E_INVALID, "merge_rows: bad job");
      if (j.n_words == 0) continue;
      js.j[js.n++] = j;
      if (j.n_words > longest) longest = j.n_words;
    }
    if (js.n == 0) continue;
    hipLaunchKernelGGL(merge_rows_kernel,
                       dim3(grid_for(longest, 256), (unsigned)js.n), dim3(256), 0,
                       stream, js);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_relu_mask_mul(float *dY, const float *Y, int64_t count,
                                  void *stream_) {
  PGNN_GUARD_BEGIN
  if (count <= 0) return 0;
  PGNN_REQUIRE(dY && Y, PGNN_E_INVALID, "relu_mask_mul: null pointer");
  hipLaunchKernelGGL(relu_mask_mul_kernel, dim3(grid_for(count, 8192)),
                     dim3(256), 0, (hipStream_t)stream_, dY, Y, count);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_scatter_max_bwd_f32(
    const float *data, int64_t ld_data, const int32_t *seg_ids, int64_t n_rows,
    int32_t n_cols, int32_t num_segments, const float *out, int64_t ld_out,
    const float *grad_out, int64_t ld_grad_out, int32_t *tie_count_ws,
    float *grad_data, int64_t ld_grad_data, int32_t relu_mask, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && n_cols > 0 && num_segments >= 0, PGNN_E_INVALID,
               "scatter_max_bwd: bad sizes");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(data && seg_ids && out && grad_out && tie_count_ws && grad_data,
               PGNN_E_INVALID, "scatter_max_bwd: null pointer");
  PGNN_HIP(hipMemsetAsync(tie_count_ws, 0, (size_t)num_segments * n_cols * 4,
                          stream));
  const bool vec4 =
      n_cols % 4 == 0 && ld_data % 4 == 0 && ld_out % 4 == 0 &&
      ld_grad_data % 4 == 0 && (uintptr_t)data % 16 == 0 &&
      (uintptr_t)out % 16 == 0 && (uintptr_t)grad_data % 16 == 0;
  if (vec4) {
    const unsigned g4 = grid_for(n_rows * (n_cols / 4), 8192);
    hipLaunchKernelGGL(segmax_count4_kernel, dim3(g4), dim3(256), 0, stream,
                       data, ld_data, seg_ids, n_rows, n_cols / 4, num_segments,
                       out, ld_out, tie_count_ws, relu_mask);
    hipLaunchKernelGGL(segmax_route4_kernel, dim3(g4), dim3(256), 0, stream,
                       data, ld_data, seg_ids, n_rows, n_cols / 4, num_segments,
                       out, ld_out, grad_out, ld_grad_out, tie_count_ws,
                       grad_data, ld_grad_data, relu_mask);
    PGNN_HIP(hipGetLastError());
    return 0;
  }
  const unsigned g = grid_for(n_rows * n_cols, 8192);
  hipLaunchKernelGGL(segmax_count_kernel, dim3(g), dim3(256), 0, stream, data,
                     ld_data, seg_ids, n_rows, n_cols, num_segments, out, ld_out,
                     tie_count_ws, relu_mask);
  hipLaunchKernelGGL(segmax_route_kernel, dim3(g), dim3(256), 0, stream, data,
                     ld_data, seg_ids, n_rows, n_cols, num_segments, out, ld_out,
                     grad_out, ld_grad_out, tie_count_ws, grad_data,
                     ld_grad_data, relu_mask);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

namespace pgnn {
// workgroups (in-blocks x row slices) one weight-gradient launch aims for
// (768 = 256 CUs x 3 resident workgroups of 48 KB LDS: exactly one wave of
// workgroups; 1024 left a quarter-filled second wave and cost 6 % of the step)
int g_wgrad_wg_target = 512;
}  // namespace pgnn

namespace {
int wg_slices(int64_t rows, int k_in) {
  // one full wave of workgroups: (in-blocks x slices) ~ g_wgrad_wg_target
  const int64_t in_blocks = ((int64_t)k_in + 1 + 63) / 64;
  int64_t s = (rows + kWgRows - 1) / kWgRows;
  int64_t cap = g_wgrad_wg_target / in_blocks;
  if (cap < 1) cap = 1;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}
}  // namespace

extern "C" size_t pgnn_weight_grad_workspace_bytes(int32_t k_in, int32_t n_out,
                                                   int64_t n_rows) {
  if (k_in <= 0 || n_out <= 0 || n_rows < 0) return 0;
  const size_t in_blocks = ((size_t)k_in + 1 + 63) / 64;
  const size_t nt = ((size_t)n_out + 15) / 16;
  // (+ one [slices][16 nt] block for the separately summed bias, k_in % 64 == 0)
  return (size_t)wg_slices(n_rows, k_in) * (in_blocks * 64 + 1) * nt * 16 * 4 +
         256;
}

extern "C" int pgnn_weight_grad_f32(const float *X, int64_t ld_x, int32_t k_in,
                                    const float *dZ, int64_t ld_dz,
                                    int32_t n_out, int64_t n_rows, float *dW,
                                    float *db, int32_t accumulate,
                                    void *workspace, size_t workspace_bytes,
                                    void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(k_in > 0 && n_out > 0 && n_rows >= 0 && dW, PGNN_E_INVALID,
               "weight_grad: bad argument");
  PGNN_REQUIRE(workspace && workspace_bytes >= pgnn_weight_grad_workspace_bytes(
                                                   k_in, n_out, n_rows),
               PGNN_E_WORKSPACE, "weight_grad: workspace too small");
  PGNN_REQUIRE(n_rows == 0 || (X && dZ && ld_x >= k_in && ld_dz >= n_out),
               PGNN_E_INVALID, "weight_grad: bad input");
  // k_in a multiple of 64: the bias "input column" would be a block of its
  // own that re-reads all of dZ for one row of dW; block 0 sums db from the
  // tile it has staged anyway
  const bool split_bias = db != nullptr && k_in % 64 == 0;
  const int in_blocks = split_bias ? k_in / 64 : (k_in + 1 + 63) / 64;
  const int slices = wg_slices(n_rows, k_in);
  int64_t rps = (n_rows + slices - 1) / slices;
  rps = (rps + kWgRows - 1) / kWgRows * kWgRows;
  if (rps < kWgRows) rps = kWgRows;
  float *partial = (float *)workspace;
  // output columns in passes of <= 320 (4 waves x <= 5 column tiles), e.g. the
  // 512-wide pooling layer of ped_cyl; the partial buffer is reused in stream
  // order
  for (int c0 = 0; c0 < n_out; c0 += 320) {
    const int nc = n_out - c0 < 320 ? n_out - c0 : 320;
    const int nt = (nc + 15) / 16;
    const size_t lds = (size_t)kWgRows * (68 + 16 * nt + 4) * 4;
    const int ntw = (nt + 3) / 4;
    dim3 grid(in_blocks, slices);
    // the pass's bias partials sit right behind its main block
    float *bias_partial =
        split_bias ? partial + (size_t)slices * in_blocks * 64 * nt * 16 : nullptr;
#define PGNN_WG(NTV)                                                          \
  hipLaunchKernelGGL((weight_grad_kernel<NTV>), grid, dim3(256), lds, stream,  \
                     X, ld_x, k_in, dZ + c0, ld_dz, nc, n_rows, rps, nt,       \
                     partial, bias_partial)
    switch (ntw) {
      case 1: PGNN_WG(1); break;
      case 2: PGNN_WG(2); break;
      case 3: PGNN_WG(3); break;
      case 4: PGNN_WG(4); break;
      default: PGNN_WG(5); break;
    }
#undef PGNN_WG
    hipLaunchKernelGGL(weight_grad_reduce_kernel<4>,
                       dim3(grid_for((int64_t)(k_in + 1) * nc * 4)), dim3(256), 0,
                       stream, partial, slices, (int64_t)in_blocks * 64,
                       nt * 16, k_in, nc, (int64_t)n_out, dW + c0,
                       db ? db + c0 : nullptr, accumulate,
                       (const float *)bias_partial);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

// ---- fused backward of PointSetPooling's narrow point-MLP layers ---------------
// car chain: feat [E,16 (4 used)] -> 32 -> 64 -> 128 (-> 300, whose adjoint is the
// sparse segment-max kernel above and hands over dZ2, the gradient w.r.t. the
// third layer's pre-activation).  Layer by layer this was three E-row weight
// gradients, two E-row dX GEMMs and two mask passes -- 0.4 ms of a 4 ms step,
// every one re-reading [E, 32..128] arrays and writing its dX back (gnn.py:256-277
// under tf.gradients).  Here a workgroup walks a slice of the rows in 32-row
// tiles: the four tiles (feat, act0, act1, dZ2) are staged once, and
//     dW2 += act1^T dZ2        dA1 = dZ2 W2^T    dZ1 = dA1 * (act1 > 0)
//     dW1 += act0^T dZ1        dA0 = dZ1 W1^T    dZ0 = dA0 * (act0 > 0)
//     dW0 += feat^T dZ0        db_l += column sums of dZ_l
// run from LDS: the dX products through the forward engine's layer pass on the
// transposed weight images (what fc_dx does per layer), the weight gradients
// with the MFMA loop of weight_grad_body, accumulated in registers over the
// slice.  Partials per slice, summed in a fixed order by weight_grad_reduce_kernel.
constexpr int kPnRows = 32;
struct PoolNarrowArgs {
  const float *feat, *act0, *act1, *dz2;
  int64_t rows, rps;
  LayerDev t2, t1;  // transposed images: 128 -> 64, 64 -> 32 (zero bias)
  float *pw2, *pb2, *pw1, *pb1, *pw0, *pb0;  // partials [slices][...]
};

// tile[r][c] <- src[(r0 + r) * ld + c] for 32 rows x W columns (W = 16..128),
// zero for rows past r_end; unconditional clamped loads.  In two halves --
// global -> registers (pn_load), registers -> LDS (pn_store) -- so that the
// NEXT tile's rows are in flight while this tile is computed: the kernel was a
// chain of (HBM round trip, barrier, ~2 us of MFMAs) per 32 rows and ran at
// 1.75 TB/s / 40 TFLOP/s, bound by neither.
#ifndef PGNN_PN_PREFETCH
#define PGNN_PN_PREFETCH 1
#endif
template <int W>
struct PnRegs {
  static constexpr int TPR = W < 64 ? W : 64;  // threads per row
  static constexpr int RPS = 256 / TPR;        // rows per sweep
  static constexpr int N = (W / TPR) * (kPnRows / RPS);
  float v[N];
};
template <int W>
__device__ __forceinline__ void pn_load(PnRegs<W> &g, const float *src, int64_t r0,
                                        int64_t r_end) {
  constexpr int TPR = PnRegs<W>::TPR, RPS = PnRegs<W>::RPS;
  const int c0 = threadIdx.x % TPR, rs = threadIdx.x / TPR;
#pragma unroll
  for (int cb = 0; cb < W; cb += TPR)
#pragma unroll
    for (int i = 0; i < kPnRows / RPS; ++i) {
      int64_t row = r0 + rs + RPS * i;
      row = row < r_end ? row : r_end - 1;
      row = row > 0 ? row : 0;
      g.v[(cb / TPR) * (kPnRows / RPS) + i] = src[row * W + cb + c0];
    }
}
template <int W>
__device__ __forceinline__ void pn_store(float *tile, int ldt, const PnRegs<W> &g,
                                         int64_t r0, int64_t r_end) {
  constexpr int TPR = PnRegs<W>::TPR, RPS = PnRegs<W>::RPS;
  const int c0 = threadIdx.x % TPR, rs = threadIdx.x / TPR;
#pragma unroll
  for (int cb = 0; cb < W; cb += TPR)
#pragma unroll
    for (int i = 0; i < kPnRows / RPS; ++i) {
      const int r = rs + RPS * i;
      tile[r * ldt + cb + c0] =
          (r0 + r < r_end) ? g.v[(cb / TPR) * (kPnRows / RPS) + i] : 0.0f;
    }
}

// acc[m][j] += X^T Z over the tile's 32 rows: X [32][ldx] (MT input tiles of
// 16), Z [32][ldz], column tiles t_j = wave + 4 j (clamped to nt - 1)
template <int MT, int NT>
__device__ __forceinline__ void pn_wgrad(const float *Xs, int ldx, const float *Zs,
                                         int ldz, int nt, int wave, int lane,
                                         v4f (&acc)[MT][NT]) {
  int toff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int t = wave + 4 * j;
    if (t > nt - 1) t = nt - 1;
    toff[j] = 16 * t + (lane & 15);
  }
#pragma unroll
  for (int q = 0; q < kPnRows / 16; ++q) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 16 * q + 4 * (lane >> 4) + s;
      float a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = Xs[r * ldx + 16 * m + (lane & 15)];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = Zs[r * ldz + toff[j]];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[j], acc[m][j],
                                                           0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void pool_narrow_bwd_kernel(PoolNarrowArgs a) {
  constexpr int LZ2 = 128 + 8, LA1 = 64 + 8, LA0 = 32 + 8, LF = 16 + 8;
  __shared__ __attribute__((aligned(16))) float Z2[kPnRows * LZ2];
  __shared__ __attribute__((aligned(16))) float A1[kPnRows * LA1];
  __shared__ __attribute__((aligned(16))) float A0[kPnRows * LA0];
  __shared__ __attribute__((aligned(16))) float F[kPnRows * LF];
  __shared__ __attribute__((aligned(16))) float D1[kPnRows * LA1];
  __shared__ __attribute__((aligned(16))) float D0[kPnRows * LA0];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = blockIdx.x;
  const int64_t r_begin = (int64_t)slice * a.rps;
  int64_t r_end = r_begin + a.rps;
  if (r_end > a.rows) r_end = a.rows;
  v4f w2[4][2], w1[2][1], w0[1][1];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) w2[m][j] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 2; ++m) w1[m][0] = (v4f){0.f, 0.f, 0.f, 0.f};
  w0[0][0] = (v4f){0.f, 0.f, 0.f, 0.f};
  float b2 = 0.0f, b1 = 0.0f, b0 = 0.0f;  // thread t: column t of db2 / db1 / db0
  PnRegs<16> gf;
  PnRegs<32> g0;
  PnRegs<64> g1;
  PnRegs<128> gz;
  if (PGNN_PN_PREFETCH && r_begin < r_end) {
    pn_load<16>(gf, a.feat, r_begin, r_end);
    pn_load<32>(g0, a.act0, r_begin, r_end);
    pn_load<64>(g1, a.act1, r_begin, r_end);
    pn_load<128>(gz, a.dz2, r_begin, r_end);
  }
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kPnRows) {
    if (!PGNN_PN_PREFETCH) {  // (A/B builds: the rows when they are needed)
      pn_load<16>(gf, a.feat, r0, r_end);
      pn_load<32>(g0, a.act0, r0, r_end);
      pn_load<64>(g1, a.act1, r0, r_end);
      pn_load<128>(gz, a.dz2, r0, r_end);
    }
    __syncthreads();  // the previous tile's buffers are free
    pn_store<16>(F, LF, gf, r0, r_end);
    pn_store<32>(A0, LA0, g0, r0, r_end);
    pn_store<64>(A1, LA1, g1, r0, r_end);
    pn_store<128>(Z2, LZ2, gz, r0, r_end);
    __syncthreads();
    if (PGNN_PN_PREFETCH && r0 + kPnRows < r_end) {  // the next tile's rows, under this tile's work
      pn_load<16>(gf, a.feat, r0 + kPnRows, r_end);
      pn_load<32>(g0, a.act0, r0 + kPnRows, r_end);
      pn_load<64>(g1, a.act1, r0 + kPnRows, r_end);
      pn_load<128>(gz, a.dz2, r0 + kPnRows, r_end);
    }
    // layer 2 (64 -> 128)
    pn_wgrad<4, 2>(A1, LA1, Z2, LZ2, 8, wave, lane, w2);
    if (threadIdx.x < 128) {
      float sum = 0.0f;
#pragma unroll 8
      for (int r = 0; r < kPnRows; ++r) sum += Z2[r * LZ2 + threadIdx.x];
      b2 += sum;
    }
    // (64 and 32 output columns: one column tile per wave -- the fixed form,
    // not layer_pass_dispatch's run-time switch, whose widest case sets the
    // kernel's register count)
    layer_pass<2, 1, false, 4>(Z2, LZ2, D1, LA1, a.t2, 0, wave, lane, false);
    for (int idx = threadIdx.x; idx < kPnRows * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      if (!(A1[r * LA1 + c] > 0.0f)) D1[r * LA1 + c] = 0.0f;
    }
    __syncthreads();
    // layer 1 (32 -> 64)
    pn_wgrad<2, 1>(A0, LA0, D1, LA1, 4, wave, lane, w1);
    if (threadIdx.x < 64) {
      float sum = 0.0f;
#pragma unroll 8
      for (int r = 0; r < kPnRows; ++r) sum += D1[r * LA1 + threadIdx.x];
      b1 += sum;
    }
    layer_pass<2, 1, false, 4>(D1, LA1, D0, LA0, a.t1, 0, wave, lane, false);
    for (int idx = threadIdx.x; idx < kPnRows * 32; idx += 256) {
      const int r = idx >> 5, c = idx & 31;
      if (!(A0[r * LA0 + c] > 0.0f)) D0[r * LA0 + c] = 0.0f;
    }
    __syncthreads();
    // layer 0 (16 -> 32)
    pn_wgrad<1, 1>(F, LF, D0, LA0, 2, wave, lane, w0);
    if (threadIdx.x < 32) {
      float sum = 0.0f;
#pragma unroll 8
      for (int r = 0; r < kPnRows; ++r) sum += D0[r * LA0 + threadIdx.x];
      b0 += sum;
    }
  }
  // partials: register rr of tile (m, t) <-> dW[16 m + 4 (lane >> 4) + rr][16 t +
  // (lane & 15)]
  {
    float *p = a.pw2 + (int64_t)slice * 64 * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = wave + 4 * j;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          p[(16 * m + 4 * (lane >> 4) + rr) * 128 + 16 * t + (lane & 15)] =
              w2[m][j][rr];
    }
    if (threadIdx.x < 128) a.pb2[(int64_t)slice * 128 + threadIdx.x] = b2;
  }
  {
    float *p = a.pw1 + (int64_t)slice * 32 * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        p[(16 * m + 4 * (lane >> 4) + rr) * 64 + 16 * wave + (lane & 15)] =
            w1[m][0][rr];
    if (threadIdx.x < 64) a.pb1[(int64_t)slice * 64 + threadIdx.x] = b1;
  }
  {
    float *p = a.pw0 + (int64_t)slice * 16 * 32;
    if (wave < 2) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        p[(4 * (lane >> 4) + rr) * 32 + 16 * wave + (lane & 15)] = w0[0][0][rr];
    }
    if (threadIdx.x < 32) a.pb0[(int64_t)slice * 32 + threadIdx.x] = b0;
  }
}

namespace {
// rows per slice / slices of one job when `wgs_per_in_block` workgroups are
// available per input block
void wg_many_split(int64_t rows, int target_slices, int64_t &rps, int &slices) {
  int64_t s = target_slices < 1 ? 1 : target_slices;
  const int64_t max_s = (rows + kWgRows - 1) / kWgRows;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  rps = (rows + s - 1) / s;
  rps = (rps + kWgRows - 1) / kWgRows * kWgRows;
  if (rps < kWgRows) rps = kWgRows;
  slices = (int)((rows + rps - 1) / rps);
  if (slices < 1) slices = 1;
}
int wg_many_target_slices(const pgnn_wgrad_job *jobs, int32_t n_jobs) {
  int64_t in_blocks = 0;
  for (int i = 0; i < n_jobs; ++i)
    in_blocks += ((int64_t)jobs[i].k_in + 1 + 63) / 64;
  if (in_blocks < 1) in_blocks = 1;
  // one full wave of workgroups over the whole batch of jobs: TWO per CU --
  // 184 registers a wave (104 + 80 accumulators) allow two 4-wave workgroups
  // per CU, whatever the 48 KB of LDS would admit: with three per CU a third of
  // the grid ran as a second round (`wgrad_wg_target`, default 2 x 256)
  int t = (int)((int64_t)g_wgrad_wg_target * device_cu_count() / 256 / in_blocks);
  return t < 1 ? 1 : t;
}
}  // namespace

extern "C" size_t pgnn_weight_grad_many_workspace_bytes(
    const pgnn_wgrad_job *jobs, int32_t n_jobs) {
  if (!jobs || n_jobs <= 0) return 0;
  const int target = wg_many_target_slices(jobs, n_jobs);
  size_t floats = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const pgnn_wgrad_job &j = jobs[i];
    if (j.k_in <= 0 || j.n_out <= 0 || j.n_rows <= 0) continue;
    int64_t rps;
    int slices;
    wg_many_split(j.n_rows, target, rps, slices);
    const size_t in_blocks = ((size_t)j.k_in + 1 + 63) / 64;
    const size_t nt = ((size_t)j.n_out + 15) / 16;
    floats += (size_t)slices * in_blocks * 64 * nt * 16;
  }
  return floats * 4 + 256;
}

extern "C" int pgnn_weight_grad_many_f32(const pgnn_wgrad_job *jobs,
                                         int32_t n_jobs, void *workspace,
                                         size_t workspace_bytes,
                                         void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs), PGNN_E_INVALID,
               "weight_grad_many: bad argument");
  if (n_jobs == 0) return 0;
  PGNN_REQUIRE(workspace && workspace_bytes >=
                                pgnn_weight_grad_many_workspace_bytes(jobs, n_jobs),
               PGNN_E_WORKSPACE, "weight_grad_many: workspace too small");
  for (int i = 0; i < n_jobs; ++i) {
    const pgnn_wgrad_job &j = jobs[i];
    PGNN_REQUIRE(j.k_in > 0 && j.n_out > 0 && j.n_out <= 320 && j.n_rows >= 0 &&
                     j.dW,
                 PGNN_E_INVALID,
                 "weight_grad_many: bad job (1 <= n_out <= 320, dW required)");
    PGNN_REQUIRE(j.n_rows == 0 ||
                     (j.X && j.dZ && j.ld_x >= j.k_in && j.ld_dz >= j.n_out),
                 PGNN_E_INVALID, "weight_grad_many: bad job input");
  }
  const int target = wg_many_target_slices(jobs, n_jobs);
  float *partial = (float *)workspace;
  int64_t part_off = 0;
  const size_t lds = (size_t)kWgRows * (68 + 16 * 20 + 4) * 4;
  {
    const int lrc = ensure_dynamic_lds(
        reinterpret_cast<const void *>(weight_grad_many_kernel), lds);
    if (lrc) return lrc;
  }
  for (int first = 0; first < n_jobs; first += kWgManyJobs) {
    WgJobsDev js = {};
    int wg = 0;
    int64_t out0 = 0;
    for (int i = first; i < n_jobs && js.n < kWgManyJobs; ++i) {
      const pgnn_wgrad_job &j = jobs[i];
      WgJobDev &d = js.j[js.n];
      if (j.n_rows <= 0) {
        // no rows: dW / db stay (accumulate) or become zero
        if (!j.accumulate) {
          PGNN_HIP(hipMemsetAsync(j.dW, 0, (size_t)j.k_in * j.n_out * 4, stream));
          if (j.db) PGNN_HIP(hipMemsetAsync(j.db, 0, (size_t)j.n_out * 4, stream));
        }
        continue;
      }
      d.X = j.X;
      d.dZ = j.dZ;
      d.dW = j.dW;
      d.db = j.db;
      d.ldx = j.ld_x;
      d.ldz = j.ld_dz;
      d.rows = j.n_rows;
      d.k_in = j.k_in;
      d.n_out = j.n_out;
      d.nt = (j.n_out + 15) / 16;
      d.in_blocks = (j.k_in + 1 + 63) / 64;
      wg_many_split(j.n_rows, target, d.rps, d.slices);
      d.wg0 = wg;
      d.accumulate = j.accumulate ? 1 : 0;
      d.part_off = part_off;
      d.out0 = out0;
      wg += d.in_blocks * d.slices;
      out0 += (int64_t)(j.k_in + 1) * j.n_out;
      part_off += (int64_t)d.slices * d.in_blocks * 64 * d.nt * 16;
      ++js.n;
    }
    if (js.n == 0) continue;
    js.total_wgs = wg;
    js.total_out = out0;
    hipLaunchKernelGGL(weight_grad_many_kernel, dim3((unsigned)wg), dim3(256),
                       lds, stream, js, partial);
    hipLaunchKernelGGL(weight_grad_reduce_many_kernel,
                       dim3(grid_for(out0 * 4)), dim3(256), 0, stream, js,
                       (const float *)partial);
  }
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

namespace {
void pn_split(int64_t rows, int64_t &rps, int &slices) {
  // one wave of workgroups (3 per CU at 48 KB of LDS; tunable with the
  // weight-gradient kernels' `wgrad_wg_target`)
  int64_t s = g_wgrad_wg_target;
  const int64_t max_s = (rows + kPnRows - 1) / kPnRows;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  rps = (rows + s - 1) / s;
  rps = (rps + kPnRows - 1) / kPnRows * kPnRows;
  slices = (int)((rows + rps - 1) / rps);
  if (slices < 1) slices = 1;
}
constexpr size_t kPnSliceFloats = 64 * 128 + 128 + 32 * 64 + 64 + 16 * 32 + 32;
}  // namespace

extern "C" size_t pgnn_pool_narrow_bwd_workspace_bytes(int64_t n_rows) {
  if (n_rows < 0) return 0;
  int64_t rps;
  int slices;
  pn_split(n_rows > 0 ? n_rows : 1, rps, slices);
  return (size_t)slices * kPnSliceFloats * 4 + 1024;
}

extern "C" int pgnn_pool_narrow_bwd_f32(
    const float *feat, const float *act0, const float *act1, const float *dz2,
    int64_t n_rows, const float *w2t_packed, const float *w1t_packed,
    int32_t k_in0, float *dW0, float *db0, float *dW1, float *db1, float *dW2,
    float *db2, int32_t accumulate, void *workspace, size_t workspace_bytes,
    void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && k_in0 >= 1 && k_in0 <= 15 && dW0 && dW1 && dW2 &&
                   db0 && db1 && db2 && w2t_packed && w1t_packed,
               PGNN_E_INVALID, "pool_narrow_bwd: bad argument");
  if (n_rows == 0) {
    if (!accumulate) {
      PGNN_HIP(hipMemsetAsync(dW0, 0, (size_t)k_in0 * 32 * 4, stream));
      PGNN_HIP(hipMemsetAsync(dW1, 0, (size_t)32 * 64 * 4, stream));
      PGNN_HIP(hipMemsetAsync(dW2, 0, (size_t)64 * 128 * 4, stream));
      PGNN_HIP(hipMemsetAsync(db0, 0, 32 * 4, stream));
      PGNN_HIP(hipMemsetAsync(db1, 0, 64 * 4, stream));
      PGNN_HIP(hipMemsetAsync(db2, 0, 128 * 4, stream));
    }
    return 0;
  }
  PGNN_REQUIRE(feat && act0 && act1 && dz2, PGNN_E_INVALID,
               "pool_narrow_bwd: null input");
  PGNN_REQUIRE(workspace &&
                   workspace_bytes >= pgnn_pool_narrow_bwd_workspace_bytes(n_rows),
               PGNN_E_WORKSPACE, "pool_narrow_bwd: workspace too small");
  PoolNarrowArgs a;
  a.feat = feat;
  a.act0 = act0;
  a.act1 = act1;
  a.dz2 = dz2;
  a.rows = n_rows;
  int slices;
  pn_split(n_rows, a.rps, slices);
  a.t2.wp = w2t_packed;
  a.t2.kq = 8;
  a.t2.nt = 4;
  a.t2.relu_from = 64;  // linear
  a.t1.wp = w1t_packed;
  a.t1.kq = 4;
  a.t1.nt = 2;
  a.t1.relu_from = 32;
  float *w = (float *)workspace;
  a.pw2 = w;
  w += (size_t)slices * 64 * 128;
  a.pb2 = w;
  w += (size_t)slices * 128;
  a.pw1 = w;
  w += (size_t)slices * 32 * 64;
  a.pb1 = w;
  w += (size_t)slices * 64;
  a.pw0 = w;
  w += (size_t)slices * 16 * 32;
  a.pb0 = w;
  hipLaunchKernelGGL(pool_narrow_bwd_kernel, dim3((unsigned)slices), dim3(256), 0,
                     stream, a);
  const int acc = accumulate ? 1 : 0;
  // (grid_for counts 256-thread blocks of work items: x 4 = one workgroup per
  // 64 outputs)
  hipLaunchKernelGGL(weight_grad_reduce_kernel<16>,
                     dim3(grid_for((int64_t)65 * 128 * 4)), dim3(1024), 0, stream,
                     (const float *)a.pw2, slices,
                     (int64_t)64, 128, 64, 128, (int64_t)128, dW2, db2, acc,
                     (const float *)a.pb2);
  hipLaunchKernelGGL(weight_grad_reduce_kernel<16>,
                     dim3(grid_for((int64_t)33 * 64 * 4)), dim3(1024), 0, stream,
                     (const float *)a.pw1, slices,
                     (int64_t)32, 64, 32, 64, (int64_t)64, dW1, db1, acc,
                     (const float *)a.pb1);
  hipLaunchKernelGGL(weight_grad_reduce_kernel<16>,
                     dim3(grid_for((int64_t)(k_in0 + 1) * 32 * 4)), dim3(1024), 0,
                     stream, (const float *)a.pw0, slices, (int64_t)16, 32,
                     (int)k_in0, 32, (int64_t)32, dW0, db0, acc,
                     (const float *)a.pb0);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_loss_fwd_bwd(const float *logits, int64_t ld_logits,
                                 const int32_t *labels, const float *pred_box,
                                 int32_t box_len, const float *gt_box,
                                 const float *valid, int64_t n_vertices,
                                 int32_t num_classes, float cls_grad_scale,
                                 float loc_grad_scale, double *sums4,
                                 float *dlogits, float *dpred_box,
                                 void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices >= 0 && num_classes > 0 && box_len > 0 && sums4,
               PGNN_E_INVALID, "loss: bad argument");
  PGNN_HIP(hipMemsetAsync(sums4, 0, 4 * sizeof(double), stream));
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(logits && labels && pred_box && gt_box && valid, PGNN_E_INVALID,
               "loss: null pointer");
  hipLaunchKernelGGL(loss_kernel, dim3(grid_for(n_vertices, 1024)), dim3(256), 0,
                     stream, logits, ld_logits, labels, pred_box, box_len, gt_box,
                     valid, n_vertices, num_classes, cls_grad_scale,
                     loc_grad_scale, sums4, dlogits, dpred_box,
                     (const double *)nullptr, 0.0, 0.0, 0, 0.0f, 0.0f,
                     (const float *)nullptr, (const float *)nullptr,
                     (const float *)nullptr, 1.0f, (float *)nullptr,
                     (float *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_loss_fwd_bwd_counts(
    const float *logits, int64_t ld_logits, const int32_t *labels,
    const float *pred_box, int32_t box_len, const float *gt_box,
    const float *valid, int64_t n_vertices, int32_t num_classes,
    double cls_loss_weight, double loc_loss_weight, const double *counts2,
    double *sums4, float *dlogits, float *dpred_box, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices >= 0 && num_classes > 0 && box_len > 0 && sums4 &&
                   counts2,
               PGNN_E_INVALID, "loss_counts: bad argument");
  PGNN_HIP(hipMemsetAsync(sums4, 0, 4 * sizeof(double), stream));
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(logits && labels && pred_box && gt_box && valid, PGNN_E_INVALID,
               "loss_counts: null pointer");
  hipLaunchKernelGGL(loss_kernel, dim3(grid_for(n_vertices, 1024)), dim3(256), 0,
                     stream, logits, ld_logits, labels, pred_box, box_len, gt_box,
                     valid, n_vertices, num_classes, 0.0f, 0.0f, sums4, dlogits,
                     dpred_box, counts2, cls_loss_weight, loc_loss_weight, 0, 0.0f,
                     0.0f, (const float *)nullptr, (const float *)nullptr,
                     (const float *)nullptr, 1.0f, (float *)nullptr,
                     (float *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_loss_fwd_bwd_ex(
    const float *logits, int64_t ld_logits, const int32_t *labels,
    const float *pred_box, int32_t box_len, const float *gt_box,
    const float *valid, int64_t n_vertices, int32_t num_classes,
    float cls_grad_scale, float loc_grad_scale, const double *counts2,
    double cls_loss_weight, double loc_loss_weight, int32_t cls_kind,
    float alpha, float gamma, const float *class_loc_weight, double *sums4,
    float *dlogits, float *dpred_box, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices >= 0 && num_classes > 0 && box_len > 0 && sums4 &&
                   cls_kind >= 0 && cls_kind <= 2 && gamma >= 0.0f,
               PGNN_E_INVALID, "loss_ex: bad argument");
  PGNN_HIP(hipMemsetAsync(sums4, 0, 4 * sizeof(double), stream));
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(logits && labels && pred_box && gt_box && valid, PGNN_E_INVALID,
               "loss_ex: null pointer");
  hipLaunchKernelGGL(loss_kernel, dim3(grid_for(n_vertices, 1024)), dim3(256), 0,
                     stream, logits, ld_logits, labels, pred_box, box_len, gt_box,
                     valid, n_vertices, num_classes, cls_grad_scale,
                     loc_grad_scale, sums4, dlogits, dpred_box, counts2,
                     cls_loss_weight, loc_loss_weight, cls_kind, alpha, gamma,
                     class_loc_weight, (const float *)nullptr,
                     (const float *)nullptr, 1.0f, (float *)nullptr,
                     (float *)nullptr);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_loss_fwd_bwd_sel(
    const float *logits, int64_t ld_logits, const int32_t *labels,
    const float *pred_box, int32_t box_len, const float *gt_box,
    const float *valid, int64_t n_vertices, int32_t num_classes,
    float cls_grad_scale, float loc_grad_scale, const double *counts2,
    double cls_loss_weight, double loc_loss_weight, int32_t cls_kind,
    float alpha, float gamma, const float *class_loc_weight,
    const float *select_cls, const float *select_loc, float cls_sum_weight,
    float *point_cls, float *point_loc, double *sums4, float *dlogits,
    float *dpred_box, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_vertices >= 0 && num_classes > 0 && box_len > 0 && sums4 &&
                   cls_kind >= 0 && cls_kind <= 2 && gamma >= 0.0f,
               PGNN_E_INVALID, "loss_sel: bad argument");
  PGNN_HIP(hipMemsetAsync(sums4, 0, 4 * sizeof(double), stream));
  if (n_vertices == 0) return 0;
  PGNN_REQUIRE(logits && labels && pred_box && gt_box && valid, PGNN_E_INVALID,
               "loss_sel: null pointer");
  hipLaunchKernelGGL(loss_kernel, dim3(grid_for(n_vertices, 1024)), dim3(256), 0,
                     stream, logits, ld_logits, labels, pred_box, box_len, gt_box,
                     valid, n_vertices, num_classes, cls_grad_scale,
                     loc_grad_scale, sums4, dlogits, dpred_box, counts2,
                     cls_loss_weight, loc_loss_weight, cls_kind, alpha, gamma,
                     class_loc_weight, select_cls, select_loc, cls_sum_weight,
                     point_cls, point_loc);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

// ---- tf.math.top_k membership (models.py:227, 281): mask[i] = 1 for the k
// largest values, equal values in ascending index order (TF: "if two elements
// are equal, the lower-index element appears first"); 0 elsewhere.  A stable
// LSD radix sort of (descending-order key, index) puts exactly that
// selection in the first k places.
namespace pgnn {
namespace {
__global__ void topk_keys_kernel(const float *__restrict__ v, int64_t n,
                                 uint32_t *__restrict__ keys,
                                 uint32_t *__restrict__ vals,
                                 float *__restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float x = v[i];
    if (x == 0.0f) x = 0.0f;            // -0 and +0 compare equal
    uint32_t u = __float_as_uint(x);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order key
    keys[i] = ~u;                                    // ... descending
    vals[i] = (uint32_t)i;
    mask[i] = 0.0f;
  }
}
__global__ void topk_mark_kernel(const uint32_t *__restrict__ order, int64_t k,
                                 float *__restrict__ mask) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < k;
       j += (int64_t)gridDim.x * blockDim.x)
    mask[order[j]] = 1.0f;
}
}  // namespace
}  // namespace pgnn

extern "C" size_t pgnn_topk_mask_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  return align_up((size_t)n * 4, 256) * 4 + radix_sort_scratch_bytes(n) + 256;
}

extern "C" int pgnn_topk_mask_f32(const float *values, int64_t n, int64_t k,
                                  float *mask, void *workspace,
                                  size_t workspace_bytes, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n >= 0 && k >= 0, PGNN_E_INVALID, "topk_mask: bad argument");
  // tf.math.top_k: "input must have at least k columns"
  PGNN_REQUIRE(k <= n, PGNN_E_INVALID,
               "topk_mask: k exceeds the number of values (tf.math.top_k "
               "fails the same way)");
  if (n == 0) return 0;
  PGNN_REQUIRE(values && mask && workspace, PGNN_E_INVALID,
               "topk_mask: null pointer");
  PGNN_REQUIRE(workspace_bytes >= pgnn_topk_mask_workspace_bytes(n),
               PGNN_E_WORKSPACE, "topk_mask: workspace too small");
  Arena ar(workspace, workspace_bytes);
  uint32_t *ka = ar.take<uint32_t>(n), *va = ar.take<uint32_t>(n);
  uint32_t *kb = ar.take<uint32_t>(n), *vb = ar.take<uint32_t>(n);
  const size_t sb = radix_sort_scratch_bytes(n);
  void *scratch = ar.take<char>(sb);
  PGNN_REQUIRE(scratch, PGNN_E_WORKSPACE, "topk_mask: workspace too small");
  hipLaunchKernelGGL(topk_keys_kernel, dim3(grid_for(n, 1024)), dim3(256), 0,
                     stream, values, n, ka, va, mask);
  uint32_t *keys = nullptr, *order = nullptr;
  int rc = radix_sort_pairs(ka, va, kb, vb, n, 32, scratch, sb, &keys, &order,
                            stream, nullptr);
  if (rc) return rc;
  if (k > 0)
    hipLaunchKernelGGL(topk_mark_kernel, dim3(grid_for(k, 1024)), dim3(256), 0,
                       stream, order, k, mask);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_sgd_step(float *params, const float *grads,
                             const float *is_weight, int64_t n, float lr,
                             float grad_scale, float l1_scale, void *stream_) {
  PGNN_GUARD_BEGIN
  if (n <= 0) return 0;
  PGNN_REQUIRE(params && grads && is_weight, PGNN_E_INVALID,
               "sgd_step: null pointer");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0,
                     (hipStream_t)stream_, params, grads, is_weight, n, lr,
                     grad_scale, l1_scale);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_optimizer_step(int32_t kind, float *params,
                                   const float *grads, const float *is_weight,
                                   float *slot0, float *slot1, int64_t n,
                                   float lr, float grad_scale, float l1_scale,
                                   float h0, float h1, float h2, void *stream_) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(kind >= 1 && kind <= 3, PGNN_E_INVALID,
               "optimizer_step: kind must be 1 (momentum), 2 (rmsprop) or 3 (adam)");
  if (n <= 0) return 0;
  PGNN_REQUIRE(params && grads && is_weight && slot0 && (kind == 1 || slot1),
               PGNN_E_INVALID, "optimizer_step: null pointer");
  hipLaunchKernelGGL(optimizer_kernel, dim3(grid_for(n)), dim3(256), 0,
                     (hipStream_t)stream_, kind, params, grads, is_weight, slot0,
                     slot1, n, lr, grad_scale, l1_scale, h0, h1, h2);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_l1_norm(const float *params, const float *is_weight,
                            int64_t n, double *out, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(out, PGNN_E_INVALID, "l1_norm: null out");
  PGNN_HIP(hipMemsetAsync(out, 0, sizeof(double), stream));
  if (n <= 0) return 0;
  PGNN_REQUIRE(params && is_weight, PGNN_E_INVALID, "l1_norm: null pointer");
  hipLaunchKernelGGL(l1_norm_kernel, dim3(grid_for((n + 3) / 4, 128)), dim3(1024),
                     0, stream, params, is_weight, n, out);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
