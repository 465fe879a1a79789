// Native training step (config 4): the forward with saved activations and the
// whole backward of MultiLayerFastLocalGraphModelV2 (models.py:79-163, 170-311
// through tf.gradients in train.py:225-297) driven from C++ -- one call each.
//
// The arithmetic is the library's own primitives (pgnn_mlp_fwd one layer at a
// time, pgnn_scatter_max_f32, pgnn_segmax_fc_bwd_f32, pgnn_weight_grad_f32,
// ...): this file only sequences them, carves every saved activation and
// temporary out of ONE caller-provided workspace (no allocation, no host
// synchronisation) and keeps the device images of the weights fresh with one
// pack launch per parameter update.  A Python-driven step issues the same
// ~300 launches through ctypes at ~10 us each and another ~100 torch glue
// kernels (concat, zero fill, slice add): host-paced.  The Python mirror
// (pointgnn_amd/train.py, Trainer.forward / .backward) is kept as the readable
// composition and the tests hold both to the same gradients.
#include <string.h>

#include <vector>

#include "pgnn_common.h"

namespace pgnn {
// 1: the fused training forward of a GNN stage also writes the gathered hidden
// rows H1 = ReLU(P[src] - Q[dst]) [E, C] and the backward reads them; 0: H1 is
// never materialised -- the backward recomputes the rows it needs from the
// K-row tables P and Q (L2-resident)
int g_train_h1 = 1;
}  // namespace pgnn

namespace {
using namespace pgnn;

inline int pad16(int n) { return (n + 15) / 16 * 16; }

// ---- glue kernels ---------------------------------------------------------------
// dst[r, dc0 + c] (op)= src[r, sc0 + c], c < ncols; ADD: accumulate
template <bool ADD>
__global__ void block_copy_kernel(float *__restrict__ dst, int64_t ldd, int dc0,
                                  const float *__restrict__ src, int64_t lds,
                                  int sc0, int64_t rows, int ncols) {
  const int64_t total = rows * ncols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ncols;
    const int c = (int)(idx - r * ncols);
    const float v = src[r * lds + sc0 + c];
    float *d = dst + r * ldd + dc0 + c;
    *d = ADD ? *d + v : v;
  }
}

// pred[r, j, :L] = y[r, :L]
__global__ void pred_slice_kernel(const float *__restrict__ y, int64_t ldy,
                                  int64_t rows, int L, int nc, int j,
                                  float *__restrict__ pred) {
  const int64_t total = rows * L;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / L;
    const int k = (int)(idx - r * L);
    pred[(r * nc + j) * L + k] = y[r * ldy + k];
  }
}

// a fused group's outputs in one pass (the inverse of fused_dy3_kernel):
// pred[r, lid0 + i, :L] = y[r, base + 8 i : base + 8 i + L], i < n_loc, and,
// for the group that carries the class head, logits[r, :ncp] = y[r, :ncp]
__global__ void pred_slices_kernel(const float *__restrict__ y, int64_t ldy,
                                   int64_t rows, int L, int nc, int lid0,
                                   int n_loc, int base, float *__restrict__ pred,
                                   float *__restrict__ logits /* nullable */,
                                   int ncp) {
  const int n_box = n_loc * L;
  const int per_row = n_box + (logits ? ncp : 0);
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / per_row;
    const int q = (int)(idx - r * per_row);
    if (q < n_box) {
      const int i = q / L, k = q - i * L;
      pred[(r * nc + lid0 + i) * L + k] = y[r * ldy + base + 8 * i + k];
    } else {
      logits[r * ncp + (q - n_box)] = y[r * ldy + (q - n_box)];
    }
  }
}

// dy[r, :] = [dpred[r, j, :L] | 0 ...] (ld = ldy)
__global__ void dpred_slice_kernel(const float *__restrict__ dpred, int64_t rows,
                                   int L, int nc, int j, float *__restrict__ dy,
                                   int ldy) {
  const int64_t total = rows * ldy;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ldy;
    const int k = (int)(idx - r * ldy);
    dy[idx] = k < L ? dpred[(r * nc + j) * L + k] : 0.0f;
  }
}

// The vertex side of a GNN stage's edge input in one pass (gnn.py:337-352):
// hx = [h[:, :c] | x | 0 ...] and, with x' = x + delta, x' itself and
// Q = x' Wx -- pgnn_offset_apply's expression, so Q has the same bits.
__global__ void pre_edge_prep_kernel(const float *__restrict__ h, int64_t ldh,
                                     int c, const float *__restrict__ xyz,
                                     const float *__restrict__ delta,
                                     int64_t ld_delta, int64_t rows,
                                     const float *__restrict__ wx,
                                     float *__restrict__ hx, int ldhx,
                                     float *__restrict__ xyz_out,
                                     float *__restrict__ Q, int ld_q,
                                     float *__restrict__ fill,
                                     int64_t fill_count) {
  // (the stage's aggregation buffer starts from lowest(): filled here instead
  // of by a launch of its own in front of the fused edge kernel)
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < fill_count; idx += (int64_t)gridDim.x * blockDim.x)
    fill[idx] = kFloatLowest;
  const int w = ldhx > ld_q ? ldhx : ld_q;
  const int64_t total = rows * w;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / w;
    const int k = (int)(idx - r * w);
    const float p0 = xyz[3 * r], p1 = xyz[3 * r + 1], p2 = xyz[3 * r + 2];
    if (k < ldhx) {
      float v = 0.0f;
      if (k < c) v = h[r * ldh + k];
      else if (k < c + 3) v = k == c ? p0 : (k == c + 1 ? p1 : p2);
      hx[r * ldhx + k] = v;
    }
    if (k < ld_q) {
      float x0 = p0, x1 = p1, x2 = p2;
      if (delta) {
        x0 = x0 + delta[r * ld_delta];
        x1 = x1 + delta[r * ld_delta + 1];
        x2 = x2 + delta[r * ld_delta + 2];
      }
      if (k < 3) xyz_out[3 * r + k] = k == 0 ? x0 : (k == 1 ? x1 : x2);
      Q[r * ld_q + k] = (x0 * wx[k] + x1 * wx[ld_q + k]) + x2 * wx[2 * ld_q + k];
    }
  }
}

// Gradient w.r.t. a GNN stage's input features (gnn.py:372 residual + the two
// places h enters the stage): dh_in = dh + [dhx[:, :c] + doff[:, :c] | 0]
__global__ void stage_input_grad_kernel(const float *__restrict__ dh, int ldh,
                                        const float *__restrict__ dhx, int ldhx,
                                        const float *__restrict__ doff, int ldo,
                                        int64_t rows, int c,
                                        float *__restrict__ dh_in) {
  const int64_t total = rows * ldh;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ldh;
    const int k = (int)(idx - r * ldh);
    float v = dh[idx];
    if (k < c) {
      v += dhx[r * ldhx + k];
      if (doff) v += doff[r * ldo + k];
    }
    dh_in[idx] = v;
  }
}

__global__ void edge_dst_kernel(const int32_t *__restrict__ edges, int64_t n,
                                int32_t *__restrict__ dst,
                                float *__restrict__ fill, int64_t fill_count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = edges[2 * i + 1];
  // (optional: a pooling stage's aggregation buffer starts from lowest())
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < fill_count;
       i += (int64_t)gridDim.x * blockDim.x)
    fill[i] = kFloatLowest;
}

inline unsigned blocks_for(int64_t total, int cap = 2048) {
  int64_t b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- the handle -------------------------------------------------------------------
struct PackJobRec {  // = train.hip's PackJob (pgnn_pack_fc_many's record)
  const float *w;
  const float *b;
  float *dst;
  int32_t k_in, n_out, kind, first_block, ld, reserved;
};
static_assert(sizeof(PackJobRec) == 48, "job record layout");

struct FcDev {  // one layer: where it lives + its device images
  pgnn_train_fc ref;  // dims; offsets into the flat buffers (not for fused layers)
  float *w = nullptr, *b = nullptr;    // weights / biases (flat buffer or fused)
  float *gw = nullptr, *gb = nullptr;  // their gradients
  float *packed = nullptr, *packed_t = nullptr, *wt = nullptr;
  size_t off_packed = 0, off_packed_t = 0, off_wt = 0;
  bool want_wt = false;
};

// The prediction heads of one group as three fused layers (the block layout
// of the inference path, pointgnn_amd/gnn.py ClassAwarePredictor): all first
// layers side by side, block-diagonal second and third layers, the class
// logits passed through the third layer by an identity block.  ~75 launches
// of 64-wide layers per step become ~13.
struct HeadGroup {
  bool has_cls = false;
  std::vector<int> lids;  // box heads in this group
  int base = 0;           // columns reserved for the logits in layers 2 / 3
  FcDev f[3];             // fused layers (weights / gradients live in `images`)
  size_t off_w[3] = {0, 0, 0}, off_b[3] = {0, 0, 0}, off_gw[3] = {0, 0, 0},
         off_gb[3] = {0, 0, 0};
};

struct StageDev {
  int kind, level;
  std::vector<FcDev> a, b, c;
  // gnn: images of Wx = rows c..c+2 of a[0]
  float *wx = nullptr, *wx_packed_t = nullptr;
  size_t off_wx = 0, off_wx_packed_t = 0;
};

struct Trainer {
  pgnn_train_model m;
  // the `train_h1` tunable as the most recent pgnn_trainer_forward saw it: the
  // backward of that step reads (or not) the H1 rows the forward wrote (or
  // not), whatever the tunable says by then
  int h1 = 1;
  std::vector<StageDev> stages;
  FcDev cls[2];
  std::vector<FcDev> loc;  // 3 per class
  std::vector<HeadGroup> groups;  // empty: heads run one by one
  float *params = nullptr, *grads = nullptr;
  char *images = nullptr;
  size_t images_bytes = 0, off_jobs = 0, off_jobs_a = 0, off_jobs_u = 0;
  size_t fused_lo = 0, fused_hi = 0;  // byte range of the fused matrices
  int n_jobs = 0, total_blocks = 0;       // fragment / transpose images
  int n_jobs_a = 0, total_blocks_a = 0;   // assemble fused matrices (before)
  int n_jobs_u = 0, total_blocks_u = 0;   // hand fused gradients back (after)
};

size_t packed_floats(int k_in, int n_out) {
  return pgnn_packed_fc_floats(k_in, n_out);
}

// groups of heads as the inference path forms them (gnn.py of the package,
// ClassAwarePredictor): returns false when the heads do not have the shipped
// shape (C -> hw -> nc and C -> hw -> hw -> L with one hw, L <= 8)
bool plan_head_groups(Trainer &t) {
  const int nc = t.m.num_classes, L = t.m.box_len;
  const int C = t.cls[0].ref.k_in, hw = t.cls[0].ref.n_out;
  if (L > 8 || nc > 16 || t.cls[1].ref.k_in != hw || t.cls[1].ref.n_out != nc)
    return false;
  for (int j = 0; j < nc; ++j) {
    const FcDev *f = &t.loc[3 * j];
    if (f[0].ref.k_in != C || f[0].ref.n_out != hw || f[1].ref.k_in != hw ||
        f[1].ref.n_out != hw || f[2].ref.k_in != hw || f[2].ref.n_out != L)
      return false;
  }
  if (hw > 160) return false;
  const int per_first = (320 / hw - 1) > 1 ? (320 / hw - 1) : 1;
  const int per_rest = (320 / hw) > 1 ? (320 / hw) : 1;
  int next = 0;
  bool first = true;
  while (first || next < nc) {
    HeadGroup g;
    g.has_cls = first;
    g.base = first ? 16 : 0;
    const int take = first ? per_first : per_rest;
    for (int i = 0; i < take && next < nc; ++i) g.lids.push_back(next++);
    const int nh = (int)g.lids.size() + (first ? 1 : 0);
    const int nl = (int)g.lids.size();
    g.f[0].ref = {0, 0, C, hw * nh};
    g.f[1].ref = {0, 0, hw * nh, g.base + hw * nl};
    g.f[2].ref = {0, 0, g.base + hw * nl, g.base + 8 * nl};
    t.groups.push_back(g);
    first = false;
  }
  return true;
}

// lay out every image; returns the bytes needed (job tables included)
size_t layout_images(Trainer &t) {
  size_t off = 0;
  auto take = [&](size_t floats) {
    const size_t at = off;
    off += align_up(floats * 4, 256);
    return at;
  };
  int jobs = 0;
  auto lay = [&](FcDev &f) {
    f.off_packed = take(packed_floats(f.ref.k_in, f.ref.n_out));
    f.off_packed_t = take(packed_floats(f.ref.n_out, f.ref.k_in));
    jobs += 2;
    if (f.want_wt) {
      f.off_wt = take((size_t)f.ref.n_out * pad16(f.ref.k_in));
      ++jobs;
    }
  };
  for (StageDev &s : t.stages) {
    for (FcDev &f : s.a) lay(f);
    for (FcDev &f : s.b) lay(f);
    for (FcDev &f : s.c) lay(f);
    if (s.kind == 1) {
      const int n_out = s.a[0].ref.n_out;
      s.off_wx = take((size_t)3 * pad16(n_out));
      s.off_wx_packed_t = take(packed_floats(n_out, 3));
      jobs += 2;
    }
  }
  int jobs_a = 0, jobs_u = 0;
  if (t.groups.empty()) {
    for (FcDev &f : t.cls) lay(f);
    for (FcDev &f : t.loc) lay(f);
  } else {
    // fused weight / bias matrices (zeroed once, parameter blocks copied in
    // before every pack) in one contiguous range
    t.fused_lo = off;
    for (HeadGroup &g : t.groups)
      for (int i = 0; i < 3; ++i) {
        g.off_w[i] = take((size_t)g.f[i].ref.k_in * g.f[i].ref.n_out);
        g.off_b[i] = take((size_t)g.f[i].ref.n_out);
      }
    t.fused_hi = off;
    for (HeadGroup &g : t.groups)
      for (int i = 0; i < 3; ++i) {
        g.off_gw[i] = take((size_t)g.f[i].ref.k_in * g.f[i].ref.n_out);
        g.off_gb[i] = take((size_t)g.f[i].ref.n_out);
        lay(g.f[i]);
      }
    for (HeadGroup &g : t.groups) {
      const int blocks = (g.has_cls ? 2 : 0) + 3 * (int)g.lids.size();
      jobs_a += 2 * blocks;  // weights + biases
      jobs_u += 2 * blocks;
    }
  }
  t.off_jobs = off;
  t.n_jobs = jobs;
  off += align_up((size_t)jobs * sizeof(PackJobRec), 256);
  t.off_jobs_a = off;
  t.n_jobs_a = jobs_a;
  off += align_up((size_t)(jobs_a + 1) * sizeof(PackJobRec), 256);
  t.off_jobs_u = off;
  t.n_jobs_u = jobs_u;
  off += align_up((size_t)(jobs_u + 1) * sizeof(PackJobRec), 256);
  return off + 256;
}

// ---- workspace ------------------------------------------------------------------------
// Bump allocator that also runs "dry" (null base) to size the workspace: the
// forward and the backward replay the same allocation sequence.
struct Bump {
  char *base;
  size_t off, cap, high;
  Bump(void *p, size_t n) : base((char *)p), off(0), cap(n), high(0) {}
  void *raw(size_t bytes) {
    const size_t at = align_up(off, 256);
    off = at + bytes;
    if (off > high) high = off;
    if (!base) return (void *)(uintptr_t)(at + 256);  // dry run: never used
    if (off > cap) return nullptr;
    return base + at;
  }
  float *f(int64_t rows, int64_t ld) { return (float *)raw((size_t)rows * ld * 4); }
  int32_t *i32(int64_t n) { return (int32_t *)raw((size_t)n * 4); }
};

struct PoolSaved {
  float *feat;                 // [E, 16]
  float *act[PGNN_TRAIN_MAX_FC];   // outputs of the point MLP layers
  int32_t *dst;
  float *agg;                  // [K, pad(n_out of last a)]
  float *oact[PGNN_TRAIN_MAX_FC];  // outputs of the output MLP layers
};
struct GnnSaved {
  const float *h_in;           // [K, ld_h]
  float *off_act[PGNN_TRAIN_MAX_FC];  // outputs of the offset MLP layers
  float *xo, *q, *hx, *p;
  float *eact[PGNN_TRAIN_MAX_FC];  // eact[0] = H1, eact[i] = output of a[i]
  int32_t *dst;
  float *agg;
  float *uact[PGNN_TRAIN_MAX_FC];  // outputs of the update MLP layers
};
struct HeadsSaved {
  float *y1[4], *y2[4], *y3[4];  // fused groups: outputs of the three layers
  float *c1, *logits;          // [K, 64], [K, pad(nc)]
  float *l1[PGNN_TRAIN_MAX_CLASSES], *l2[PGNN_TRAIN_MAX_CLASSES],
      *l3[PGNN_TRAIN_MAX_CLASSES];
  float *pred;                 // [K, nc, L]
};
struct Saved {  // lives at the start of the workspace (host-visible copy kept
                // in the handle would break re-entrancy: it is recomputed by
                // replaying the allocation sequence in backward)
  PoolSaved pool[PGNN_TRAIN_MAX_STAGES];
  GnnSaved gnn[PGNN_TRAIN_MAX_STAGES];
  HeadsSaved heads;
  const float *h_final;
  int ld_h_final;
  int64_t k_final;
  float *scratch;              // weight-grad / segmax workspace
  size_t scratch_bytes;
};

struct Ctx {
  Trainer &t;
  const pgnn_train_batch &b;
  Bump &ws;
  hipStream_t stream;
  bool dry;  // sizing run: allocate only
  // weight gradients of the K-row layers, recorded by the backward and run in
  // ONE launch pair at its end (pgnn_weight_grad_many_f32): only dX is on the
  // backward's critical path.  Recorded in the sizing run too (the partial
  // buffer comes out of the same workspace).
  std::vector<pgnn_wgrad_job> wjobs;
  // a second batch, run after the first: jobs that add into a dW the first
  // batch also writes (the Wx rows of a GNN stage's first edge layer)
  std::vector<pgnn_wgrad_job> wjobs2;
};

int check_batch(const Trainer &t, const pgnn_train_batch *b) {
  PGNN_REQUIRE(b && b->n_levels >= 1 && b->n_levels <= PGNN_TRAIN_MAX_LEVELS &&
                   b->n_feat >= 0 && b->n_feat <= 13,
               PGNN_E_INVALID, "trainer: bad batch header");
  for (int l = 0; l <= b->n_levels; ++l)
    PGNN_REQUIRE(b->n_vertices[l] >= 0, PGNN_E_INVALID, "trainer: n_vertices < 0");
  for (int l = 0; l < b->n_levels; ++l)
    PGNN_REQUIRE(b->n_edges[l] >= 0, PGNN_E_INVALID, "trainer: n_edges < 0");
  for (const StageDev &s : t.stages)
    PGNN_REQUIRE(s.level >= 0 && s.level < b->n_levels, PGNN_E_INVALID,
                 "trainer: stage graph_level outside the batch");
  return 0;
}

// one FC layer forward on `rows` rows: y = act(x[:, :k_in] W + b) (+ residual);
// relu_from: ReLU on output columns >= relu_from (0 = all, n_out = linear)
int fc_fwd_from(Ctx &c, const FcDev &f, const float *x, int64_t ldx,
                int64_t rows, int relu_from, const float *residual,
                int64_t ld_res, float *y) {
  if (c.dry || rows == 0) return 0;
  pgnn_fc_layer L;
  L.packed = f.packed;
  L.k_in = f.ref.k_in;
  L.n_out = f.ref.n_out;
  L.relu_from = relu_from;
  return pgnn_mlp_fwd(x, ldx, f.ref.k_in, nullptr, 0, 0, rows, &L, 1, residual,
                      ld_res, y, pad16(f.ref.n_out), c.stream);
}
int fc_fwd(Ctx &c, const FcDev &f, const float *x, int64_t ldx, int64_t rows,
           bool relu, const float *residual, int64_t ld_res, float *y) {
  return fc_fwd_from(c, f, x, ldx, rows, relu ? 0 : f.ref.n_out, residual,
                     ld_res, y);
}

// A chain of FC layers on the K vertex rows with every layer's output kept
// (the backward needs them): ONE launch of the 8-wave row kernel when it
// applies (mlp_rows_chain: the intermediate rows stay in LDS and are written
// out on the way), else layer by layer.  relu_from[i] as in pgnn_fc_layer; the
// residual is added to the last layer's rows.
int fc_chain_fwd(Ctx &c, const FcDev *const *f, const int *relu_from, int n,
                 const float *x, int64_t ldx, int64_t rows, const float *residual,
                 int64_t ld_res, float *const *outs) {
  if (c.dry || rows == 0 || n == 0) return 0;
  if (n >= 2 && n <= PGNN_MAX_LAYERS) {
    pgnn_fc_layer L[PGNN_MAX_LAYERS];
    RowsTap taps[PGNN_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
      L[i].packed = f[i]->packed;
      L[i].k_in = f[i]->ref.k_in;
      L[i].n_out = f[i]->ref.n_out;
      L[i].relu_from = relu_from[i];
      taps[i].mid = outs[i];
      taps[i].gate = nullptr;
    }
    const int rc = mlp_rows_chain(x, ldx, f[0]->ref.k_in, rows, L, n, taps,
                                  residual, ld_res, 0, outs[n - 1],
                                  pad16(f[n - 1]->ref.n_out), c.stream);
    if (rc != PGNN_E_UNSUPPORTED) return rc;
  }
  for (int i = 0; i < n; ++i) {
    const bool last = i + 1 == n;
    const int rc = fc_fwd_from(c, *f[i], x, ldx, rows, relu_from[i],
                               last ? residual : nullptr, last ? ld_res : 0,
                               outs[i]);
    if (rc) return rc;
    x = outs[i];
    ldx = pad16(f[i]->ref.n_out);
  }
  return 0;
}

// dX = dY W^T through the forward engine on the transposed image
// gate (nullable, [rows, ld_gate]): the layer's forward input when that is a
// ReLU output -- the ReluGrad of the layer below, dx = gate > 0 ? dx : 0,
// leaves with the rows instead of taking a launch of its own
int fc_dx(Ctx &c, const FcDev &f, const float *dy, int64_t lddy, int64_t rows,
          float *dx, const float *gate = nullptr, int64_t ld_gate = 0) {
  if (c.dry || rows == 0) return 0;
  pgnn_fc_layer L;
  L.packed = f.packed_t;
  L.k_in = f.ref.n_out;
  L.n_out = f.ref.k_in;
  L.relu_from = f.ref.k_in;  // linear
  if (gate) {
    // (every caller's gate has the padded rows of the layer's input; a
    // narrower one would need an ld-aware mask)
    PGNN_REQUIRE(ld_gate >= pad16(f.ref.k_in), PGNN_E_INVALID,
                 "fc_dx: gate rows narrower than the padded input width");
    return mlp_rows_gated(dy, lddy, f.ref.n_out, rows, &L, gate, ld_gate, dx,
                          pad16(f.ref.k_in), c.stream);
  }
  return pgnn_mlp_fwd(dy, lddy, f.ref.n_out, nullptr, 0, 0, rows, &L, 1, nullptr,
                      0, dx, pad16(f.ref.k_in), c.stream);
}

// record dW/db = (x^T dy, column sums) for the end of the backward; x and dy
// must stay untouched until then
void defer_wgrad(Ctx &c, const float *x, int64_t ldx, int k_in, const float *dy,
                 int64_t lddy, int n_out, int64_t rows, float *gw, float *gb,
                 bool accumulate) {
  pgnn_wgrad_job j = {};
  j.X = x;
  j.ld_x = ldx;
  j.dZ = dy;
  j.ld_dz = lddy;
  j.n_rows = rows;
  j.dW = gw;
  j.db = gb;
  j.k_in = k_in;
  j.n_out = n_out;
  j.accumulate = accumulate ? 1 : 0;
  c.wjobs.push_back(j);
}

int fc_wgrad(Ctx &c, Saved &sv, const FcDev &f, const float *x, int64_t ldx,
             const float *dy, int64_t lddy, int64_t rows, bool accumulate = true,
             bool defer = false) {
  if (rows == 0) return 0;
  if (defer && f.ref.n_out <= 320) {
    defer_wgrad(c, x, ldx, f.ref.k_in, dy, lddy, f.ref.n_out, rows, f.gw, f.gb,
                accumulate);
    return 0;
  }
  if (c.dry) return 0;
  return pgnn_weight_grad_f32(x, ldx, f.ref.k_in, dy, lddy, f.ref.n_out, rows,
                              f.gw, f.gb, accumulate ? 1 : 0, sv.scratch,
                              sv.scratch_bytes, c.stream);
}

// fc_bwd of the Python mirror: optional ReluGrad (in place on dy), dW/db, dX.
// defer: the weight gradient is recorded for the end of the backward -- the
// caller guarantees that x and dy are not written again before that.
// gate_dx: x is the output of a ReLU layer whose backward comes next: dx leaves
// already masked (the next fc_bwd is then called with relu = false).
int fc_bwd(Ctx &c, Saved &sv, const FcDev &f, const float *x, int64_t ldx,
           const float *y, float *dy, int64_t rows, bool relu, float *dx,
           bool accumulate = true, bool defer = false, bool gate_dx = false) {
  if (rows == 0) return 0;
  const int ldy = pad16(f.ref.n_out);
  int rc = 0;
  if (relu && !c.dry) {
    rc = pgnn_relu_mask_mul(dy, y, rows * ldy, c.stream);
    if (rc) return rc;
  }
  rc = fc_wgrad(c, sv, f, x, ldx, dy, ldy, rows, accumulate, defer);
  if (rc) return rc;
  if (dx && !c.dry)
    rc = fc_dx(c, f, dy, ldy, rows, dx, gate_dx ? x : nullptr, ldx);
  return rc;
}

// Backward of a chain of K-row layers f[0..n-1] (forward order) whose top
// gradient dy_top is final: every weight gradient is recorded for the end of
// the backward, and ALL the dX -- dxs[i] = dy_i W_i^T, masked by xin[i] > 0
// where gate[i] (xin[i] is then the ReLU output of layer i - 1: its ReluGrad)
// -- come from one launch when mlp_rows_chain applies (dxs[i] is the dy of
// layer i - 1: it stays in LDS and is written out on the way).
// pre (nullable): dy_top itself is a linear map of other rows, dy_top = pre->x
// pre->W (a dX step without a weight gradient of its own) -- it becomes the
// chain's first layer.
struct PreDx {
  const void *packed;  // device image of W [k_in, n_out]
  int k_in, n_out;
  const float *x;
  int64_t ldx;
};
int pre_dx(Ctx &c, const PreDx &p, int64_t rows, float *dy_top) {
  pgnn_fc_layer L;
  L.packed = (const float *)p.packed;
  L.k_in = p.k_in;
  L.n_out = p.n_out;
  L.relu_from = p.n_out;
  return pgnn_mlp_fwd(p.x, p.ldx, p.k_in, nullptr, 0, 0, rows, &L, 1, nullptr, 0,
                      dy_top, pad16(p.n_out), c.stream);
}
int fc_chain_bwd(Ctx &c, Saved &sv, const FcDev *const *f, int n,
                 const float *const *xin, const int64_t *ldx, float *dy_top,
                 int64_t rows, float *const *dxs, const bool *gate,
                 bool accumulate = true, const PreDx *pre = nullptr) {
  if (rows == 0 || n == 0) return 0;
  const int np = pre ? 1 : 0;
  bool chain = n + np >= 2 && n + np <= PGNN_MAX_LAYERS;
  for (int i = 0; i < n && chain; ++i)
    chain = f[i]->ref.n_out <= 320 && f[i]->packed_t != nullptr &&
            (!gate[i] || ldx[i] == pad16(f[i]->ref.k_in));
  int rc = 0;
  if (chain) {
    // (recorded in the sizing run too; a job only holds pointers)
    const float *dy = dy_top;
    for (int i = n - 1; i >= 0; --i) {
      rc = fc_wgrad(c, sv, *f[i], xin[i], ldx[i], dy, pad16(f[i]->ref.n_out), rows,
                    accumulate, true);
      if (rc) return rc;
      dy = dxs[i];
    }
    if (c.dry) return 0;
    pgnn_fc_layer L[PGNN_MAX_LAYERS];
    RowsTap taps[PGNN_MAX_LAYERS];
    if (pre) {
      L[0].packed = (const float *)pre->packed;
      L[0].k_in = pre->k_in;
      L[0].n_out = pre->n_out;
      L[0].relu_from = pre->n_out;
      taps[0].mid = dy_top;
      taps[0].gate = nullptr;
    }
    for (int j = 0; j < n; ++j) {
      const FcDev &g = *f[n - 1 - j];
      L[np + j].packed = g.packed_t;
      L[np + j].k_in = g.ref.n_out;
      L[np + j].n_out = g.ref.k_in;
      L[np + j].relu_from = g.ref.k_in;  // linear
      taps[np + j].mid = dxs[n - 1 - j];
      taps[np + j].gate = gate[n - 1 - j] ? xin[n - 1 - j] : nullptr;
    }
    rc = mlp_rows_chain(pre ? pre->x : dy_top,
                        pre ? pre->ldx : (int64_t)pad16(f[n - 1]->ref.n_out),
                        pre ? pre->k_in : f[n - 1]->ref.n_out, rows, L, n + np, taps,
                        gate[0] ? xin[0] : nullptr, ldx[0], gate[0] ? 1 : 0,
                        dxs[0], pad16(f[0]->ref.k_in), c.stream);
    if (rc != PGNN_E_UNSUPPORTED) return rc;
    // the dX launches one by one (the weight gradients are already recorded)
    if (pre) {
      rc = pre_dx(c, *pre, rows, dy_top);
      if (rc) return rc;
    }
    float *d = dy_top;
    for (int i = n - 1; i >= 0; --i) {
      rc = fc_dx(c, *f[i], d, pad16(f[i]->ref.n_out), rows, dxs[i],
                 gate[i] ? xin[i] : nullptr, ldx[i]);
      if (rc) return rc;
      d = dxs[i];
    }
    return 0;
  }
  if (pre && !c.dry) {
    rc = pre_dx(c, *pre, rows, dy_top);
    if (rc) return rc;
  }
  float *d = dy_top;
  for (int i = n - 1; i >= 0; --i) {
    rc = fc_bwd(c, sv, *f[i], xin[i], ldx[i], nullptr, d, rows, false, dxs[i],
                accumulate, true, gate[i]);
    if (rc) return rc;
    d = dxs[i];
  }
  return 0;
}

// run the recorded weight gradients (one launch pair).  Their partial sums
// take a fixed slice of the workspace (the sizing run records no jobs: it
// does not walk the launch code): pgnn_weight_grad_many_f32 aims for 4
// workgroups per CU over all jobs, each writing a [64 x <= 320] block.
size_t wgrad_many_bound() {
  return ((size_t)4 * device_cu_count() + 8 * 64) * 64 * 320 * 4 + 256;
}
int flush_wgrads(Ctx &c, Saved &sv) {
  const size_t bound = wgrad_many_bound();
  void *part = c.ws.raw(bound);
  int rc = 0;
  for (std::vector<pgnn_wgrad_job> *jobs : {&c.wjobs, &c.wjobs2}) {
    if (!c.dry && !jobs->empty() && rc == 0) {
      PGNN_REQUIRE(part != nullptr, PGNN_E_WORKSPACE,
                   "trainer: workspace too small");
      const size_t bytes =
          pgnn_weight_grad_many_workspace_bytes(jobs->data(), (int32_t)jobs->size());
      if (bytes <= bound) {
        rc = pgnn_weight_grad_many_f32(jobs->data(), (int32_t)jobs->size(), part,
                                       bound, c.stream);
      } else {  // (more jobs than the bound foresees: one by one)
        for (const pgnn_wgrad_job &j : *jobs) {
          rc = pgnn_weight_grad_f32(j.X, j.ld_x, j.k_in, j.dZ, j.ld_dz, j.n_out,
                                    j.n_rows, j.dW, j.db, j.accumulate, sv.scratch,
                                    sv.scratch_bytes, c.stream);
          if (rc) break;
        }
      }
    }
    jobs->clear();
  }
  return rc;
}

size_t scratch_need(const Trainer &t, const pgnn_train_batch &b) {
  size_t need = 256;
  auto wg = [&](const FcDev &f, int64_t rows) {
    const size_t n = pgnn_weight_grad_workspace_bytes(f.ref.k_in, f.ref.n_out, rows);
    if (n > need) need = n;
  };
  for (const StageDev &s : t.stages) {
    const int64_t E = b.n_edges[s.level], K = b.n_vertices[s.level + 1];
    for (const FcDev &f : s.a) wg(f, s.kind == 0 ? E : (&f == &s.a[0] ? K : E));
    for (const FcDev &f : s.b) wg(f, K);
    for (const FcDev &f : s.c) wg(f, K);
    const FcDev &last = s.a.back();
    if (last.want_wt) {
      const size_t n = pgnn_segmax_fc_bwd_workspace_bytes(E, last.ref.n_out,
                                                          (int32_t)K, last.ref.k_in);
      if (n > need) need = n;
    }
    if (s.kind == 1) {  // the Wx rows: k_in = 3
      const size_t n = pgnn_weight_grad_workspace_bytes(3, s.a[0].ref.n_out, K);
      if (n > need) need = n;
    } else {  // the fused backward of the narrow pooling layers
      const size_t n = pgnn_pool_narrow_bwd_workspace_bytes(E);
      if (n > need) need = n;
    }
  }
  const int64_t K = b.n_vertices[b.n_levels];
  for (const FcDev &f : t.cls) wg(f, K);
  for (const FcDev &f : t.loc) wg(f, K);
  for (const HeadGroup &g : t.groups)
    for (int i = 0; i < 3; ++i) wg(g.f[i], K);
  return need;
}

// ---- forward ---------------------------------------------------------------------------
// Allocation sequence of the forward (shared by the sizing run, the forward
// and the backward, which re-derives the same pointers).
int forward_impl(Ctx &c, Saved &sv) {
  Trainer &t = c.t;
  const pgnn_train_batch &b = c.b;
  int rc = 0;
  sv.scratch_bytes = scratch_need(t, b);
  sv.scratch = (float *)c.ws.raw(sv.scratch_bytes);
  const float *h = nullptr;
  int ld_h = 0;
  int64_t k_h = 0;
  // the destination column of a level's edge list, extracted once per level
  // (the three GNN stages of the shipped models share level 1's)
  int32_t *dst_of[PGNN_TRAIN_MAX_LEVELS] = {nullptr};
  // (fill / fill_count: a buffer the same launch sets to lowest(); *filled
  // says whether that happened -- not when the level's column existed already)
  auto level_dst = [&](int lvl, int64_t E, float *fill = nullptr,
                       int64_t fill_count = 0,
                       bool *filled = nullptr) -> int32_t * {
    if (filled) *filled = false;
    if (dst_of[lvl]) return dst_of[lvl];
    int32_t *d = c.ws.i32(E > 0 ? E : 1);
    if (!c.dry && E > 0 && d) {
      hipLaunchKernelGGL(edge_dst_kernel, dim3(blocks_for(E)), dim3(256), 0,
                         c.stream, b.edges[lvl], E, d, fill,
                         fill ? fill_count : (int64_t)0);
      if (filled && fill) *filled = true;
    }
    dst_of[lvl] = d;
    return d;
  };
  for (size_t si = 0; si < t.stages.size(); ++si) {
    StageDev &s = t.stages[si];
    const int lvl = s.level;
    const int64_t E = b.n_edges[lvl], K = b.n_vertices[lvl + 1];
    if (s.kind == 0) {
      PoolSaved &p = sv.pool[si];
      p.feat = c.ws.f(E, 16);
      for (size_t i = 0; i < s.a.size(); ++i) p.act[i] = c.ws.f(E, pad16(s.a[i].ref.n_out));
      const int wa = pad16(s.a.back().ref.n_out);
      p.agg = c.ws.f(K, wa);
      // (the launch that extracts the level's dst column also sets the
      // aggregation buffer to lowest(): no fill launch of its own)
      bool agg_filled = false;
      p.dst = level_dst(lvl, E, p.agg, (int64_t)K * wa, &agg_filled);
      for (size_t i = 0; i < s.b.size(); ++i) p.oact[i] = c.ws.f(K, pad16(s.b[i].ref.n_out));
      if (!c.dry) {
        PGNN_REQUIRE(s.a[0].ref.k_in == b.n_feat + 3, PGNN_E_INVALID,
                     "trainer: point MLP input width != n_feat + 3");
        rc = pgnn_pool_features_fwd(b.input_v, b.n_feat, b.coords[lvl],
                                    b.keypoints[lvl], b.edges[lvl], E, p.feat,
                                    c.stream);
        if (rc) return rc;
        const float *x = p.feat;
        int64_t ldx = 16;
        bool fused = false;
        if (s.a.size() == 4 && s.a.back().want_wt &&
            pad16(s.a[0].ref.n_out) == 32 && pad16(s.a[1].ref.n_out) == 64 &&
            pad16(s.a[2].ref.n_out) == 128) {
          // gather + the whole point MLP + scatter-max in ONE kernel that
          // also writes the four layers' activations
          pgnn_fc_layer Ls[4];
          for (int i = 0; i < 4; ++i) {
            Ls[i].packed = s.a[i].packed;
            Ls[i].k_in = s.a[i].ref.k_in;
            Ls[i].n_out = s.a[i].ref.n_out;
            Ls[i].relu_from = 0;
          }
          float *acts[4] = {p.act[0], p.act[1], p.act[2], p.act[3]};
          rc = pgnn_point_set_pooling_rows_fwd(
              b.input_v, b.n_feat, b.coords[lvl], b.keypoints[lvl], b.edges[lvl],
              E, (int32_t)K, Ls, 4,
              (b.edges_sorted[lvl] ? 1 : 0) | (agg_filled ? 2 : 0), p.agg, wa,
              acts, wa, c.stream);
          if (rc == 0) fused = true;
          else if (rc != PGNN_E_UNSUPPORTED) return rc;
        }
        if (!fused) {
          for (size_t i = 0; i < s.a.size(); ++i) {
            rc = fc_fwd(c, s.a[i], x, ldx, E, true, nullptr, 0, p.act[i]);
            if (rc) return rc;
            x = p.act[i];
            ldx = pad16(s.a[i].ref.n_out);
          }
          rc = pgnn_scatter_max_f32(x, ldx, p.dst, E, (int32_t)ldx, (int32_t)K,
                                    p.agg, wa, b.edges_sorted[lvl] ? 1 : 0,
                                    c.stream);
          if (rc) return rc;
        }
        {  // output MLP, every layer a ReLU layer: one launch
          const FcDev *fs[PGNN_TRAIN_MAX_FC];
          int rf[PGNN_TRAIN_MAX_FC];
          for (size_t i = 0; i < s.b.size(); ++i) {
            fs[i] = &s.b[i];
            rf[i] = 0;
          }
          rc = fc_chain_fwd(c, fs, rf, (int)s.b.size(), p.agg, wa, K, nullptr, 0,
                            p.oact);
          if (rc) return rc;
        }
      }
      h = p.oact[s.b.size() - 1];
      ld_h = pad16(s.b.back().ref.n_out);
      k_h = K;
    } else {
      GnnSaved &g = sv.gnn[si];
      PGNN_REQUIRE(h != nullptr || c.dry, PGNN_E_INVALID,
                   "trainer: a GNN stage needs vertex features from a pooling stage");
      PGNN_REQUIRE(c.dry || k_h == K, PGNN_E_INVALID,
                   "trainer: GNN stage vertex count != feature rows");
      const FcDev &w1 = s.a[0];
      const int cc = w1.ref.k_in - 3;
      const int wq = pad16(w1.ref.n_out);
      g.h_in = h;
      for (size_t i = 0; i < s.c.size(); ++i) g.off_act[i] = c.ws.f(K, pad16(s.c[i].ref.n_out));
      g.xo = c.ws.f(K, 3);
      g.q = c.ws.f(K, wq);
      g.hx = c.ws.f(K, pad16(cc + 3));
      g.p = c.ws.f(K, wq);
      g.eact[0] = c.ws.f(E, wq);
      for (size_t i = 1; i < s.a.size(); ++i) g.eact[i] = c.ws.f(E, pad16(s.a[i].ref.n_out));
      g.dst = level_dst(lvl, E);
      const int wa = pad16(s.a.back().ref.n_out);
      g.agg = c.ws.f(K, wa);
      for (size_t i = 0; i < s.b.size(); ++i) g.uact[i] = c.ws.f(K, pad16(s.b[i].ref.n_out));
      if (!c.dry) {
        PGNN_REQUIRE(ld_h >= cc, PGNN_E_INVALID,
                     "trainer: vertex features narrower than the edge MLP input");
        PGNN_REQUIRE(s.b.back().ref.n_out == cc && ld_h >= pad16(cc), PGNN_E_INVALID,
                     "trainer: update MLP must preserve the feature width");
        const float *delta = nullptr;
        int64_t ld_delta = 0;
        if (!s.c.empty()) {  // offset MLP, last layer linear: one launch
          PGNN_REQUIRE(s.c.back().ref.n_out == 3, PGNN_E_INVALID,
                       "trainer: the offset MLP must end in 3 outputs");
          const FcDev *fs[PGNN_TRAIN_MAX_FC];
          int rf[PGNN_TRAIN_MAX_FC];
          for (size_t i = 0; i < s.c.size(); ++i) {
            fs[i] = &s.c[i];
            rf[i] = i + 1 < s.c.size() ? 0 : s.c[i].ref.n_out;
          }
          rc = fc_chain_fwd(c, fs, rf, (int)s.c.size(), h, ld_h, K, nullptr, 0,
                            g.off_act);
          if (rc) return rc;
          delta = g.off_act[s.c.size() - 1];
          ld_delta = pad16(s.c.back().ref.n_out);
        }
        if (K > 0) {
          const int ldhx = pad16(cc + 3);
          hipLaunchKernelGGL(pre_edge_prep_kernel,
                             dim3(blocks_for(K * (ldhx > wq ? ldhx : wq), 8192)),
                             dim3(256), 0, c.stream, h, (int64_t)ld_h, cc,
                             b.coords[lvl], delta, ld_delta, K, s.wx, g.hx, ldhx,
                             g.xo, g.q, wq, g.agg, (int64_t)K * wa);
        }
        rc = fc_fwd(c, w1, g.hx, pad16(cc + 3), K, false, nullptr, 0, g.p);
        if (rc) return rc;
        bool fused = false;
        if (s.a.size() == 2 && s.a[1].want_wt) {
          // gather + last edge layer + scatter-max in ONE kernel that also
          // writes the layer's rows and the gathered hidden rows H1
          pgnn_fc_layer L2;
          L2.packed = s.a[1].packed;
          L2.k_in = s.a[1].ref.k_in;
          L2.n_out = s.a[1].ref.n_out;
          L2.relu_from = 0;
          rc = pgnn_edge_mlp_scatter_max_rows_fwd(
              g.p, g.q, wq, s.a[1].ref.k_in, b.edges[lvl], E, (int32_t)K, &L2,
              (b.edges_sorted[lvl] ? 1 : 0) | (K > 0 ? 2 : 0), g.agg, wa,
              g.eact[1], wa, c.t.h1 ? g.eact[0] : nullptr, c.stream);
          if (rc == 0) fused = true;
          else if (rc != PGNN_E_UNSUPPORTED) return rc;
        }
        if (!fused) {
          rc = pgnn_edge_hidden_fwd(g.p, g.q, wq, b.edges[lvl], E, g.eact[0],
                                    c.stream);
          if (rc) return rc;
          for (size_t i = 1; i < s.a.size(); ++i) {
            rc = fc_fwd(c, s.a[i], g.eact[i - 1], pad16(s.a[i - 1].ref.n_out), E,
                        true, nullptr, 0, g.eact[i]);
            if (rc) return rc;
          }
          rc = pgnn_scatter_max_f32(g.eact[s.a.size() - 1], wa, g.dst, E, wa,
                                    (int32_t)K, g.agg, wa,
                                    b.edges_sorted[lvl] ? 1 : 0, c.stream);
          if (rc) return rc;
        }
        {  // update MLP, last layer linear, + h (gnn.py:372): one launch
          const FcDev *fs[PGNN_TRAIN_MAX_FC];
          int rf[PGNN_TRAIN_MAX_FC];
          for (size_t i = 0; i < s.b.size(); ++i) {
            fs[i] = &s.b[i];
            rf[i] = i + 1 < s.b.size() ? 0 : s.b[i].ref.n_out;
          }
          rc = fc_chain_fwd(c, fs, rf, (int)s.b.size(), g.agg, wa, K, h, ld_h,
                            g.uact);
          if (rc) return rc;
        }
      }
      h = g.uact[s.b.size() - 1];
      ld_h = pad16(s.b.back().ref.n_out);
      k_h = K;
    }
  }
  // prediction heads (gnn.py:133-163)
  const int64_t K = b.n_vertices[b.n_levels];
  const int nc = t.m.num_classes, L = t.m.box_len;
  HeadsSaved &hs = sv.heads;
  sv.h_final = h;
  sv.ld_h_final = ld_h;
  sv.k_final = K;
  PGNN_REQUIRE(c.dry || (h != nullptr && k_h == K), PGNN_E_INVALID,
               "trainer: heads need the last level's vertex features");
  if (!t.groups.empty()) {
    PGNN_REQUIRE(t.groups.size() <= 4, PGNN_E_UNSUPPORTED,
                 "trainer: too many head groups");
    for (size_t gi = 0; gi < t.groups.size(); ++gi) {
      HeadGroup &g = t.groups[gi];
      hs.y1[gi] = c.ws.f(K, pad16(g.f[0].ref.n_out));
      hs.y2[gi] = c.ws.f(K, pad16(g.f[1].ref.n_out));
      hs.y3[gi] = c.ws.f(K, pad16(g.f[2].ref.n_out));
    }
    hs.logits = c.ws.f(K, pad16(nc));
    hs.pred = c.ws.f(K, (int64_t)nc * L);
    if (!c.dry) {
      for (size_t gi = 0; gi < t.groups.size(); ++gi) {
        HeadGroup &g = t.groups[gi];
        {  // the group's three layers (ReLU | ReLU from `base` | linear)
          const FcDev *fs[3] = {&g.f[0], &g.f[1], &g.f[2]};
          const int rf[3] = {0, g.base, g.f[2].ref.n_out};
          float *outs[3] = {hs.y1[gi], hs.y2[gi], hs.y3[gi]};
          rc = fc_chain_fwd(c, fs, rf, 3, h, ld_h, K, nullptr, 0, outs);
          if (rc) return rc;
        }
        if (K > 0) {
          const int ld3 = pad16(g.f[2].ref.n_out);
          bool run = true;  // lids of a group are consecutive
          for (size_t i = 1; i < g.lids.size(); ++i)
            run = run && g.lids[i] == g.lids[0] + (int)i;
          if (run) {
            const int ncp = pad16(nc);
            hipLaunchKernelGGL(
                pred_slices_kernel,
                dim3(blocks_for(K * (L * (int64_t)g.lids.size() + ncp))),
                dim3(256), 0, c.stream, hs.y3[gi], (int64_t)ld3, K, L, nc,
                g.lids.empty() ? 0 : g.lids[0], (int)g.lids.size(), g.base,
                hs.pred, g.has_cls ? hs.logits : nullptr, ncp);
          } else {
            if (g.has_cls)
              hipLaunchKernelGGL(block_copy_kernel<false>,
                                 dim3(blocks_for(K * pad16(nc))), dim3(256), 0,
                                 c.stream, hs.logits, (int64_t)pad16(nc), 0,
                                 hs.y3[gi], (int64_t)ld3, 0, K, pad16(nc));
            for (size_t i = 0; i < g.lids.size(); ++i)
              hipLaunchKernelGGL(pred_slice_kernel, dim3(blocks_for(K * L)),
                                 dim3(256), 0, c.stream,
                                 hs.y3[gi] + g.base + 8 * (int)i, (int64_t)ld3, K,
                                 L, nc, g.lids[i], hs.pred);
          }
        }
      }
      PGNN_HIP(hipGetLastError());
    }
    return 0;
  }
  hs.c1 = c.ws.f(K, pad16(t.cls[0].ref.n_out));
  hs.logits = c.ws.f(K, pad16(nc));
  for (int j = 0; j < nc; ++j) {
    hs.l1[j] = c.ws.f(K, pad16(t.loc[3 * j].ref.n_out));
    hs.l2[j] = c.ws.f(K, pad16(t.loc[3 * j + 1].ref.n_out));
    hs.l3[j] = c.ws.f(K, pad16(L));
  }
  hs.pred = c.ws.f(K, (int64_t)nc * L);
  if (!c.dry) {
    rc = fc_fwd(c, t.cls[0], h, ld_h, K, true, nullptr, 0, hs.c1);
    if (rc) return rc;
    rc = fc_fwd(c, t.cls[1], hs.c1, pad16(t.cls[0].ref.n_out), K, false, nullptr,
                0, hs.logits);
    if (rc) return rc;
    for (int j = 0; j < nc; ++j) {
      rc = fc_fwd(c, t.loc[3 * j], h, ld_h, K, true, nullptr, 0, hs.l1[j]);
      if (rc) return rc;
      rc = fc_fwd(c, t.loc[3 * j + 1], hs.l1[j], pad16(t.loc[3 * j].ref.n_out), K,
                  true, nullptr, 0, hs.l2[j]);
      if (rc) return rc;
      rc = fc_fwd(c, t.loc[3 * j + 2], hs.l2[j],
                  pad16(t.loc[3 * j + 1].ref.n_out), K, false, nullptr, 0,
                  hs.l3[j]);
      if (rc) return rc;
      if (K > 0)
        hipLaunchKernelGGL(pred_slice_kernel, dim3(blocks_for(K * L)), dim3(256), 0,
                           c.stream, hs.l3[j], (int64_t)pad16(L), K, L, nc, j,
                           hs.pred);
    }
    PGNN_HIP(hipGetLastError());
  }
  return 0;
}

// dy3 of a fused group: [dlogits | 0.. | dpred of the group's box heads | 0..]
__global__ void fused_dy3_kernel(const float *__restrict__ dlogits,
                                 const float *__restrict__ dpred, int64_t rows,
                                 int nc, int L, int has_cls, int base, int n_loc,
                                 int lid0, float *__restrict__ dy, int ld) {
  const int64_t total = rows * ld;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / ld;
    const int k = (int)(idx - r * ld);
    float v = 0.0f;
    if (has_cls && k < nc) {
      v = dlogits[r * nc + k];
    } else if (k >= base && k < base + 8 * n_loc) {
      const int i = (k - base) >> 3, q = (k - base) & 7;
      if (q < L) v = dpred[(r * nc + lid0 + i) * L + q];
    }
    dy[idx] = v;
  }
}

// dy[r, c] = 0 where y[r, c] <= 0, for columns c0 <= c < c1 only
__global__ void relu_mask_cols_kernel(float *__restrict__ dy,
                                      const float *__restrict__ y, int64_t rows,
                                      int ld, int c0, int c1) {
  const int w = c1 - c0;
  const int64_t total = rows * w;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / w;
    const int c = c0 + (int)(idx - r * w);
    if (!(y[r * ld + c] > 0.0f)) dy[r * ld + c] = 0.0f;
  }
}

// ---- backward --------------------------------------------------------------------------
int backward_impl(Ctx &c, Saved &sv, const float *dlogits, const float *dpred) {
  Trainer &t = c.t;
  const pgnn_train_batch &b = c.b;
  const int64_t K = sv.k_final;
  const int nc = t.m.num_classes, L = t.m.box_len;
  const int hw = sv.ld_h_final;
  const HeadsSaved &hs = sv.heads;
  int rc = 0;
  // gradient w.r.t. the output of stage i: dhs[i + 1] (dhs[n] = w.r.t. the
  // input of the heads).  One buffer per boundary: each is also the dY of a
  // deferred weight gradient and must survive until the end of the backward.
  const int n_stages = (int)t.stages.size();
  std::vector<float *> dhs((size_t)n_stages + 1);
  for (int i = 0; i <= n_stages; ++i) dhs[(size_t)i] = c.ws.f(K, hw);
  float *dh = dhs[(size_t)n_stages];
  float *dxh = c.ws.f(K, hw);
  const int cw = t.cls[0].ref.k_in;
  if (!t.groups.empty()) {
    // fused heads: three layers per group
    int w1 = 16, w2 = 16, w3 = 16;
    for (const HeadGroup &g : t.groups) {
      w1 = pad16(g.f[0].ref.n_out) > w1 ? pad16(g.f[0].ref.n_out) : w1;
      w2 = pad16(g.f[1].ref.n_out) > w2 ? pad16(g.f[1].ref.n_out) : w2;
      w3 = pad16(g.f[2].ref.n_out) > w3 ? pad16(g.f[2].ref.n_out) : w3;
    }
    // (one set per group: the buffers are the dY of deferred weight gradients)
    std::vector<float *> dy3s, dy2s, dy1s;
    for (size_t gi = 0; gi < t.groups.size(); ++gi) {
      dy3s.push_back(c.ws.f(K, w3));
      dy2s.push_back(c.ws.f(K, w2));
      dy1s.push_back(c.ws.f(K, w1));
    }
    if (!c.dry && K > 0) {
      for (size_t gi = 0; gi < t.groups.size(); ++gi) {
        HeadGroup &g = t.groups[gi];
        float *dy3 = dy3s[gi], *dy2 = dy2s[gi], *dy1 = dy1s[gi];
        const int ld1 = pad16(g.f[0].ref.n_out), ld2 = pad16(g.f[1].ref.n_out),
                  ld3 = pad16(g.f[2].ref.n_out);
        hipLaunchKernelGGL(fused_dy3_kernel, dim3(blocks_for(K * ld3)), dim3(256),
                           0, c.stream, dlogits, dpred, K, nc, L,
                           g.has_cls ? 1 : 0, g.base, (int)g.lids.size(),
                           g.lids.empty() ? 0 : g.lids[0], dy3, ld3);
        // the fused gradients are overwritten (accumulate = false) and handed
        // to the flat buffer by the unpack jobs at the end
        rc = fc_bwd(c, sv, g.f[2], sv.heads.y2[gi], ld2, nullptr, dy3, K, false,
                    dy2, false, true);
        if (rc) return rc;
        // second layer: ReLU on the box-head columns only (logits are linear)
        if (g.f[1].ref.n_out > g.base)
          hipLaunchKernelGGL(relu_mask_cols_kernel,
                             dim3(blocks_for(K * (g.f[1].ref.n_out - g.base))),
                             dim3(256), 0, c.stream, dy2, sv.heads.y2[gi], K, ld2,
                             g.base, g.f[1].ref.n_out);
        {  // layers 2 and 1 in one launch (dy1 leaves masked by y1 > 0: the
           // first layer's ReluGrad)
          const FcDev *fs[2] = {&g.f[0], &g.f[1]};
          const float *xin[2] = {sv.h_final, sv.heads.y1[gi]};
          const int64_t ldx[2] = {hw, ld1};
          float *dxs[2] = {gi == 0 ? dh : dxh, dy1};
          const bool gate[2] = {false, true};
          rc = fc_chain_bwd(c, sv, fs, 2, xin, ldx, dy2, K, dxs, gate, false);
          if (rc) return rc;
        }
        if (gi > 0)
          hipLaunchKernelGGL(block_copy_kernel<true>, dim3(blocks_for(K * cw)),
                             dim3(256), 0, c.stream, dh, (int64_t)hw, 0, dxh,
                             (int64_t)pad16(cw), 0, K, cw);
      }
      // (the unpack of the fused gradients follows the deferred weight
      // gradients at the end of the backward)
    }
  } else {
  const int w64 = pad16(t.cls[0].ref.n_out);
  int wh = w64;  // widest hidden layer of the heads
  for (const FcDev &f : t.loc)
    if (pad16(f.ref.n_out) > wh) wh = pad16(f.ref.n_out);
  float *dy = c.ws.f(K, pad16(nc > L ? nc : L));
  float *d1 = c.ws.f(K, wh), *d2 = c.ws.f(K, wh);
  if (!c.dry && K > 0) {
    PGNN_HIP(hipMemsetAsync(dh, 0, (size_t)K * hw * 4, c.stream));
    // class head
    PGNN_HIP(hipMemsetAsync(dy, 0, (size_t)K * pad16(nc) * 4, c.stream));
    hipLaunchKernelGGL(block_copy_kernel<false>, dim3(blocks_for(K * nc)),
                       dim3(256), 0, c.stream, dy, (int64_t)pad16(nc), 0, dlogits,
                       (int64_t)nc, 0, K, nc);
    rc = fc_bwd(c, sv, t.cls[1], hs.c1, w64, hs.logits, dy, K, false, d1, true,
                false, true);
    if (rc) return rc;
    rc = fc_bwd(c, sv, t.cls[0], sv.h_final, hw, hs.c1, d1, K, false, dxh);
    if (rc) return rc;
    hipLaunchKernelGGL(block_copy_kernel<true>, dim3(blocks_for(K * cw)),
                       dim3(256), 0, c.stream, dh, (int64_t)hw, 0, dxh,
                       (int64_t)pad16(cw), 0, K, cw);
    for (int j = 0; j < nc; ++j) {
      hipLaunchKernelGGL(dpred_slice_kernel, dim3(blocks_for(K * pad16(L))),
                         dim3(256), 0, c.stream, dpred, K, L, nc, j, dy,
                         pad16(L));
      rc = fc_bwd(c, sv, t.loc[3 * j + 2], hs.l2[j],
                  pad16(t.loc[3 * j + 1].ref.n_out), hs.l3[j], dy, K, false, d2,
                  true, false, true);
      if (rc) return rc;
      rc = fc_bwd(c, sv, t.loc[3 * j + 1], hs.l1[j],
                  pad16(t.loc[3 * j].ref.n_out), hs.l2[j], d2, K, false, d1, true,
                  false, true);
      if (rc) return rc;
      rc = fc_bwd(c, sv, t.loc[3 * j], sv.h_final, hw, hs.l1[j], d1, K, false, dxh);
      if (rc) return rc;
      hipLaunchKernelGGL(block_copy_kernel<true>, dim3(blocks_for(K * cw)),
                         dim3(256), 0, c.stream, dh, (int64_t)hw, 0, dxh,
                         (int64_t)pad16(cw), 0, K, cw);
    }
  }
  }
  // stages in reverse
  for (int si = (int)t.stages.size() - 1; si >= 0; --si) {
    StageDev &s = t.stages[si];
    const int lvl = s.level;
    const int64_t E = b.n_edges[lvl], Ks = b.n_vertices[lvl + 1];
    // gradient w.r.t. this stage's output / input
    float *dh = dhs[(size_t)si + 1], *dh2 = dhs[(size_t)si];
    if (s.kind == 1) {
      GnnSaved &g = sv.gnn[si];
      const FcDev &w1 = s.a[0];
      const int cc = w1.ref.k_in - 3;
      const int wq = pad16(w1.ref.n_out);
      const int wa = pad16(s.a.back().ref.n_out);
      const int ld_h = hw;
      // K-row buffers first: they are the dY of deferred weight gradients and
      // stay until the end of the backward; the E-row temporaries behind the
      // mark are released with the stage
      float *du[PGNN_TRAIN_MAX_FC + 1];
      for (size_t i = 0; i < s.b.size(); ++i)
        du[i] = c.ws.f(Ks, pad16(s.b[i].ref.k_in));
      float *dp = c.ws.f(2 * Ks, wq);  // dP | dQ back to back: one fill
      float *dq = dp ? dp + Ks * wq : nullptr;
      float *dhx = c.ws.f(Ks, pad16(cc + 3));
      float *dxo = c.ws.f(Ks, 16);
      float *doff[PGNN_TRAIN_MAX_FC];
      for (size_t i = 0; i < s.c.size(); ++i)
        doff[i] = c.ws.f(Ks, pad16(s.c[i].ref.k_in));
      const size_t mark = c.ws.off;
      float *ge[PGNN_TRAIN_MAX_FC] = {nullptr};
      if (!(s.a.back().want_wt && s.a.size() == 2))
        for (size_t i = 0; i + 1 < s.a.size(); ++i)
          ge[i] = c.ws.f(E, pad16(s.a[i + 1].ref.k_in));  // grad w.r.t. eact[i]
      float *gz = nullptr;  // dense fallback: grad w.r.t. the last edge layer
      if (!s.a.back().want_wt) gz = c.ws.f(E, wa);
      int32_t *ties = nullptr;
      if (!s.a.back().want_wt) ties = c.ws.i32(Ks * wa > 0 ? Ks * wa : 1);
      if (!c.dry && Ks > 0) {
        // (residual branch, gnn.py:372: dh_in = dh + what the edge input and
        // the offset MLP send back -- summed in one pass at the stage's end)
        // update MLP, last layer linear
        // (one launch; the ReluGrad of layer i - 1 leaves with du[i])
        {
          const FcDev *fs[PGNN_TRAIN_MAX_FC];
          const float *xin[PGNN_TRAIN_MAX_FC];
          int64_t ldx[PGNN_TRAIN_MAX_FC];
          bool gate[PGNN_TRAIN_MAX_FC];
          for (size_t i = 0; i < s.b.size(); ++i) {
            fs[i] = &s.b[i];
            xin[i] = i == 0 ? g.agg : g.uact[i - 1];
            ldx[i] = i == 0 ? wa : pad16(s.b[i - 1].ref.n_out);
            gate[i] = i > 0;
          }
          rc = fc_chain_bwd(c, sv, fs, (int)s.b.size(), xin, ldx, dh, Ks, du, gate);
          if (rc) return rc;
        }
        const float *dagg = du[0];  // [Ks, pad(k_in of b[0])] = [Ks, wa]
        const int na = (int)s.a.size();
        float *gcur = nullptr;  // grad w.r.t. eact[i], ReLU-masked
        int from;
        bool scattered = false;
        if (s.a.back().want_wt && na == 2) {
          // last edge layer + scatter-max + the gather's adjoint in one
          // routing pass: dP / dQ directly, dH1 is never written
          rc = pgnn_edge_segmax_fc_bwd_f32(
              g.eact[1], wa, b.edges[lvl], g.dst, E, s.a[1].ref.n_out,
              (int32_t)Ks, g.agg, wa, dagg, wa, c.t.h1 ? g.eact[0] : nullptr,
              wq, g.p, g.q,
              s.a[1].ref.k_in, s.a[1].wt, pad16(s.a[1].ref.k_in), dp, dq, wq,
              s.a[1].gw, s.a[1].gb, sv.scratch, sv.scratch_bytes, c.stream);
          if (rc) return rc;
          scattered = true;
          from = 0;
        } else if (s.a.back().want_wt) {
          rc = pgnn_segmax_fc_bwd_f32(
              g.eact[na - 1], wa, g.dst, E, s.a.back().ref.n_out, (int32_t)Ks,
              g.agg, wa, dagg, wa, g.eact[na - 2], pad16(s.a.back().ref.k_in),
              s.a.back().ref.k_in, s.a.back().wt, pad16(s.a.back().ref.k_in),
              ge[na - 2], pad16(s.a.back().ref.k_in), pad16(s.a.back().ref.k_in),
              1, s.a.back().gw, s.a.back().gb, sv.scratch, sv.scratch_bytes,
              c.stream);
          if (rc) return rc;
          gcur = ge[na - 2];
          from = na - 2;
        } else {
          rc = pgnn_scatter_max_bwd_f32(g.eact[na - 1], wa, g.dst, E, wa,
                                        (int32_t)Ks, g.agg, wa, dagg, wa, ties,
                                        gz, wa, 1, c.stream);
          if (rc) return rc;
          gcur = gz;
          from = na - 1;
        }
        for (int i = from; i >= 1; --i) {  // dense middle edge layers
          rc = fc_bwd(c, sv, s.a[i], g.eact[i - 1], pad16(s.a[i].ref.k_in),
                      g.eact[i], gcur, E, false, ge[i - 1], true, false, true);
          if (rc) return rc;
          gcur = ge[i - 1];
        }
        if (!scattered) {
          rc = pgnn_edge_hidden_bwd(gcur, wq, b.edges[lvl], E, Ks, dp, dq,
                                    c.stream);
          if (rc) return rc;
        }
        // P = [h, x] W1 + b1
        rc = fc_bwd(c, sv, w1, g.hx, pad16(cc + 3), nullptr, dp, Ks, false, dhx,
                    true, true);
        if (rc) return rc;
        // Q = x' Wx, Wx = rows cc..cc+2 of W1 (the minus sign is in dq)
        // (it adds into rows of the same dW as the deferred job of w1 above,
        // and jobs of one batch must not share outputs: the second batch)
        {
          pgnn_wgrad_job j = {};
          j.X = g.xo;
          j.ld_x = 3;
          j.dZ = dq;
          j.ld_dz = wq;
          j.n_rows = Ks;
          j.dW = w1.gw + (int64_t)cc * w1.ref.n_out;
          j.db = nullptr;
          j.k_in = 3;
          j.n_out = w1.ref.n_out;
          j.accumulate = 1;
          c.wjobs2.push_back(j);
        }
        const float *d_off = nullptr;  // grad w.r.t. h through the offset MLP
        if (!s.c.empty()) {
          // dx' = dQ Wx^T, then the offset MLP's backward: one launch
          const PreDx pre = {s.wx_packed_t, w1.ref.n_out, 3, dq, wq};
          {
            const FcDev *fs[PGNN_TRAIN_MAX_FC];
            const float *xin[PGNN_TRAIN_MAX_FC];
            int64_t ldx[PGNN_TRAIN_MAX_FC];
            bool gate[PGNN_TRAIN_MAX_FC];
            for (size_t i = 0; i < s.c.size(); ++i) {
              fs[i] = &s.c[i];
              xin[i] = i == 0 ? g.h_in : g.off_act[i - 1];
              ldx[i] = i == 0 ? ld_h : pad16(s.c[i - 1].ref.n_out);
              gate[i] = i > 0;
            }
            rc = fc_chain_bwd(c, sv, fs, (int)s.c.size(), xin, ldx, dxo, Ks, doff,
                              gate, true, &pre);
            if (rc) return rc;
          }
          d_off = doff[0];
        }
        hipLaunchKernelGGL(stage_input_grad_kernel, dim3(blocks_for(Ks * ld_h)),
                           dim3(256), 0, c.stream, dh, ld_h, dhx, pad16(cc + 3),
                           d_off, s.c.empty() ? 0 : pad16(s.c[0].ref.k_in), Ks, cc,
                           dh2);
      }
      c.ws.off = mark;
    } else {
      PoolSaved &p = sv.pool[si];
      const int wa = pad16(s.a.back().ref.n_out);
      float *dob[PGNN_TRAIN_MAX_FC];  // K-row: kept (deferred weight gradients)
      for (size_t i = 0; i < s.b.size(); ++i)
        dob[i] = c.ws.f(Ks, pad16(s.b[i].ref.k_in));
      const size_t mark = c.ws.off;
      float *ga[PGNN_TRAIN_MAX_FC];  // grad w.r.t. act[i] (input of layer i+1)
      for (size_t i = 0; i + 1 < s.a.size(); ++i)
        ga[i] = c.ws.f(E, pad16(s.a[i + 1].ref.k_in));
      float *gz = nullptr;
      int32_t *ties = nullptr;
      if (!s.a.back().want_wt) {
        gz = c.ws.f(E, wa);
        ties = c.ws.i32(Ks * wa > 0 ? Ks * wa : 1);
      }
      if (!c.dry && Ks > 0) {
        // every layer is a ReLU layer: the last one's mask is applied to dh
        // here, the others' leave with the dX of the layer above (one launch)
        rc = pgnn_relu_mask_mul(dh, p.oact[s.b.size() - 1],
                                Ks * pad16(s.b.back().ref.n_out), c.stream);
        if (rc) return rc;
        {
          const FcDev *fs[PGNN_TRAIN_MAX_FC];
          const float *xin[PGNN_TRAIN_MAX_FC];
          int64_t ldx[PGNN_TRAIN_MAX_FC];
          bool gate[PGNN_TRAIN_MAX_FC];
          for (size_t i = 0; i < s.b.size(); ++i) {
            fs[i] = &s.b[i];
            xin[i] = i == 0 ? p.agg : p.oact[i - 1];
            ldx[i] = i == 0 ? wa : pad16(s.b[i - 1].ref.n_out);
            gate[i] = i > 0;
          }
          rc = fc_chain_bwd(c, sv, fs, (int)s.b.size(), xin, ldx, dh, Ks, dob, gate);
          if (rc) return rc;
        }
        float *d = dob[0];
        const int na = (int)s.a.size();
        float *gcur;
        int from;
        if (s.a.back().want_wt) {
          rc = pgnn_segmax_fc_bwd_f32(
              p.act[na - 1], wa, p.dst, E, s.a.back().ref.n_out, (int32_t)Ks,
              p.agg, wa, d, wa, p.act[na - 2], pad16(s.a.back().ref.k_in),
              s.a.back().ref.k_in, s.a.back().wt, pad16(s.a.back().ref.k_in),
              ga[na - 2], pad16(s.a.back().ref.k_in), pad16(s.a.back().ref.k_in),
              1, s.a.back().gw, s.a.back().gb, sv.scratch, sv.scratch_bytes,
              c.stream);
          if (rc) return rc;
          gcur = ga[na - 2];
          from = na - 2;
        } else {
          rc = pgnn_scatter_max_bwd_f32(p.act[na - 1], wa, p.dst, E, wa,
                                        (int32_t)Ks, p.agg, wa, d, wa, ties, gz,
                                        wa, 1, c.stream);
          if (rc) return rc;
          gcur = gz;
          from = na - 1;
        }
        // the shipped car chain's three narrow layers (feat -> 32 -> 64 -> 128
        // below the sparse 128 -> 300 layer): one fused pass over the E rows
        const bool narrow =
            s.a.back().want_wt && na == 4 && from == 2 &&
            s.a[0].ref.k_in <= 15 && s.a[0].ref.n_out == 32 &&
            s.a[1].ref.k_in == 32 && s.a[1].ref.n_out == 64 &&
            s.a[2].ref.k_in == 64 && s.a[2].ref.n_out == 128 &&
            s.a[1].packed_t && s.a[2].packed_t &&
            sv.scratch_bytes >= pgnn_pool_narrow_bwd_workspace_bytes(E);
        if (narrow) {
          rc = pgnn_pool_narrow_bwd_f32(
              p.feat, p.act[0], p.act[1], gcur, E, s.a[2].packed_t,
              s.a[1].packed_t, s.a[0].ref.k_in, s.a[0].gw, s.a[0].gb, s.a[1].gw,
              s.a[1].gb, s.a[2].gw, s.a[2].gb, 1, sv.scratch, sv.scratch_bytes,
              c.stream);
          if (rc) return rc;
          from = -1;  // done
        }
        for (int i = from; i >= 0; --i) {
          const float *xin = i == 0 ? p.feat : p.act[i - 1];
          const int64_t ldx = i == 0 ? 16 : pad16(s.a[i - 1].ref.n_out);
          rc = fc_bwd(c, sv, s.a[i], xin, ldx, p.act[i], gcur, E, false,
                      i > 0 ? ga[i - 1] : nullptr, true, false, i > 0);
          if (rc) return rc;
          if (i > 0) gcur = ga[i - 1];
        }
      }
      c.ws.off = mark;
    }
  }
  // every K-row weight gradient of the step, in one launch pair; then the
  // fused heads' gradients go to the flat buffer
  rc = flush_wgrads(c, sv);
  if (rc) return rc;
  if (!c.dry && K > 0 && !t.groups.empty()) {
    rc = pgnn_pack_fc_many(t.images + t.off_jobs_u, t.n_jobs_u, t.total_blocks_u,
                           c.stream);
    if (rc) return rc;
  }
  if (!c.dry) PGNN_HIP(hipGetLastError());
  return 0;
}

int make_fc(const pgnn_train_fc &r, int64_t n_params, FcDev &f) {
  PGNN_REQUIRE(r.k_in > 0 && r.n_out > 0 && r.k_in <= 4096 && r.n_out <= 4096 &&
                   r.w_off >= 0 && r.b_off >= 0 &&
                   r.w_off + (int64_t)r.k_in * r.n_out <= n_params &&
                   r.b_off + r.n_out <= n_params,
               PGNN_E_INVALID, "trainer: layer outside the flat parameter buffer");
  f.ref = r;
  return 0;
}

}  // namespace

extern "C" int pgnn_trainer_create(const pgnn_train_model *m, void **handle) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(m && handle, PGNN_E_INVALID, "trainer_create: null argument");
  PGNN_REQUIRE(m->n_stages >= 1 && m->n_stages <= PGNN_TRAIN_MAX_STAGES &&
                   m->num_classes >= 1 &&
                   m->num_classes <= PGNN_TRAIN_MAX_CLASSES && m->box_len >= 1 &&
                   m->box_len <= 16 && m->n_params > 0,
               PGNN_E_INVALID, "trainer_create: bad model header");
  Trainer *t = new Trainer();
  t->m = *m;
  int rc = 0;
  for (int si = 0; si < m->n_stages && !rc; ++si) {
    const pgnn_train_stage &s = m->stages[si];
    StageDev d;
    d.kind = s.kind;
    d.level = s.graph_level;
    if (!((s.kind == 0 || s.kind == 1) && s.n_a >= 1 && s.n_a <= PGNN_TRAIN_MAX_FC &&
          s.n_b >= 1 && s.n_b <= PGNN_TRAIN_MAX_FC && s.n_c >= 0 &&
          s.n_c <= PGNN_TRAIN_MAX_FC && (s.kind == 1 || s.n_c == 0) &&
          (s.kind == 0 || s.n_a >= 2))) {
      rc = fail(PGNN_E_INVALID, "trainer_create: bad stage description");
      break;
    }
    d.a.resize(s.n_a);
    d.b.resize(s.n_b);
    d.c.resize(s.n_c);
    for (int i = 0; i < s.n_a && !rc; ++i) rc = make_fc(s.a[i], m->n_params, d.a[i]);
    for (int i = 0; i < s.n_b && !rc; ++i) rc = make_fc(s.b[i], m->n_params, d.b[i]);
    for (int i = 0; i < s.n_c && !rc; ++i) rc = make_fc(s.c[i], m->n_params, d.c[i]);
    // the last per-edge layer feeds the scatter-max: sparse adjoint when it
    // has a materialised input activation (n_a >= 2) and fits the kernel
    if (!rc && s.n_a >= 2 && d.a.back().ref.n_out <= 512 &&
        d.a.back().ref.k_in <= 512)
      d.a.back().want_wt = true;
    t->stages.push_back(d);
  }
  for (int i = 0; i < 2 && !rc; ++i) rc = make_fc(m->cls[i], m->n_params, t->cls[i]);
  t->loc.resize(3 * (size_t)m->num_classes);
  for (int j = 0; j < m->num_classes && !rc; ++j)
    for (int i = 0; i < 3 && !rc; ++i)
      rc = make_fc(m->loc[j][i], m->n_params, t->loc[3 * j + i]);
  // Shapes the orchestration below ASSUMES (every shipped config has them; any
  // other layer_configs used to give silent garbage): a pooling stage only in
  // front, one feature width from the first stage's output to the heads'
  // inputs (the residual adds and the one dh buffer rely on it).
  if (!rc) {
    const int hw = t->cls[0].ref.k_in;
    bool ok = true;
    for (size_t si = 0; si < t->stages.size(); ++si) {
      const StageDev &d = t->stages[si];
      if (d.kind == 0 && si > 0) ok = false;             // pooling at si > 0
      if (d.b.back().ref.n_out != hw) ok = false;        // stage output width
      if (d.kind == 1 && d.a[0].ref.k_in != hw + 3) ok = false;
    }
    for (int j = 0; j < m->num_classes; ++j)
      if (t->loc[3 * (size_t)j].ref.k_in != hw) ok = false;
    if (!ok)
      rc = fail(PGNN_E_UNSUPPORTED,
                "trainer_create: the native step needs a pooling stage only "
                "in front and one feature width from the first stage to the "
                "heads (use the Python-driven step: Trainer.native = False)");
  }
  if (rc) {
    delete t;
    return rc;
  }
  if (!plan_head_groups(*t)) t->groups.clear();  // heads one by one
  t->images_bytes = layout_images(*t);
  *handle = t;
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_trainer_destroy(void *handle) {
  PGNN_GUARD_BEGIN
  delete (Trainer *)handle;
  return 0;
  PGNN_GUARD_END
}

extern "C" size_t pgnn_trainer_images_bytes(void *handle) {
  return handle ? ((Trainer *)handle)->images_bytes : 0;
}

extern "C" int pgnn_trainer_bind(void *handle, float *params, float *grads,
                                 void *images, size_t images_bytes,
                                 void *stream_) {
  PGNN_GUARD_BEGIN
  Trainer *t = (Trainer *)handle;
  PGNN_REQUIRE(t && params && grads && images, PGNN_E_INVALID,
               "trainer_bind: null argument");
  PGNN_REQUIRE(images_bytes >= t->images_bytes, PGNN_E_WORKSPACE,
               "trainer_bind: images buffer too small");
  t->params = params;
  t->grads = grads;
  t->images = (char *)images;
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<PackJobRec> jobs, jobs_a, jobs_u;
  auto add_to = [](std::vector<PackJobRec> &v, int &first, const float *w,
                   const float *b, float *dst, int k_in, int n_out, int kind,
                   size_t elems, int ld) {
    PackJobRec j;
    j.w = w;
    j.b = b;
    j.dst = dst;
    j.k_in = k_in;
    j.n_out = n_out;
    j.kind = kind;
    j.first_block = first;
    j.ld = ld;
    j.reserved = 0;
    first += (int)((elems + 255) / 256);
    v.push_back(j);
  };
  int first = 0, first_a = 0, first_u = 0;
  auto add = [&](const float *w, const float *b, float *dst, int k_in, int n_out,
                 int kind, size_t elems) {
    add_to(jobs, first, w, b, dst, k_in, n_out, kind, elems, 0);
  };
  auto images_of = [&](FcDev &f) {  // f.w / f.b are set
    f.packed = (float *)(t->images + f.off_packed);
    f.packed_t = (float *)(t->images + f.off_packed_t);
    add(f.w, f.b, f.packed, f.ref.k_in, f.ref.n_out, 0,
        packed_floats(f.ref.k_in, f.ref.n_out));
    add(f.w, nullptr, f.packed_t, f.ref.k_in, f.ref.n_out, 1,
        packed_floats(f.ref.n_out, f.ref.k_in));
    if (f.want_wt) {
      f.wt = (float *)(t->images + f.off_wt);
      add(f.w, nullptr, f.wt, f.ref.k_in, f.ref.n_out, 2,
          (size_t)f.ref.n_out * pad16(f.ref.k_in));
    }
  };
  auto bind_flat = [&](FcDev &f) {
    f.w = params + f.ref.w_off;
    f.b = params + f.ref.b_off;
    f.gw = grads + f.ref.w_off;
    f.gb = grads + f.ref.b_off;
  };
  auto bind_fc = [&](FcDev &f) {
    bind_flat(f);
    images_of(f);
  };
  for (StageDev &s : t->stages) {
    for (FcDev &f : s.a) bind_fc(f);
    for (FcDev &f : s.b) bind_fc(f);
    for (FcDev &f : s.c) bind_fc(f);
    if (s.kind == 1) {
      const FcDev &w1 = s.a[0];
      const int cc = w1.ref.k_in - 3, n_out = w1.ref.n_out;
      const float *wx = w1.w + (int64_t)cc * n_out;
      s.wx = (float *)(t->images + s.off_wx);
      s.wx_packed_t = (float *)(t->images + s.off_wx_packed_t);
      add(wx, nullptr, s.wx, 3, n_out, 3, (size_t)3 * pad16(n_out));
      add(wx, nullptr, s.wx_packed_t, 3, n_out, 1, packed_floats(n_out, 3));
    }
  }
  for (FcDev &f : t->cls) bind_flat(f);
  for (FcDev &f : t->loc) bind_flat(f);
  if (t->groups.empty()) {
    for (FcDev &f : t->cls) images_of(f);
    for (FcDev &f : t->loc) images_of(f);
  } else {
    // fused matrices: zero once (the blocks outside the parameter blocks
    // stay zero for good), identity pass-through of the logits once
    PGNN_HIP(hipMemsetAsync(t->images + t->fused_lo, 0, t->fused_hi - t->fused_lo,
                            stream));
    std::vector<PackJobRec> once;
    int first_o = 0;
    const int nc = t->m.num_classes, L = t->m.box_len;
    const int hw = t->cls[0].ref.n_out;
    for (HeadGroup &g : t->groups) {
      for (int i = 0; i < 3; ++i) {
        g.f[i].w = (float *)(t->images + g.off_w[i]);
        g.f[i].b = (float *)(t->images + g.off_b[i]);
        g.f[i].gw = (float *)(t->images + g.off_gw[i]);
        g.f[i].gb = (float *)(t->images + g.off_gb[i]);
      }
      const int n1 = g.f[0].ref.n_out, n2 = g.f[1].ref.n_out, n3 = g.f[2].ref.n_out;
      // one parameter block <-> its place in a fused matrix (and back for the
      // gradient): rows x cols at (r0, c0) of a matrix with row stride ld
      auto block = [&](const FcDev &src, int layer, int r0, int c0) {
        const FcDev &dst = g.f[layer];
        const int ld = dst.ref.n_out;
        const int rows = src.ref.k_in, cols = src.ref.n_out;
        add_to(jobs_a, first_a, src.w, nullptr, dst.w + (int64_t)r0 * ld + c0, rows,
               cols, 4, (size_t)rows * cols, ld);
        add_to(jobs_a, first_a, src.b, nullptr, dst.b + c0, 1, cols, 4, (size_t)cols,
               ld);
        add_to(jobs_u, first_u, dst.gw + (int64_t)r0 * ld + c0, nullptr, src.gw, rows,
               cols, 5, (size_t)rows * cols, ld);
        add_to(jobs_u, first_u, dst.gb + c0, nullptr, src.gb, 1, cols, 5,
               (size_t)cols, ld);
      };
      int slot = 0;
      if (g.has_cls) {
        block(t->cls[0], 0, 0, 0);
        block(t->cls[1], 1, 0, 0);
        // logits ride through the third layer
        add_to(once, first_o, nullptr, nullptr, g.f[2].w, nc, nc, 6, (size_t)nc, n3);
        slot = 1;
      }
      for (size_t i = 0; i < g.lids.size(); ++i) {
        const int j = g.lids[i], sl = slot + (int)i;
        block(t->loc[3 * j], 0, 0, hw * sl);
        block(t->loc[3 * j + 1], 1, hw * sl, g.base + hw * (int)i);
        block(t->loc[3 * j + 2], 2, g.base + hw * (int)i, g.base + 8 * (int)i);
      }
      (void)n1;
      (void)n2;
      (void)L;
      for (int i = 0; i < 3; ++i) images_of(g.f[i]);
    }
    if (!once.empty()) {
      // run the one-off jobs from the (not yet used) assemble table slot
      PGNN_REQUIRE(once.size() <= (size_t)t->n_jobs_a + 1, PGNN_E_INVALID,
                   "trainer_bind: one-off job table too large");
      PGNN_HIP(hipMemcpyAsync(t->images + t->off_jobs_a, once.data(),
                              once.size() * sizeof(PackJobRec),
                              hipMemcpyHostToDevice, stream));
      PGNN_HIP(hipStreamSynchronize(stream));
      int rc1 = pgnn_pack_fc_many(t->images + t->off_jobs_a, (int)once.size(),
                                  first_o, stream_);
      if (rc1) return rc1;
      PGNN_HIP(hipStreamSynchronize(stream));
    }
  }
  PGNN_REQUIRE((int)jobs.size() == t->n_jobs && (int)jobs_a.size() == t->n_jobs_a &&
                   (int)jobs_u.size() == t->n_jobs_u,
               PGNN_E_INVALID, "trainer_bind: job count mismatch");
  t->total_blocks = first;
  t->total_blocks_a = first_a;
  t->total_blocks_u = first_u;
  PGNN_HIP(hipMemcpyAsync(t->images + t->off_jobs, jobs.data(),
                          jobs.size() * sizeof(PackJobRec), hipMemcpyHostToDevice,
                          stream));
  if (!jobs_a.empty()) {
    PGNN_HIP(hipMemcpyAsync(t->images + t->off_jobs_a, jobs_a.data(),
                            jobs_a.size() * sizeof(PackJobRec),
                            hipMemcpyHostToDevice, stream));
    PGNN_HIP(hipMemcpyAsync(t->images + t->off_jobs_u, jobs_u.data(),
                            jobs_u.size() * sizeof(PackJobRec),
                            hipMemcpyHostToDevice, stream));
  }
  PGNN_HIP(hipStreamSynchronize(stream));  // the tables are host temporaries
  return pgnn_trainer_repack(handle, stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_trainer_repack(void *handle, void *stream_) {
  PGNN_GUARD_BEGIN
  Trainer *t = (Trainer *)handle;
  PGNN_REQUIRE(t && t->images, PGNN_E_INVALID, "trainer_repack: not bound");
  if (t->n_jobs_a > 0) {  // fused head matrices from their parameter blocks
    int rc = pgnn_pack_fc_many(t->images + t->off_jobs_a, t->n_jobs_a,
                               t->total_blocks_a, stream_);
    if (rc) return rc;
  }
  return pgnn_pack_fc_many(t->images + t->off_jobs, t->n_jobs, t->total_blocks,
                           stream_);
  PGNN_GUARD_END
}

extern "C" size_t pgnn_trainer_workspace_bytes(void *handle,
                                               const pgnn_train_batch *batch) {
  Trainer *t = (Trainer *)handle;
  if (!t || !batch || check_batch(*t, batch)) return 0;
  Bump ws(nullptr, 0);
  Saved sv;
  memset(&sv, 0, sizeof sv);
  Ctx c{*t, *batch, ws, nullptr, true};
  if (forward_impl(c, sv)) return 0;
  if (backward_impl(c, sv, nullptr, nullptr)) return 0;
  return ws.high + 4096;
}

extern "C" int pgnn_trainer_forward(void *handle, const pgnn_train_batch *batch,
                                    void *workspace, size_t workspace_bytes,
                                    const float **logits, int64_t *ld_logits,
                                    const float **pred_box, void *stream_) {
  PGNN_GUARD_BEGIN
  Trainer *t = (Trainer *)handle;
  PGNN_REQUIRE(t && t->images && workspace && logits && ld_logits && pred_box,
               PGNN_E_INVALID, "trainer_forward: null argument / not bound");
  int rc = check_batch(*t, batch);
  if (rc) return rc;
  PGNN_REQUIRE(workspace_bytes >= pgnn_trainer_workspace_bytes(handle, batch),
               PGNN_E_WORKSPACE, "trainer_forward: workspace too small");
  Bump ws(workspace, workspace_bytes);
  Saved sv;
  memset(&sv, 0, sizeof sv);
  t->h1 = g_train_h1 ? 1 : 0;
  Ctx c{*t, *batch, ws, (hipStream_t)stream_, false};
  rc = forward_impl(c, sv);
  if (rc) return rc;
  *logits = sv.heads.logits;
  *ld_logits = pad16(t->m.num_classes);
  *pred_box = sv.heads.pred;
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_trainer_backward(void *handle, const pgnn_train_batch *batch,
                                     void *workspace, size_t workspace_bytes,
                                     const float *dlogits, const float *dpred_box,
                                     void *stream_) {
  PGNN_GUARD_BEGIN
  Trainer *t = (Trainer *)handle;
  PGNN_REQUIRE(t && t->images && workspace && dlogits && dpred_box,
               PGNN_E_INVALID, "trainer_backward: null argument / not bound");
  int rc = check_batch(*t, batch);
  if (rc) return rc;
  PGNN_REQUIRE(workspace_bytes >= pgnn_trainer_workspace_bytes(handle, batch),
               PGNN_E_WORKSPACE, "trainer_backward: workspace too small");
  // re-derive the forward's pointers by replaying its allocation sequence
  Bump ws(workspace, workspace_bytes);
  Saved sv;
  memset(&sv, 0, sizeof sv);
  Ctx dryc{*t, *batch, ws, nullptr, true};
  rc = forward_impl(dryc, sv);
  if (rc) return rc;
  Ctx c{*t, *batch, ws, (hipStream_t)stream_, false};
  return backward_impl(c, sv, dlogits, dpred_box);
  PGNN_GUARD_END
}

extern "C" int pgnn_trainer_backward_sync(void *handle, const pgnn_train_batch *batch,
                                          void *workspace, size_t workspace_bytes,
                                          const float *dlogits, const float *dpred_box,
                                          void *comm, double *sums, int64_t n_sums,
                                          void *stream_) {
  int rc = pgnn_trainer_backward(handle, batch, workspace, workspace_bytes,
                                 dlogits, dpred_box, stream_);
  if (rc || !comm) return rc;
  Trainer *t = (Trainer *)handle;
  return pgnn_allreduce_step(comm, t->grads, t->m.n_params, sums, n_sums, stream_);
}
