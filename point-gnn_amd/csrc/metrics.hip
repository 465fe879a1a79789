// Streaming classification metrics of the training / evaluation loops, the
// step right after `model.loss` (train.py:301-368, eval.py:176-245): per class
// tf.metrics.recall / tf.metrics.precision of argmax(probs) and
// tf.metrics.auc(curve='PR', num_thresholds=200,
// summation_method='careful_interpolation') of probs[:, c].
//
// TensorFlow keeps four float32 counters per threshold and updates them with a
// [T, K] broadcast compare per step.  Here a vertex finds, per class, how many
// thresholds lie strictly below its probability (the thresholds ascend, so that
// one number fixes all T comparisons) and bumps one histogram bin; workgroups
// merge their LDS histograms into int64 global counters.  The T confusion
// counts are suffix sums of the bins, taken when the values are asked for.
// Integer/HBM-latency work on a few thousand rows per step; no MFMA.
#include "pgnn_common.h"

namespace pgnn {
namespace {

constexpr int kMaxThresholds = 1024;

// tf.metrics.auc: [0 - 1e-7] + [(i+1)/(T-1) for i in range(T-2)] + [1 + 1e-7],
// Python doubles turned into a float32 constant.
__device__ __forceinline__ float threshold_at(int i, int n_thresholds) {
  if (i == 0) return (float)(0.0 - 1e-7);
  if (i == n_thresholds - 1) return (float)(1.0 + 1e-7);
  return (float)(((double)i * 1.0) / (double)(n_thresholds - 1));
}

// state: per class c a block of 3 + 2*(T+1) counters:
//   [0] true positives, [1] false positives, [2] false negatives (argmax),
//   [3 .. 3+T]        bin b of the rows with label == c,
//   [4+T .. 4+2T]     bin b of the rows with label != c,
// bin b = number of thresholds t with probs[row, c] > thresholds[t].
__host__ __device__ inline int64_t class_block(int n_thresholds) {
  return 3 + 2 * ((int64_t)n_thresholds + 1);
}

__global__ __launch_bounds__(256) void metrics_update_kernel(
    const float *__restrict__ probs, int64_t ld, const int32_t *__restrict__ labels,
    int64_t n_rows, int32_t n_classes, int32_t n_thresholds,
    unsigned long long *__restrict__ state) {
  extern __shared__ unsigned int hist[];  // n_classes * class_block
  __shared__ float thr[kMaxThresholds];
  const int block = (int)class_block(n_thresholds);
  const int total = n_classes * block;
  for (int i = threadIdx.x; i < total; i += blockDim.x) hist[i] = 0u;
  for (int i = threadIdx.x; i < n_thresholds; i += blockDim.x)
    thr[i] = threshold_at(i, n_thresholds);
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       row < n_rows; row += (int64_t)gridDim.x * blockDim.x) {
    const float *p = probs + row * ld;
    const int label = labels[row];
    // tf.argmax: the first maximum
    int pred = 0;
    float best = p[0];
    for (int c = 1; c < n_classes; ++c) {
      const float v = p[c];
      if (v > best) {
        best = v;
        pred = c;
      }
    }
    for (int c = 0; c < n_classes; ++c) {
      const bool is_c = label == c, said_c = pred == c;
      unsigned int *h = hist + c * block;
      if (is_c && said_c) atomicAdd(h + 0, 1u);
      if (!is_c && said_c) atomicAdd(h + 1, 1u);
      if (is_c && !said_c) atomicAdd(h + 2, 1u);
      // bin = first threshold index that is NOT below v (v > thr[t] fails)
      const float v = p[c];
      int lo = 0, hi = n_thresholds;  // invariant: v > thr[t] for t < lo
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v > thr[mid]) lo = mid + 1; else hi = mid;
      }
      atomicAdd(h + 3 + (is_c ? 0 : n_thresholds + 1) + lo, 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x)
    if (hist[i]) atomicAdd(state + i, (unsigned long long)hist[i]);
}

// One thread per class: float32 arithmetic in TensorFlow's order
// (metrics_impl.py: recall / precision = where(denom > 0, tp / denom, 0);
// auc -> interpolate_pr_auc).
__global__ void metrics_compute_kernel(const unsigned long long *__restrict__ state,
                                       int32_t n_classes, int32_t n_thresholds,
                                       float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_classes) return;
  const unsigned long long *s = state + (int64_t)c * class_block(n_thresholds);
  const float tp0 = (float)s[0], fp0 = (float)s[1], fn0 = (float)s[2];
  out[3 * c + 0] = tp0 + fn0 > 0.0f ? tp0 / (tp0 + fn0) : 0.0f;
  out[3 * c + 1] = tp0 + fp0 > 0.0f ? tp0 / (tp0 + fp0) : 0.0f;
  const unsigned long long *pos = s + 3, *neg = s + 4 + n_thresholds;
  unsigned long long all_pos = 0;
  for (int b = 0; b <= n_thresholds; ++b) all_pos += pos[b];
  // walk the thresholds from the top: tp[t] = sum of bins above t
  unsigned long long tp_i = 0, fp_i = 0;  // at threshold t + 1
  float auc = 0.0f;
  tp_i = pos[n_thresholds];
  fp_i = neg[n_thresholds];
  // pairs (t, t+1) for t = T-2 .. 0; the sum order follows descending t, the
  // test tolerance covers TensorFlow's unspecified reduction order
  for (int t = n_thresholds - 2; t >= 0; --t) {
    const unsigned long long tp_lo = tp_i + pos[t + 1], fp_lo = fp_i + neg[t + 1];
    const float tp_a = (float)tp_lo, tp_b = (float)tp_i;  // tp[t], tp[t+1]
    const float p_a = tp_a + (float)fp_lo, p_b = tp_b + (float)fp_i;
    const float dtp = tp_a - tp_b;
    const float dp = fmaxf(p_a - p_b, 0.0f);
    const float slope = dp != 0.0f ? dtp / dp : 0.0f;  // div_no_nan
    const float intercept = tp_b - slope * p_b;
    const float pb0 = fmaxf(p_b, 0.0f);
    const float ratio =
        (p_a > 0.0f && p_b > 0.0f) ? (pb0 != 0.0f ? p_a / pb0 : 0.0f) : 1.0f;
    const float fn_b = (float)(all_pos - tp_i);
    const float denom = fmaxf(tp_b + fn_b, 0.0f);
    const float num = slope * (dtp + intercept * logf(ratio));
    auc += denom != 0.0f ? num / denom : 0.0f;
    tp_i = tp_lo;
    fp_i = fp_lo;
  }
  out[3 * c + 2] = auc;
}

}  // namespace
}  // namespace pgnn

extern "C" size_t pgnn_metrics_state_bytes(int32_t n_classes,
                                           int32_t n_thresholds) {
  if (n_classes <= 0 || n_thresholds < 2) return 0;
  return (size_t)n_classes * (size_t)pgnn::class_block(n_thresholds) * 8;
}

extern "C" int pgnn_metrics_update(const float *probs, int64_t ld_probs,
                                   const int32_t *labels, int64_t n_rows,
                                   int32_t n_classes, int32_t n_thresholds,
                                   void *state, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && n_classes > 0, PGNN_E_INVALID,
               "metrics_update: negative size");
  PGNN_REQUIRE(n_thresholds >= 2 && n_thresholds <= pgnn::kMaxThresholds,
               PGNN_E_INVALID, "metrics_update: num_thresholds out of range");
  PGNN_REQUIRE(ld_probs >= n_classes, PGNN_E_INVALID,
               "metrics_update: row stride smaller than n_classes");
  const size_t lds = (size_t)n_classes * pgnn::class_block(n_thresholds) * 4;
  PGNN_REQUIRE(lds <= 48 * 1024, PGNN_E_UNSUPPORTED,
               "metrics_update: n_classes * num_thresholds too large");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(probs && labels && state, PGNN_E_INVALID,
               "metrics_update: null pointer");
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(pgnn::metrics_update_kernel, dim3((unsigned)blocks),
                     dim3(256), lds, stream, probs, ld_probs, labels, n_rows,
                     n_classes, n_thresholds, (unsigned long long *)state);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_metrics_compute(const void *state, int32_t n_classes,
                                    int32_t n_thresholds, float *out,
                                    void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_classes > 0 && n_thresholds >= 2 &&
                   n_thresholds <= pgnn::kMaxThresholds,
               PGNN_E_INVALID, "metrics_compute: size out of range");
  PGNN_REQUIRE(state && out, PGNN_E_INVALID, "metrics_compute: null pointer");
  hipLaunchKernelGGL(pgnn::metrics_compute_kernel,
                     dim3((unsigned)((n_classes + 63) / 64)), dim3(64), 0,
                     stream, (const unsigned long long *)state, n_classes,
                     n_thresholds, out);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
