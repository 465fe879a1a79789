// Training-target construction for the vertices of the last graph level
// (SURVEY.md §8(f) 4): which ground-truth box contains each vertex
// (dataset/kitti_dataset.py:143-162 sel_xyz_in_box3d, :1132-1284 the three
// assign_classaware_*_label_to_points variants) and the float64 flavour of the
// box encoder that train.py:120-130 feeds with those float64 boxes.
//
// The reference loops over the label boxes and overwrites NumPy slices, so a
// vertex inside several boxes keeps what the LAST one wrote; here one thread
// per vertex walks the (few dozen) boxes in the same order.  Float64 like the
// reference (`np.matmul(xyz_f32, normals_f64.T)`); latency-bound, negligible
// bytes.
#include "pgnn_common.h"

namespace pgnn {
namespace {

// one ground-truth box, prepared on the host (box3d_to_normals, :118-141)
struct LabelRecord {
  double normals[9];  // rows wx, wy, wz
  double lower[3], upper[3];
  double action;      // 0 skip, 1 object (class + box + valid), 2 class only
  double cls;         // class value written for action 1 / 2
  double box[7];      // x3d, y3d, z3d, length, height, width, wrapped yaw
};
static_assert(sizeof(LabelRecord) == 24 * sizeof(double), "record layout");

__global__ void assign_labels_kernel(const float *__restrict__ xyz, int64_t n,
                                     const LabelRecord *__restrict__ rec,
                                     int n_rec, int32_t *__restrict__ cls,
                                     double *__restrict__ boxes,
                                     float *__restrict__ valid,
                                     int32_t *__restrict__ owner) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  int c = 0, own = -1, box_from = -1;
  float v = 0.0f;
  for (int r = 0; r < n_rec; ++r) {
    const LabelRecord &b = rec[r];
    if (b.action == 0.0) continue;
    bool in = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double p = (x * b.normals[3 * k] + y * b.normals[3 * k + 1]) +
                       z * b.normals[3 * k + 2];
      in = in && p > b.lower[k] && p < b.upper[k];
    }
    if (!in) continue;
    c = (int)b.cls;
    own = r;
    if (b.action == 1.0) {
      box_from = r;
      v = 1.0f;
    } else {
      v = 0.0f;  // class only: the box slot keeps what an earlier object wrote
    }
  }
  if (cls) cls[i] = c;
  if (valid) valid[i] = v;
  if (owner) owner[i] = own;
  if (boxes) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
      boxes[7 * i + k] = box_from >= 0 ? rec[box_from].box[k] : 0.0;
  }
}

// box_encoding.py:231-263 evaluated in float64 (float64 boxes, float32 points,
// Python-float medians) and rounded to float32 once, as train.py:120-130 does
// with `.astype(np.float32)`.  table64 rows = {l, h, w, yaw_offset, active}.
__global__ void box_encode_f64_kernel(const int32_t *__restrict__ labels,
                                      const float *__restrict__ xyz,
                                      const double *__restrict__ boxes,
                                      const double *__restrict__ table64,
                                      int n_table, int64_t rows, int per_row,
                                      float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * per_row) return;
  const int64_t r = idx / per_row;
  const int col = (int)(idx - r * per_row);
  const double *b = boxes + idx * 7;
  double d[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = b[i];
  d[0] = b[0] - (double)xyz[3 * r];
  d[1] = b[1] - (double)xyz[3 * r + 1];
  d[2] = b[2] - (double)xyz[3 * r + 2];
  const int lab = labels[r];
  if (col == 0 && lab >= 0 && lab < n_table && table64[5 * lab + 4] != 0.0) {
    const double l = table64[5 * lab], h = table64[5 * lab + 1],
                 w = table64[5 * lab + 2], yo = table64[5 * lab + 3];
    d[0] = d[0] / l;
    d[1] = d[1] / h;
    d[2] = d[2] / w;
    d[3] = log(b[3] / l);
    d[4] = log(b[4] / h);
    d[5] = log(b[5] / w);
    const double y = yo != 0.0 ? b[6] - yo : b[6];
    d[6] = y / 0.78539816339744830962;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) out[idx * 7 + i] = (float)d[i];
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" int pgnn_assign_box_labels(const float *xyz, int64_t n_points,
                                      const double *label_records,
                                      int32_t n_records, int32_t *cls_labels,
                                      double *boxes_3d, float *valid_boxes,
                                      int32_t *owner, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_points >= 0 && n_records >= 0, PGNN_E_INVALID,
               "assign_box_labels: bad size");
  if (n_points == 0) return 0;
  PGNN_REQUIRE(xyz && (label_records || n_records == 0), PGNN_E_INVALID,
               "assign_box_labels: null pointer");
  hipLaunchKernelGGL(assign_labels_kernel,
                     dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     stream, xyz, n_points,
                     reinterpret_cast<const LabelRecord *>(label_records),
                     n_records, cls_labels, boxes_3d, valid_boxes, owner);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_box_encode_f64(const int32_t *cls_labels, const float *xyz,
                                   const double *boxes,
                                   const double *class_table, int32_t n_table,
                                   int64_t n_rows, int32_t boxes_per_row,
                                   float *encoded, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_rows >= 0 && boxes_per_row > 0 && n_table >= 0,
               PGNN_E_INVALID, "box_encode_f64: bad size");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(cls_labels && xyz && boxes && encoded &&
                   (class_table || n_table == 0),
               PGNN_E_INVALID, "box_encode_f64: null pointer");
  const int64_t total = n_rows * boxes_per_row;
  hipLaunchKernelGGL(box_encode_f64_kernel,
                     dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     stream, cls_labels, xyz, boxes, class_table, n_table,
                     n_rows, boxes_per_row, encoded);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
