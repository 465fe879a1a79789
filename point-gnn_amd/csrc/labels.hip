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
#include <string.h>

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

// T: float32 vertices (inference-side callers) or float64 (train.py:100-118
// passes the float64 vertex_coord_list of the augmented cloud)
template <typename T>
__global__ void assign_labels_kernel(const T *__restrict__ xyz, int64_t n,
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
template <typename T>
__global__ void box_encode_f64_kernel(const int32_t *__restrict__ labels,
                                      const T *__restrict__ xyz,
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

// ---- augmentation primitives (models/preprocess.py) on float64 points -------
// The reference's augmentations turn the float32 cloud into float64 with the
// first `xyz.dot(R.T)` and keep it there until train.py:124 casts back; the
// point-wise parts run here on a float64 [n,3] buffer.

// xyz <- xyz @ rot^T + shift for the selected points (select null: all)
__global__ void points_affine_kernel(double *__restrict__ xyz, int64_t n,
                                     const double r0, const double r1,
                                     const double r2, const double r3,
                                     const double r4, const double r5,
                                     const double r6, const double r7,
                                     const double r8, const double s0,
                                     const double s1, const double s2,
                                     int has_rot,
                                     const int32_t *__restrict__ select) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (select && select[i] == 0) return;
  double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  if (has_rot) {
    const double nx = (x * r0 + y * r1) + z * r2;
    const double ny = (x * r3 + y * r4) + z * r5;
    const double nz = (x * r6 + y * r7) + z * r8;
    x = nx;
    y = ny;
    z = nz;
  }
  xyz[3 * i] = x + s0;
  xyz[3 * i + 1] = y + s1;
  xyz[3 * i + 2] = z + s2;
}

// inside[i] = point i lies strictly inside box `rec` (sel_xyz_in_box3d on
// float64 points); *count += number of inside points with exclude[i] == 0
__global__ void points_in_box_f64_kernel(const double *__restrict__ xyz,
                                         int64_t n, LabelRecord rec,
                                         const int32_t *__restrict__ exclude,
                                         int32_t *__restrict__ inside,
                                         int32_t *__restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool in = false;
  if (i < n) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    in = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double p = (x * rec.normals[3 * k] + y * rec.normals[3 * k + 1]) +
                       z * rec.normals[3 * k + 2];
      in = in && p > rec.lower[k] && p < rec.upper[k];
    }
    if (inside) inside[i] = in ? 1 : 0;
    if (exclude && exclude[i] != 0) in = false;
  }
  if (count) {
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(count, (int)__popcll(bal));
  }
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" int pgnn_points_affine_f64(double *xyz, int64_t n_points,
                                      const double *rot_3x3,
                                      const double *shift_3,
                                      const int32_t *select, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_points >= 0, PGNN_E_INVALID, "points_affine: bad size");
  if (n_points == 0) return 0;
  PGNN_REQUIRE(xyz, PGNN_E_INVALID, "points_affine: null pointer");
  double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, s[3] = {0, 0, 0};
  if (rot_3x3)
    for (int i = 0; i < 9; ++i) r[i] = rot_3x3[i];
  if (shift_3)
    for (int i = 0; i < 3; ++i) s[i] = shift_3[i];
  hipLaunchKernelGGL(points_affine_kernel,
                     dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     stream, xyz, n_points, r[0], r[1], r[2], r[3], r[4], r[5],
                     r[6], r[7], r[8], s[0], s[1], s[2], rot_3x3 ? 1 : 0,
                     select);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_points_in_box_f64(const double *xyz, int64_t n_points,
                                      const double *box_record_24,
                                      const int32_t *exclude, int32_t *inside,
                                      int32_t *count, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_points >= 0 && box_record_24, PGNN_E_INVALID,
               "points_in_box: bad argument");
  if (count) PGNN_HIP(hipMemsetAsync(count, 0, 4, stream));
  if (n_points == 0) return 0;
  PGNN_REQUIRE(xyz, PGNN_E_INVALID, "points_in_box: null pointer");
  LabelRecord rec;
  memcpy(&rec, box_record_24, sizeof rec);
  hipLaunchKernelGGL(points_in_box_f64_kernel,
                     dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     stream, xyz, n_points, rec, exclude, inside, count);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}

namespace {
template <typename T>
int assign_impl(const T *xyz, int64_t n_points, const double *label_records,
                int32_t n_records, int32_t *cls_labels, double *boxes_3d,
                float *valid_boxes, int32_t *owner, hipStream_t stream) {
  PGNN_REQUIRE(n_points >= 0 && n_records >= 0, PGNN_E_INVALID,
               "assign_box_labels: bad size");
  if (n_points == 0) return 0;
  PGNN_REQUIRE(xyz && (label_records || n_records == 0), PGNN_E_INVALID,
               "assign_box_labels: null pointer");
  hipLaunchKernelGGL(assign_labels_kernel<T>,
                     dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     stream, xyz, n_points,
                     reinterpret_cast<const LabelRecord *>(label_records),
                     n_records, cls_labels, boxes_3d, valid_boxes, owner);
  PGNN_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int encode64_impl(const int32_t *cls_labels, const T *xyz, const double *boxes,
                  const double *class_table, int32_t n_table, int64_t n_rows,
                  int32_t boxes_per_row, float *encoded, hipStream_t stream) {
  PGNN_REQUIRE(n_rows >= 0 && boxes_per_row > 0 && n_table >= 0,
               PGNN_E_INVALID, "box_encode_f64: bad size");
  if (n_rows == 0) return 0;
  PGNN_REQUIRE(cls_labels && xyz && boxes && encoded &&
                   (class_table || n_table == 0),
               PGNN_E_INVALID, "box_encode_f64: null pointer");
  const int64_t total = n_rows * boxes_per_row;
  hipLaunchKernelGGL(box_encode_f64_kernel<T>,
                     dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     stream, cls_labels, xyz, boxes, class_table, n_table,
                     n_rows, boxes_per_row, encoded);
  PGNN_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int pgnn_assign_box_labels(const float *xyz, int64_t n_points,
                                      const double *label_records,
                                      int32_t n_records, int32_t *cls_labels,
                                      double *boxes_3d, float *valid_boxes,
                                      int32_t *owner, void *stream_) {
  PGNN_GUARD_BEGIN
  return assign_impl(xyz, n_points, label_records, n_records, cls_labels,
                     boxes_3d, valid_boxes, owner, (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_assign_box_labels_f64(const double *xyz, int64_t n_points,
                                          const double *label_records,
                                          int32_t n_records,
                                          int32_t *cls_labels, double *boxes_3d,
                                          float *valid_boxes, int32_t *owner,
                                          void *stream_) {
  PGNN_GUARD_BEGIN
  return assign_impl(xyz, n_points, label_records, n_records, cls_labels,
                     boxes_3d, valid_boxes, owner, (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_box_encode_f64(const int32_t *cls_labels, const float *xyz,
                                   const double *boxes,
                                   const double *class_table, int32_t n_table,
                                   int64_t n_rows, int32_t boxes_per_row,
                                   float *encoded, void *stream_) {
  PGNN_GUARD_BEGIN
  return encode64_impl(cls_labels, xyz, boxes, class_table, n_table, n_rows,
                       boxes_per_row, encoded, (hipStream_t)stream_);
  PGNN_GUARD_END
}

extern "C" int pgnn_box_encode_f64_xyz64(const int32_t *cls_labels,
                                         const double *xyz, const double *boxes,
                                         const double *class_table,
                                         int32_t n_table, int64_t n_rows,
                                         int32_t boxes_per_row, float *encoded,
                                         void *stream_) {
  PGNN_GUARD_BEGIN
  return encode64_impl(cls_labels, xyz, boxes, class_table, n_table, n_rows,
                       boxes_per_row, encoded, (hipStream_t)stream_);
  PGNN_GUARD_END
}
