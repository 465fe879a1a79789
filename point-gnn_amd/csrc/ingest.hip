// KITTI frame ingest, the step right before the path:
// velodyne scan -> camera frame -> camera-image crop, as one filter with
// ordered compaction (dataset/kitti_dataset.py:587-609 get_velo_points,
// :998-1006 velo_points_to_cam, :1036-1052 cam_points_to_image, :666-689
// get_cam_points_in_image, :990-996 rgb_to_cam_points).
//
// HBM-bound streaming: 16 B read per scan point, 16-28 B written per kept
// point; two passes (count per block, scan, write) keep np.nonzero order.
#include "pgnn_common.h"
#include "sort.h"

namespace pgnn {
namespace {

constexpr int kBlock = 256;

struct IngestArgs {
  const float *velo;  // [n,4] x y z reflectance
  int64_t n;
  float r[9];         // velo_to_cam[:3,:3] as float32, row-major
  float t[3];         // velo_to_cam[:3,3] as float32
  double p[9];        // cam_to_image[:, :3] (float64 holding P2's float32)
  double width, height;
  const uint8_t *image;  // optional [H,W,3] BGR (cv2.imread layout)
  int64_t img_h, img_w;
};

struct Projected {
  float x, y, z;
  double u, v;
  bool keep;
};

// kitti_dataset.py:1002-1005: float32 matmul + float32 add.  The products are
// accumulated in k order with fused multiply-adds, which is what the sgemm
// micro-kernels NumPy dispatches to do; the parity test states the (<= 1 ulp)
// bound for BLAS builds that associate differently.
__device__ __forceinline__ Projected project_point(const IngestArgs &a,
                                                   int64_t i) {
  const float4 q = reinterpret_cast<const float4 *>(a.velo)[i];
  Projected o;
  o.x = __fmaf_rn(q.z, a.r[2], __fmaf_rn(q.y, a.r[1], q.x * a.r[0])) + a.t[0];
  o.y = __fmaf_rn(q.z, a.r[5], __fmaf_rn(q.y, a.r[4], q.x * a.r[3])) + a.t[1];
  o.z = __fmaf_rn(q.z, a.r[8], __fmaf_rn(q.y, a.r[7], q.x * a.r[6])) + a.t[2];
  // :675 front points, :678-684 projection in float64 and the image test
  const double X = o.x, Y = o.y, Z = o.z;
  const double iu = (X * a.p[0] + Y * a.p[1]) + Z * a.p[2];
  const double iv = (X * a.p[3] + Y * a.p[4]) + Z * a.p[5];
  const double iw = (X * a.p[6] + Y * a.p[7]) + Z * a.p[8];
  o.u = iu / iw;
  o.v = iv / iw;
  o.keep = o.z > 0.1f && o.u > 0.0 && o.u < a.width && o.v > 0.0 &&
           o.v < a.height;
  return o;
}

__global__ __launch_bounds__(kBlock) void ingest_count_kernel(
    IngestArgs a, int32_t *__restrict__ block_count) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool keep = i < a.n && project_point(a, i).keep;
  __shared__ int wave_tot[kBlock / 64];
  const unsigned long long bal = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += wave_tot[w];
    block_count[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void ingest_write_kernel(
    IngestArgs a, const int32_t *__restrict__ block_offset,
    float *__restrict__ out_xyz, float *__restrict__ out_attr, int attr_dim,
    int64_t capacity, int32_t *__restrict__ out_count, int n_blocks) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  Projected o;
  o.keep = false;
  if (i < a.n) o = project_point(a, i);
  __shared__ int wave_tot[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(o.keep);
  if (lane == 0) wave_tot[wave] = __popcll(bal);
  __syncthreads();
  int64_t slot = block_offset[blockIdx.x];
  for (int w = 0; w < wave; ++w) slot += wave_tot[w];
  slot += __popcll(bal & ((1ull << lane) - 1ull));
  if (o.keep && slot < capacity) {
    out_xyz[3 * slot] = o.x;
    out_xyz[3 * slot + 1] = o.y;
    out_xyz[3 * slot + 2] = o.z;
    float *at = out_attr + slot * attr_dim;
    at[0] = a.velo[4 * i + 3];
    if (attr_dim == 4) {
      // :994-995 image[int32(v), int32(u), ::-1] / 255 (BGR -> RGB)
      const int64_t px = (int64_t)(int)o.u, py = (int64_t)(int)o.v;
      float r = 0.0f, g = 0.0f, b = 0.0f;
      if (a.image && px >= 0 && px < a.img_w && py >= 0 && py < a.img_h) {
        const uint8_t *c = a.image + (py * a.img_w + px) * 3;
        b = (float)c[0] / 255.0f;
        g = (float)c[1] / 255.0f;
        r = (float)c[2] / 255.0f;
      }
      at[1] = r;
      at[2] = g;
      at[3] = b;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = block_offset[n_blocks];
}

// workspace: [block counts | block offsets (+ total) | scan scratch]
struct IngestLayout {
  int32_t *counts, *offsets;
  void *scan_ws;
  size_t scan_bytes;
};

bool carve_ingest(Arena &ar, int64_t n_points, IngestLayout *L) {
  const size_t nb = (size_t)((n_points + kBlock - 1) / kBlock) + 2;
  L->scan_bytes = scan_scratch_bytes((int64_t)nb);
  L->counts = ar.take<int32_t>(nb);
  L->offsets = ar.take<int32_t>(nb + 1);
  L->scan_ws = ar.take<char>(L->scan_bytes > 0 ? L->scan_bytes : 1);
  return L->scan_ws != nullptr;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

extern "C" size_t pgnn_kitti_ingest_workspace_bytes(int64_t n_points) {
  if (n_points < 0) return 0;
  Arena ar(nullptr, 0);
  IngestLayout L;
  carve_ingest(ar, n_points, &L);
  return align_up(ar.used, 256);
}

extern "C" int pgnn_kitti_cam_points_in_image(
    const float *velo_points, int64_t n_points, const float *velo_to_cam_3x4,
    const double *cam_to_image_3x3, double image_width, double image_height,
    const uint8_t *image_bgr, int64_t image_rows, int64_t image_cols,
    void *workspace, size_t workspace_bytes, float *out_xyz, float *out_attr,
    int32_t attr_dim, int64_t capacity, int32_t *out_count, void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_points >= 0 && capacity >= 0 && out_count &&
                   (attr_dim == 1 || attr_dim == 4),
               PGNN_E_INVALID, "kitti_cam_points_in_image: bad argument");
  PGNN_REQUIRE(velo_to_cam_3x4 && cam_to_image_3x3, PGNN_E_INVALID,
               "kitti_cam_points_in_image: null calibration (host pointers)");
  if (n_points == 0) {
    PGNN_HIP(hipMemsetAsync(out_count, 0, 4, stream));
    return 0;
  }
  PGNN_REQUIRE(velo_points && (capacity == 0 || (out_xyz && out_attr)),
               PGNN_E_INVALID, "kitti_cam_points_in_image: null pointer");
  PGNN_REQUIRE(attr_dim == 1 || image_bgr == nullptr ||
                   (image_rows > 0 && image_cols > 0),
               PGNN_E_INVALID, "kitti_cam_points_in_image: bad image shape");
  const int64_t nb = (n_points + kBlock - 1) / kBlock;
  Arena ar(workspace, workspace_bytes);
  IngestLayout L;
  PGNN_REQUIRE(carve_ingest(ar, n_points, &L), PGNN_E_WORKSPACE,
               "kitti_cam_points_in_image: workspace too small "
               "(see pgnn_kitti_ingest_workspace_bytes)");
  int32_t *counts = L.counts, *offsets = L.offsets;
  void *scan_ws = L.scan_ws;
  const size_t scan_bytes = L.scan_bytes;
  IngestArgs a;
  a.velo = velo_points;
  a.n = n_points;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      a.r[3 * r + c] = velo_to_cam_3x4[4 * r + c];
      a.p[3 * r + c] = cam_to_image_3x3[3 * r + c];
    }
    a.t[r] = velo_to_cam_3x4[4 * r + 3];
  }
  a.width = image_width;
  a.height = image_height;
  a.image = attr_dim == 4 ? image_bgr : nullptr;
  a.img_h = image_rows;
  a.img_w = image_cols;
  hipLaunchKernelGGL(ingest_count_kernel, dim3((unsigned)nb), dim3(kBlock), 0,
                     stream, a, counts);
  PGNN_HIP(hipGetLastError());
  int rc = exclusive_scan_i32(counts, offsets, nb, scan_ws, scan_bytes, stream);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(ingest_write_kernel, dim3((unsigned)nb), dim3(kBlock), 0,
                     stream, a, offsets, out_xyz, out_attr, attr_dim, capacity,
                     out_count, (int)nb);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
