// Raster overlap of box footprints: models/nms.py:29-62 overlapped_boxes_3d,
// i.e. two cv2.fillPoly calls and three cv2.countNonZero per box pair, the
// rejection test inside random_box_shift under every shipped train config
// (models/preprocess.py:281-301, `max_overlap_rate: 0.01`).
//
// cv2.fillPoly(LINE_8, shift 0) writes, per polygon, (a) the 8-connected
// Bresenham line of every side (drawing.cpp Line -> LineIterator, left to
// right, after clipLine) and (b) the scan-line fill between pairs of active
// sides in 16.16 fixed point (FillEdgeCollection).  Nothing is rasterised
// here.  Both parts have closed forms per image row:
//   (a) step k of a line with major extent a and minor extent b sits at minor
//       offset floor((2bk + a - 1) / 2a); a row of an x-major line is one run
//       of k, a row of a y-major line one pixel;
//   (b) a side active on row y (y0 <= y < y1) is at x0 + (y - y0) dx exactly
//       (integer adds); the sorted active x are filled pairwise.
// So a row of a quadrilateral is the union of <= 6 integer intervals, and the
// three pixel counts are sums over rows of interval measures -- one lane per
// row, one wave per box pair, all in integers.  The only floating point is
// clipLine's double intersection (truncated like the C++ cast) and the final
// quotient np.float32(intersection) / (union - intersection).
// Checked against the literal restatement (oracle/raster_oracle.py) on random
// convex, self-intersecting, degenerate and clipped quadrilaterals.
#include "pgnn_common.h"

namespace {
using namespace pgnn;

constexpr int kShift = 16;

struct Line8 {   // one clipped, left-to-right polygon side
  int x, y;      // start pixel
  int a, b;      // major / minor extent
  int sy;        // row direction
  int ymajor;    // major axis is y
  int valid;
};

struct Side {    // scan-line edge of FillEdgeCollection
  int y0, y1;
  long long x, dx;
};

struct Poly {
  Line8 line[4];
  Side side[4];
  int n_sides;
  int fill_rows;  // scan-line fill covers rows [0, fill_rows) at most
};

__device__ inline long long cdiv(long long a, long long b) { return a / b; }

// cv::clipLine(Size2l, Point2l&, Point2l&)
__device__ bool clip_line(long long w, long long h, long long &x1,
                          long long &y1, long long &x2, long long &y2) {
  const long long right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

__device__ void setup_poly(const int (&px)[4], const int (&py)[4], int w, int h,
                           Poly &p) {
  p.n_sides = 0;
  int ymin = INT_MAX, ymax = INT_MIN;
  long long xmin = LLONG_MAX, xmax = -1;
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 3) & 3;  // previous vertex
    // ---- outline: LineIterator(img, pt0, pt1, 8, left_to_right = true)
    Line8 &l = p.line[i];
    long long x1 = px[j], y1 = py[j], x2 = px[i], y2 = py[i];
    bool ok = true;
    if (!(x1 >= 0 && x1 < w && x2 >= 0 && x2 < w && y1 >= 0 && y1 < h &&
          y2 >= 0 && y2 < h))
      ok = clip_line(w, h, x1, y1, x2, y2);
    long long dx = x2 - x1, dy = y2 - y1;
    if (dx < 0) {
      dx = -dx;
      dy = -dy;
      x1 = x2;
      y1 = y2;
    }
    int sy = 1;
    if (dy < 0) {
      dy = -dy;
      sy = -1;
    }
    l.valid = ok ? 1 : 0;
    l.x = (int)x1;
    l.y = (int)y1;
    l.sy = sy;
    l.ymajor = dy > dx;
    l.a = (int)(l.ymajor ? dy : dx);
    l.b = (int)(l.ymajor ? dx : dy);
    // ---- scan-line side (CollectPolyEdges), on the UNCLIPPED vertices
    if (py[j] != py[i]) {
      Side &s = p.side[p.n_sides++];
      const long long fx0 = (long long)px[j] << kShift,
                      fx1 = (long long)px[i] << kShift;
      if (py[j] < py[i]) {
        s.y0 = py[j];
        s.y1 = py[i];
        s.x = fx0;
      } else {
        s.y0 = py[i];
        s.y1 = py[j];
        s.x = fx1;
      }
      s.dx = cdiv(fx1 - fx0, (long long)py[i] - py[j]);
      const long long xe = s.x + (long long)(s.y1 - s.y0) * s.dx;
      ymin = min(ymin, s.y0);
      ymax = max(ymax, s.y1);
      xmin = min(xmin, min(s.x, xe));
      xmax = max(xmax, max(s.x, xe));
    }
  }
  // FillEdgeCollection's early returns
  p.fill_rows = 0;
  if (p.n_sides >= 2 && !(ymax < 0 || ymin >= h || xmax < 0 ||
                          xmin >= ((long long)w << kShift)))
    p.fill_rows = min(ymax, h);
}

// inclusive pixel intervals of one polygon on row y; returns their number
__device__ int row_intervals(const Poly &p, int w, int y, int (&lo)[6],
                             int (&hi)[6]) {
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    const Line8 &l = p.line[i];
    if (!l.valid) continue;
    const long long t = (long long)(y - l.y) * l.sy;  // rows from the start
    const long long a = l.a, b = l.b;
    if (l.ymajor) {
      if (t >= 0 && t <= a) {
        const int x = l.x + (int)((2 * b * t + a - 1) / (2 * a));
        lo[n] = x;
        hi[n] = x;
        ++n;
      }
    } else if (a == 0) {
      if (t == 0) {
        lo[n] = l.x;
        hi[n] = l.x;
        ++n;
      }
    } else if (t >= 0 && t <= b) {
      long long klo = 0, khi = a;
      if (b > 0) {
        const long long num = 2 * a * t - a + 1;  // ceil(num / 2b)
        klo = num <= 0 ? 0 : (num + 2 * b - 1) / (2 * b);
        khi = (2 * a * t + a) / (2 * b);
        if (khi > a) khi = a;
      }
      if (klo <= khi) {
        lo[n] = l.x + (int)klo;
        hi[n] = l.x + (int)khi;
        ++n;
      }
    }
  }
  if (y < p.fill_rows) {
    long long act[4];
    int na = 0;
    for (int i = 0; i < p.n_sides; ++i) {
      const Side &s = p.side[i];
      if (s.y0 <= y && y < s.y1) {
        long long x = s.x + (long long)(y - s.y0) * s.dx;
        int k = na++;
        while (k > 0 && act[k - 1] > x) {  // insertion sort
          act[k] = act[k - 1];
          --k;
        }
        act[k] = x;
      }
    }
    for (int k = 0; k + 1 < na; k += 2) {
      // FillEdgeCollection: left end rounded up, right end down
      long long xa = (act[k] + ((1ll << kShift) - 1)) >> kShift,
                xb = act[k + 1] >> kShift;
      if (xa < w && xb >= 0) {
        if (xa < 0) xa = 0;
        if (xb >= w) xb = w - 1;
        if (xa <= xb) {
          lo[n] = (int)xa;
          hi[n] = (int)xb;
          ++n;
        }
      }
    }
  }
  return n;
}

// sort by lo, merge overlaps; returns the number of disjoint intervals
__device__ int normalize(int (&lo)[6], int (&hi)[6], int n) {
  for (int i = 1; i < n; ++i) {
    const int l = lo[i], h = hi[i];
    int k = i;
    while (k > 0 && lo[k - 1] > l) {
      lo[k] = lo[k - 1];
      hi[k] = hi[k - 1];
      --k;
    }
    lo[k] = l;
    hi[k] = h;
  }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (m > 0 && lo[i] <= hi[m - 1]) {
      hi[m - 1] = max(hi[m - 1], hi[i]);
    } else {
      lo[m] = lo[i];
      hi[m] = hi[i];
      ++m;
    }
  }
  return m;
}

__device__ inline long long wave_sum(long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// one wave per pair (single_box, box_list[i]); corners are int32 [8,3]
__global__ __launch_bounds__(64) void raster_overlap_kernel(
    const int32_t *__restrict__ single_box, const int32_t *__restrict__ boxes,
    int64_t n_boxes, double *__restrict__ overlap) {
  const int64_t i = blockIdx.x;
  if (i >= n_boxes) return;
  const int lane = threadIdx.x;
  const int32_t *b0 = single_box, *b1 = boxes + i * 24;
  int mn0[3], mx0[3], mn1[3], mx1[3];
  for (int c = 0; c < 3; ++c) {
    mn0[c] = mn1[c] = INT_MAX;
    mx0[c] = mx1[c] = INT_MIN;
    for (int v = 0; v < 8; ++v) {
      mn0[c] = min(mn0[c], b0[3 * v + c]);
      mx0[c] = max(mx0[c], b0[3 * v + c]);
      mn1[c] = min(mn1[c], b1[3 * v + c]);
      mx1[c] = max(mx1[c], b1[3 * v + c]);
    }
  }
  bool apart = false;
  for (int c = 0; c < 3; ++c)
    apart = apart || mx0[c] < mn1[c] || mn0[c] > mx1[c];
  if (apart) {  // nms.py:36-44
    if (lane == 0) overlap[i] = 0.0;
    return;
  }
  const int x_draw_min = min(mn0[0], mn1[0]), x_draw_max = max(mx0[0], mx1[0]);
  const int z_draw_min = min(mn0[2], mn1[2]), z_draw_max = max(mx0[2], mx1[2]);
  const int w = x_draw_max - x_draw_min, h = z_draw_max - z_draw_min;
  int px[4], py[4];
  Poly p0, p1;
  for (int v = 0; v < 4; ++v) {
    px[v] = b0[3 * v] - x_draw_min;
    py[v] = b0[3 * v + 2] - z_draw_min;
  }
  setup_poly(px, py, w, h, p0);
  for (int v = 0; v < 4; ++v) {
    px[v] = b1[3 * v] - x_draw_min;
    py[v] = b1[3 * v + 2] - z_draw_min;
  }
  setup_poly(px, py, w, h, p1);
  long long area0 = 0, area1 = 0, shared = 0;
  for (int y = lane; y < h; y += 64) {
    int lo0[6], hi0[6], lo1[6], hi1[6];
    int n0 = row_intervals(p0, w, y, lo0, hi0);
    int n1 = row_intervals(p1, w, y, lo1, hi1);
    n0 = normalize(lo0, hi0, n0);
    n1 = normalize(lo1, hi1, n1);
    for (int a = 0; a < n0; ++a) area0 += hi0[a] - lo0[a] + 1;
    for (int b = 0; b < n1; ++b) area1 += hi1[b] - lo1[b] + 1;
    for (int a = 0; a < n0; ++a)
      for (int b = 0; b < n1; ++b) {
        const int s = min(hi0[a], hi1[b]) - max(lo0[a], lo1[b]) + 1;
        if (s > 0) shared += s;
      }
  }
  area0 = wave_sum(area0);
  area1 = wave_sum(area1);
  shared = wave_sum(shared);
  if (lane == 0) {  // nms.py:57-60
    const long long shared_y =
        (long long)min(mx1[1], mx0[1]) - (long long)max(mn1[1], mn0[1]);
    const long long intersection = shared_y * shared;
    const long long uni = (long long)(mx1[1] - mn1[1]) * area1 +
                          (long long)(mx0[1] - mn0[1]) * area0;
    overlap[i] = (double)(float)intersection / (double)(uni - intersection);
  }
}

}  // namespace

extern "C" int pgnn_overlapped_boxes_3d_raster(const int32_t *single_box_corners,
                                               const int32_t *box_corners,
                                               int64_t n_boxes, double *overlap,
                                               void *stream_) {
  PGNN_GUARD_BEGIN
  hipStream_t stream = (hipStream_t)stream_;
  PGNN_REQUIRE(n_boxes >= 0 && n_boxes < (1ll << 31), PGNN_E_INVALID,
               "overlapped_boxes_3d_raster: bad size");
  if (n_boxes == 0) return 0;
  PGNN_REQUIRE(single_box_corners && box_corners && overlap, PGNN_E_INVALID,
               "overlapped_boxes_3d_raster: null pointer");
  hipLaunchKernelGGL(raster_overlap_kernel, dim3((unsigned)n_boxes), dim3(64), 0,
                     stream, single_box_corners, box_corners, n_boxes, overlap);
  PGNN_HIP(hipGetLastError());
  return 0;
  PGNN_GUARD_END
}
