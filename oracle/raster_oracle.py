"""TEST INFRASTRUCTURE (oracle) -- never imported by the product path.

CPU restatement of the cv2 raster overlap the reference's training
augmentation goes through under every shipped train config
(`max_overlap_rate: 0.01`):

    models/preprocess.py:281-301  random_box_shift -> nms.overlapped_boxes_3d
    models/nms.py:29-62           overlapped_boxes_3d (cv2.fillPoly twice,
                                  cv2.countNonZero three times per box pair)

The arithmetic of `cv2.fillPoly` lives in a THIRD-PARTY dependency that is
absent here and unpinned by the reference (`README.md:27-31` lists
`opencv-python` without a version).  Restated below is the published algorithm
of OpenCV 4.2.0 (the `opencv-python` release current when the reference was
published, March 2020), `modules/imgproc/src/drawing.cpp`:

    fillPoly            -> CollectPolyEdges (+ Line per polygon side)
                        -> FillEdgeCollection
    Line                -> LineIterator(img, pt1, pt2, 8, left_to_right=true)
    LineIterator        -> clipLine + the 8-connected Bresenham stepping of
                           LineIterator::operator++
    clipLine            -> the int64 Cohen-Sutherland variant with double
                           intersections truncated toward zero

for `lineType = LINE_8`, `shift = 0`, `offset = (0, 0)` -- the defaults the
reference's call `cv2.fillPoly(buf, [pts], color=1)` uses -- on a
single-channel int32 image.  Integer arithmetic throughout (XY_SHIFT = 16
fixed point for the scan-line edges), so a restatement is either identical
or visibly wrong.

PARITY UNPINNED against a real cv2 (not installable: no network).  Pinned
instead by properties that hold for cv2.fillPoly whatever the version
(tests/test_raster_cpu.py): axis-aligned integer rectangles fill inclusively
((w+1)(h+1) pixels), single points / degenerate polygons draw their outline,
the fill is invariant under cyclic vertex rotation and reversal, pixels of a
convex polygon = outline pixels + strict interior, clipping never writes
outside the buffer; and by tests/golden/aug_kitti.npz, written by the
reference's OWN preprocess.random_box_shift / nms.overlapped_boxes_3d running
on top of this `cv2` stand-in (tests/golden/make_golden_aug.py).
"""
import types

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _trunc(x):
    """C++ (int64)(double) conversion: toward zero."""
    return int(x)


def clip_line(width, height, pt1, pt2):
    """cv::clipLine(Size2l, Point2l&, Point2l&) -- drawing.cpp.
    Returns (inside, pt1, pt2)."""
    x1, y1 = int(pt1[0]), int(pt1[1])
    x2, y2 = int(pt2[0]), int(pt2[1])
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, (x1, y1), (x2, y2)
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += _trunc(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += _trunc(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += _trunc(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += _trunc(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
        assert (c1 & c2) != 0 or (x1 | y1 | x2 | y2) >= 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line_pixels(width, height, pt1, pt2):
    """Pixels visited by cv::LineIterator(img, pt1, pt2, 8, left_to_right=true)
    -- what drawing.cpp's Line() writes.  Returns a list of (x, y)."""
    x1, y1 = int(pt1[0]), int(pt1[1])
    x2, y2 = int(pt2[0]), int(pt2[1])
    if not (0 <= x1 < width and 0 <= x2 < width and
            0 <= y1 < height and 0 <= y2 < height):
        ok, (x1, y1), (x2, y2) = clip_line(width, height, (x1, y1), (x2, y2))
        if not ok:
            return []
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:              # left_to_right: start from the left end
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    step_x, step_y = 1, 1
    if dy < 0:
        dy, step_y = -dy, -1
    if dy > dx:             # y is the major axis
        major, minor = dy, dx
        major_step, minor_step = (0, step_y), (step_x, 0)
    else:
        major, minor = dx, dy
        major_step, minor_step = (step_x, 0), (0, step_y)
    err = major - 2 * minor
    plus_delta, minus_delta = 2 * major, -2 * minor
    out = []
    x, y = x1, y1
    for _ in range(major + 1):
        out.append((x, y))
        neg = err < 0
        err += minus_delta + (plus_delta if neg else 0)
        x += major_step[0] + (minor_step[0] if neg else 0)
        y += major_step[1] + (minor_step[1] if neg else 0)
    return out


class _Edge:
    __slots__ = ("y0", "y1", "x", "dx", "next")

    def __init__(self):
        self.y0 = self.y1 = 0
        self.x = self.dx = 0
        self.next = None


def _cdiv(a, b):
    """C++ integer division (toward zero)."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def fill_poly(img, polygons, color=1):
    """cv2.fillPoly(img, pts, color) for LINE_8, shift 0, no offset.
    `img` [H, W] is written in place and returned; `polygons` is a list of
    integer [n, 2] (x, y) arrays."""
    height, width = img.shape[:2]
    edges = []
    for poly in polygons:
        v = [(int(p[0]), int(p[1])) for p in np.asarray(poly).reshape(-1, 2)]
        count = len(v)
        if count == 0:
            continue
        # CollectPolyEdges: outline with Line(), sides as fixed-point edges
        px0, py0 = v[-1]
        pt0 = (px0 << XY_SHIFT, py0)
        for i in range(count):
            px1, py1 = v[i]
            pt1 = (px1 << XY_SHIFT, py1)
            t0 = ((pt0[0] + (XY_ONE >> 1)) >> XY_SHIFT, pt0[1])
            t1 = ((pt1[0] + (XY_ONE >> 1)) >> XY_SHIFT, pt1[1])
            for (x, y) in line_pixels(width, height, t0, t1):
                img[y, x] = color
            if pt0[1] != pt1[1]:
                e = _Edge()
                if pt0[1] < pt1[1]:
                    e.y0, e.y1, e.x = pt0[1], pt1[1], pt0[0]
                else:
                    e.y0, e.y1, e.x = pt1[1], pt0[1], pt1[0]
                e.dx = _cdiv(pt1[0] - pt0[0], pt1[1] - pt0[1])
                edges.append(e)
            pt0 = pt1
    _fill_edge_collection(img, edges, color)
    return img


def _fill_edge_collection(img, edges, color):
    height, width = img.shape[:2]
    total = len(edges)
    if total < 2:
        return
    y_max, y_min = -(1 << 31), (1 << 31) - 1
    x_max, x_min = -1, (1 << 63) - 1
    for e in edges:
        assert e.y0 < e.y1
        x1 = e.x + (e.y1 - e.y0) * e.dx
        y_min, y_max = min(y_min, e.y0), max(y_max, e.y1)
        x_min, x_max = min(x_min, e.x, x1), max(x_max, e.x, x1)
    if y_max < 0 or y_min >= height or x_max < 0 or \
            x_min >= (width << XY_SHIFT):
        return
    edges.sort(key=lambda e: (e.y0, e.x, e.dx))      # CmpEdges
    sentinel = _Edge()
    sentinel.y0 = (1 << 31) - 1
    edges = edges + [sentinel]
    tmp = _Edge()
    tmp.next = None
    i = 0
    e = edges[0]
    y_max = min(y_max, height)
    y = e.y0
    while y < y_max:
        draw = 0
        clipline = y < 0
        prelast, last = tmp, tmp.next
        while last is not None or e.y0 == y:
            if last is not None and last.y1 == y:
                prelast.next = last.next       # edge ends above this row
                last = last.next
                continue
            keep_prelast = prelast
            if last is not None and (e.y0 > y or last.x < e.x):
                prelast, last = last, last.next
            elif i < total:
                prelast.next = e               # edge starts on this row
                e.next = last
                prelast = e
                i += 1
                e = edges[i]
            else:
                break
            if draw:
                if not clipline:
                    # the left end of a span is rounded UP, the right end
                    # down: x1 = (x + XY_ONE - 1) >> XY_SHIFT, x2 = x >> XY_SHIFT
                    if keep_prelast.x > prelast.x:
                        xa = (prelast.x + XY_ONE - 1) >> XY_SHIFT
                        xb = keep_prelast.x >> XY_SHIFT
                    else:
                        xa = (keep_prelast.x + XY_ONE - 1) >> XY_SHIFT
                        xb = prelast.x >> XY_SHIFT
                    if xa < width and xb >= 0:
                        xa = max(xa, 0)
                        xb = min(xb, width - 1)
                        if xa <= xb:
                            img[y, xa:xb + 1] = color
                keep_prelast.x += keep_prelast.dx
                prelast.x += prelast.dx
            draw ^= 1
        # bubble sort of the active list by x
        keep_prelast = None
        while True:
            prelast, last = tmp, tmp.next
            sort_flag = False
            while last is not keep_prelast and last is not None and \
                    last.next is not None:
                te = last.next
                if last.x > te.x:
                    prelast.next = te
                    last.next = te.next
                    te.next = last
                    prelast = te
                    sort_flag = True
                else:
                    prelast, last = last, te
            keep_prelast = prelast
            if not (sort_flag and keep_prelast is not tmp.next and
                    keep_prelast is not tmp):
                break
        y += 1


def count_non_zero(img):
    """cv2.countNonZero: a Python int."""
    return int(np.count_nonzero(img))


def cv2_stub():
    """A module object with the two cv2 entry points nms.overlapped_boxes_3d
    uses, so that the reference's own file runs on top of this restatement."""
    m = types.ModuleType("cv2")

    def fillPoly(img, pts, color, lineType=8, shift=0, offset=(0, 0)):
        assert lineType == 8 and shift == 0 and tuple(offset) == (0, 0)
        assert img.ndim == 2
        return fill_poly(img, pts, color)

    m.fillPoly = fillPoly
    m.countNonZero = count_non_zero
    m.__oracle_restatement__ = "OpenCV 4.2.0 drawing.cpp (oracle/raster_oracle.py)"
    return m


def overlapped_boxes_3d(single_box, box_list):
    """nms.py:29-62 on integer corner arrays: `single_box` [8,3], `box_list`
    [n,8,3] (what np.int32(appr_factor * boxes_3d_to_corners(.)) gives)."""
    single_box = np.asarray(single_box)
    x0_max, y0_max, z0_max = (int(v) for v in np.max(single_box, axis=0))
    x0_min, y0_min, z0_min = (int(v) for v in np.min(single_box, axis=0))
    overlap = np.zeros(len(box_list))
    for i, box in enumerate(box_list):
        box = np.asarray(box)
        x_max, y_max, z_max = (int(v) for v in np.max(box, axis=0))
        x_min, y_min, z_min = (int(v) for v in np.min(box, axis=0))
        if x0_max < x_min or x0_min > x_max:
            continue
        if y0_max < y_min or y0_min > y_max:
            continue
        if z0_max < z_min or z0_min > z_max:
            continue
        x_draw_min, x_draw_max = min(x0_min, x_min), max(x0_max, x_max)
        z_draw_min, z_draw_max = min(z0_min, z_min), max(z0_max, z_max)
        offset = np.array([x_draw_min, z_draw_min])
        buf1 = np.zeros((z_draw_max - z_draw_min, x_draw_max - x_draw_min),
                        dtype=np.int32)
        buf2 = np.zeros_like(buf1)
        fill_poly(buf1, [single_box[:4, [0, 2]] - offset], 1)
        fill_poly(buf2, [box[:4, [0, 2]] - offset], 1)
        shared_area = count_non_zero(buf1 * buf2)
        area1 = count_non_zero(buf1)
        area2 = count_non_zero(buf2)
        shared_y = min(y_max, y0_max) - max(y_min, y0_min)
        intersection = shared_y * shared_area
        union = (y_max - y_min) * area2 + (y0_max - y0_min) * area1
        with np.errstate(divide="ignore", invalid="ignore"):
            overlap[i] = np.float64(np.float32(intersection)) / \
                np.float64(union - intersection)
    return overlap
