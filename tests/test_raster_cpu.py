"""oracle/raster_oracle.py (the cv2.fillPoly restatement behind
nms.overlapped_boxes_3d, nms.py:29-62) against properties cv2.fillPoly has in
every OpenCV version, against the reference's own nms.overlapped_boxes_3d
running on the stand-in, and against the committed fixture."""
import os
import sys

import numpy as np
import pytest

from oracle import raster_oracle as RO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference"


def _fill(pts, h, w):
    return RO.fill_poly(np.zeros((h, w), np.int32), [np.asarray(pts)], 1)


def test_axis_aligned_rectangles_fill_inclusively():
    # cv2.fillPoly([[x0,y0],[x1,y0],[x1,y1],[x0,y1]]) covers (x1-x0+1)(y1-y0+1)
    rng = np.random.default_rng(0)
    for _ in range(200):
        x0, y0 = rng.integers(0, 20, 2)
        x1, y1 = x0 + rng.integers(0, 15), y0 + rng.integers(0, 15)
        img = _fill([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], 40, 40)
        want = np.zeros((40, 40), np.int32)
        want[y0:y1 + 1, x0:x1 + 1] = 1
        assert np.array_equal(img, want)


def test_clipped_rectangle_and_outside_polygon():
    img = _fill([[-5, -5], [7, -5], [7, 3], [-5, 3]], 10, 10)
    want = np.zeros((10, 10), np.int32)
    want[0:4, 0:8] = 1
    assert np.array_equal(img, want)
    assert _fill([[20, 20], [30, 20], [30, 30], [20, 30]], 10, 10).sum() == 0
    assert _fill([[-9, 2], [-2, 2], [-2, 8], [-9, 8]], 10, 10).sum() == 0
    # the reference's buffers exclude the maximal row / column (nms.py:49-51)
    img = _fill([[0, 0], [6, 0], [6, 4], [0, 4]], 4, 6)
    assert img.all() and img.shape == (4, 6)


def test_degenerate_polygons_draw_their_outline():
    assert _fill([[3, 3]] * 4, 8, 8).sum() == 1
    img = _fill([[1, 1], [6, 1], [6, 1], [1, 1]], 8, 8)     # a horizontal segment
    assert img.sum() == 6 and img[1, 1:7].all()
    img = _fill([[2, 0], [2, 7], [2, 7], [2, 0]], 8, 8)     # a vertical segment
    assert img.sum() == 8 and img[:, 2].all()
    img = _fill([[0, 0], [7, 7], [7, 7], [0, 0]], 8, 8)     # a diagonal
    assert np.array_equal(img, np.eye(8, dtype=np.int32))


def test_line_is_symmetric_and_8_connected():
    rng = np.random.default_rng(1)
    for _ in range(300):
        p, q = rng.integers(0, 30, 2), rng.integers(0, 30, 2)
        a = RO.line_pixels(30, 30, p, q)
        b = RO.line_pixels(30, 30, q, p)
        assert sorted(a) == sorted(b)              # left_to_right canonical form
        assert len(a) == max(abs(int(p[0] - q[0])), abs(int(p[1] - q[1]))) + 1
        assert tuple(p) in a and tuple(q) in a
        for (x0, y0), (x1, y1) in zip(a[:-1], a[1:]):
            assert max(abs(x1 - x0), abs(y1 - y0)) == 1


def test_fill_is_invariant_under_vertex_rotation_and_reversal():
    rng = np.random.default_rng(2)
    for t in range(150):
        c = rng.uniform(5, 35, 2)
        l, w = rng.uniform(3, 25, 2)
        a = rng.uniform(0, np.pi)
        rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        p = (np.array([[l, w], [l, -w], [-l, -w], [-l, w]]) / 2 @ rot.T
             + c).astype(np.int32)
        h, w_ = rng.integers(20, 45, 2)
        ref = _fill(p, h, w_)
        for k in range(4):
            assert np.array_equal(_fill(np.roll(p, k, 0), h, w_), ref)
            assert np.array_equal(_fill(np.roll(p, k, 0)[::-1], h, w_), ref)


def test_convex_fill_is_outline_plus_interior():
    """Every pixel strictly inside a convex polygon (by exact integer half-plane
    tests) is set, every set pixel is inside or within one pixel of the
    boundary, and all four vertices are set."""
    rng = np.random.default_rng(3)
    for t in range(100):
        c = rng.uniform(12, 28, 2)
        l, w = rng.uniform(4, 20, 2)
        a = rng.uniform(0, np.pi)
        rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        p = (np.array([[l, w], [l, -w], [-l, -w], [-l, w]]) / 2 @ rot.T
             + c).astype(np.int64)
        img = _fill(p, 40, 40)
        ys, xs = np.mgrid[0:40, 0:40]
        cross = []
        for i in range(4):
            (x0, y0), (x1, y1) = p[i], p[(i + 1) % 4]
            cross.append((x1 - x0) * (ys - y0) - (y1 - y0) * (xs - x0))
        cross = np.array(cross)
        strictly_in = np.all(cross > 0, 0) | np.all(cross < 0, 0)
        assert img[strictly_in].all()
        # distance of set pixels to the polygon: at most one pixel outside
        edge_len = np.array([np.hypot(*(p[(i + 1) % 4] - p[i]))
                             for i in range(4)])[:, None, None]
        sign = 1 if np.all(cross[:, int(c[1]), int(c[0])] >= 0) else -1
        outside_by = np.max(-sign * cross / np.maximum(edge_len, 1e-9), 0)
        assert outside_by[img > 0].max() <= 1.0
        for x, y in p:
            assert img[y, x] == 1


def test_span_fill_stays_between_the_outline_pixels():
    """FillEdgeCollection rounds the left end of a span UP and the right end
    DOWN (x1 = (x + XY_ONE - 1) >> XY_SHIFT, x2 = x >> XY_SHIFT), so on every
    row of a convex polygon the filled pixels lie between the leftmost and the
    rightmost outline pixel of that row; flooring the left end sets the pixel
    OUTSIDE a slanted left side whenever the crossing's fraction is >= 0.5."""
    rng = np.random.default_rng(11)
    n_rows = 0
    for t in range(200):
        c = rng.uniform(12, 28, 2)
        l, w = rng.uniform(4, 20, 2)
        a = rng.uniform(0, np.pi)
        rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
        p = (np.array([[l, w], [l, -w], [-l, -w], [-l, w]]) / 2 @ rot.T
             + c).astype(np.int64)
        img = _fill(p, 40, 40)
        outline = np.zeros_like(img)
        for i in range(4):
            for (x, y) in RO.line_pixels(40, 40, tuple(p[i]),
                                         tuple(p[(i + 1) % 4])):
                outline[y, x] = 1
        assert np.all(img[outline > 0] == 1)
        for y in range(40):
            xs = np.nonzero(img[y])[0]
            if len(xs) == 0:
                assert outline[y].sum() == 0
                continue
            ox = np.nonzero(outline[y])[0]
            assert xs[0] == ox[0] and xs[-1] == ox[-1], (t, y)
            n_rows += 1
    assert n_rows > 2000


def test_fixture_reproduces():
    fix = np.load(os.path.join(GOLD, "raster_overlap.npz"))
    for tag in ("a10", "a100", "quad"):
        with np.errstate(all="ignore"):
            got = RO.overlapped_boxes_3d(fix[tag + "_single"],
                                         fix[tag + "_list"])
        assert np.array_equal(got, fix[tag + "_overlap"], equal_nan=True)
    assert np.sum(fix["a10_overlap"] > 0) > 50


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")),
                    reason="reference tree only exists in the build container")
def test_restated_overlap_equals_reference_function_on_the_stand_in():
    """oracle.overlapped_boxes_3d (restatement of nms.py:29-62) == the
    reference's nms.overlapped_boxes_3d with `cv2` = the fillPoly stand-in."""
    import types
    saved = {k: sys.modules.get(k) for k in ("cv2", "shapely",
                                             "shapely.geometry", "models",
                                             "models.nms")}
    shp, geo = types.ModuleType("shapely"), types.ModuleType("shapely.geometry")
    geo.Polygon = object
    shp.geometry = geo
    sys.modules.update({"cv2": RO.cv2_stub(), "shapely": shp,
                        "shapely.geometry": geo})
    sys.modules.pop("models.nms", None)
    sys.modules.pop("models", None)
    sys.path.insert(0, REF)
    try:
        from models import nms as ref_nms
        fix = np.load(os.path.join(GOLD, "raster_overlap.npz"))
        for tag in ("a10", "quad"):
            with np.errstate(all="ignore"):
                want = ref_nms.overlapped_boxes_3d(fix[tag + "_single"],
                                                   fix[tag + "_list"])
                got = RO.overlapped_boxes_3d(fix[tag + "_single"],
                                             fix[tag + "_list"])
            assert np.array_equal(got, want, equal_nan=True)
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
