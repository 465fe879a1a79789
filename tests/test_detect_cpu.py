"""Detection post-processing oracle (oracle/detect_oracle.py) against the
fixtures produced by the reference's own code (tests/golden/
make_golden_detect.py) and against closed-form geometry.  No GPU."""
import os

import numpy as np
import pytest

from oracle import detect_oracle as DO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LABEL_MAPS = {
    "car": {'Background': 0, 'Car': 1, 'DontCare': 3},
    "ped": {'Background': 0, 'Pedestrian': 1, 'Cyclist': 3, 'DontCare': 5},
}
MODES = ("plain", "uncertainty", "merge_only", "score_only")
N_CASES = 5


@pytest.fixture(scope="module")
def codec():
    return np.load(os.path.join(GOLD, "detect_codec.npz"))


@pytest.fixture(scope="module")
def nmsfix():
    return np.load(os.path.join(GOLD, "detect_nms.npz"))


@pytest.mark.parametrize("name", ["car", "ped"])
def test_codec_oracle_equals_reference(codec, name):
    lm = LABEL_MAPS[name]
    g = lambda k: codec["codec_%s_%s" % (name, k)]  # noqa: E731
    dec = DO.box_decoding(g("labels"), g("xyz"), g("encoded"), lm)
    assert dec.dtype == np.float32
    assert np.array_equal(dec, g("decoded"))
    enc = DO.box_encoding(g("labels"), g("xyz"), g("boxes"), lm)
    assert np.array_equal(enc, g("boxes_encoded"))
    # encode -> decode round trip (box_encoding.py:419-442 tests the same)
    back = DO.box_decoding(g("labels"), g("xyz"), enc, lm)
    np.testing.assert_allclose(back, g("boxes"), rtol=2e-5, atol=2e-5)


def test_corners_and_overlap_equal_reference(nmsfix):
    boxes = nmsfix["geom_boxes"]
    corners = DO.boxes_3d_to_corners(boxes)
    np.testing.assert_allclose(corners, nmsfix["geom_corners"], rtol=0,
                               atol=1e-12)
    ov = DO.overlapped_boxes_3d_fast_poly(corners[0], corners[1:])
    np.testing.assert_allclose(ov, nmsfix["geom_overlap_row0"], rtol=1e-12,
                               atol=0)


@pytest.mark.parametrize("case", range(N_CASES))
@pytest.mark.parametrize("mode", MODES)
def test_nms_oracle_equals_reference(nmsfix, case, mode):
    thres, top_k = nmsfix["nms%d_params" % case]
    lab, box, sco, att = DO.nms_boxes_3d(
        nmsfix["nms%d_labels" % case], nmsfix["nms%d_boxes" % case],
        nmsfix["nms%d_scores" % case], overlapped_thres=thres, mode=mode,
        appr_factor=100.0, top_k=int(top_k))
    pre = "nms%d_%s_" % (case, mode)
    assert np.array_equal(att, nmsfix[pre + "attrs"])
    assert np.array_equal(lab, nmsfix[pre + "labels"])
    assert np.array_equal(box, nmsfix[pre + "boxes"])
    np.testing.assert_allclose(sco, nmsfix[pre + "scores"], rtol=1e-6)
    assert len(att) < len(nmsfix["nms%d_labels" % case])


def _rect(cx, cz, l, w, yaw):
    box = np.array([[cx, 0.0, cz, l, 1.0, w, yaw]], np.float64)
    return DO.ConvexPolygon(DO.boxes_3d_to_corners(box)[0][:4][:, [0, 2]])


def test_polygon_intersection_closed_forms():
    unit = _rect(0, 0, 2, 2, 0.0)
    assert unit.area == pytest.approx(4.0)
    # axis-aligned shift: overlap is a rectangle
    for dx, dz in [(0.5, 0.0), (1.0, 1.0), (1.5, -0.25), (0.0, 0.0)]:
        other = _rect(dx, dz, 2, 2, 0.0)
        want = max(0.0, 2 - abs(dx)) * max(0.0, 2 - abs(dz))
        assert unit.intersection(other).area == pytest.approx(want, abs=1e-12)
    # disjoint and corner-touching
    assert unit.intersection(_rect(5, 5, 2, 2, 0.3)).area == 0.0
    assert unit.intersection(_rect(2, 2, 2, 2, 0.0)).area == \
        pytest.approx(0.0, abs=1e-12)
    # contained
    assert unit.intersection(_rect(0.1, -0.2, 0.5, 0.3, 0.7)).area == \
        pytest.approx(0.15, abs=1e-12)
    # square turned by 45 degrees about the same centre: regular octagon
    s = 2.0
    octagon = 2 * (np.sqrt(2) - 1) * s * s
    assert unit.intersection(_rect(0, 0, s, s, np.pi / 4)).area == \
        pytest.approx(octagon, rel=1e-12)
    # symmetric in its arguments, any orientation of the rings
    a, b = _rect(0.3, 0.1, 3.9, 1.6, 0.4), _rect(-0.2, 0.5, 4.1, 1.7, -1.1)
    ab, ba = a.intersection(b).area, b.intersection(a).area
    assert ab == pytest.approx(ba, rel=1e-12) and 0 < ab < min(a.area, b.area)
    rev = DO.ConvexPolygon(b.pts[::-1])
    assert a.intersection(rev).area == pytest.approx(ab, rel=1e-12)
    # Monte-Carlo cross-check of a generic pair
    rng = np.random.default_rng(0)
    q = rng.uniform(-3, 3, (400000, 2))
    inside = np.array([a._contains(p) and b._contains(p) for p in q[:40000]])
    assert inside.mean() * 36.0 == pytest.approx(ab, rel=0.05)


def test_overlap_is_iou_for_aligned_boxes():
    # two axis-aligned boxes: overlap = intersection / (union) in 3D
    b = np.array([[0, 0, 0, 2, 2, 2, 0.0], [1, -1, 0, 2, 2, 2, 0.0]],
                 np.float32)
    c = DO.boxes_3d_to_corners(b)
    ov = DO.overlapped_boxes_3d_fast_poly(c[0], c[1:])
    # x overlap 1, z overlap 2, y overlap 1 (y spans [y-h, y])
    inter = 1 * 2 * 1
    assert ov[0] == pytest.approx(inter / (8 + 8 - inter), rel=1e-6)


def test_select_candidates_rule():
    probs = np.array([[0.1, 0.5, 0.3, 0.1],
                      [0.7, 0.1, 0.1, 0.1],
                      [0.0, 0.25, 0.26, 0.49]], np.float32)
    idx, lab = DO.select_candidates(probs)
    # class 0 (background) and class nc-1 (don't care) never qualify; 0.25 is
    # not > 1/4
    assert idx.tolist() == [1, 2, 10]
    assert lab.tolist() == [1, 1, 1]


def test_kitti_label_tuples_oracle_equals_reference_composition():
    """run.py:360-412 restated (oracle) vs the fixture composed from the
    reference's own functions."""
    from oracle import ingest_oracle as IO
    fix = np.load(os.path.join(GOLD, "detect_output.npz"))
    calib = IO.get_calib(IO.CALIB_LINES)
    rows = DO.kitti_labels(fix["labels"], fix["boxes"], fix["scores"],
                           calib['cam_to_image'], 'Car', fix["cand_xyz"])
    assert len(rows) == len(fix["rows"]) < len(fix["boxes"])
    assert [r[0] for r in rows] == list(fix["names"])
    got = np.array([r[4:] for r in rows], np.float64)
    np.testing.assert_allclose(got, fix["rows"][:, :12], rtol=1e-12)
    assert np.any(got[:, 11] > fix["rows"][:, 12] * 1.0001)   # rescored
