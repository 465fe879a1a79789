"""Training augmentations (pointgnn_amd.preprocess; per-point work in
pgnn_points_affine_f64 / pgnn_points_in_box_f64) against fixtures written by
the reference's own models/preprocess.py with NumPy's RNG seeded.

Bars: the same random decisions (the global RNG ends in the same state),
identical label lists, augmented points within 1 float32 ulp of the
reference's (float64 matmul order; 0 ulp expected almost everywhere)."""
import copy
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from oracle import labels_oracle as LO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LABEL_KEYS = ('x3d', 'y3d', 'z3d', 'yaw', 'length', 'height', 'width')
CASES = [(0, ("rot", "flip", "shift")), (1, ("rot", "flip", "shift")),
         (2, ("rot", "flip", "shift")), (3, ("flip", "rot")), (7, ("shift",)),
         (4, ("rot", "shift100")), (5, ("flip", "shift_none")),
         (6, ("shift_dense",))]
KW = {
    "rot": dict(method_name='normal', yaw_std=0.39269908169872414,
                expend_factor=(1.0, 1.0, 1.0)),
    "flip": dict(flip_prob=0.5),
    # car_auto_T3_train_train_config:27-39 as shipped: every trial goes through
    # the cv2 raster overlap (nms.overlapped_boxes_3d)
    "shift": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                  max_overlap_num_allowed=100, max_overlap_rate=0.01,
                  max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
    "shift100": dict(appr_factor=100, expend_factor=(1.1, 1.1, 1.1),
                     max_overlap_num_allowed=100, max_overlap_rate=0.05,
                     max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
    "shift_none": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                       max_overlap_num_allowed=100, max_overlap_rate=None,
                       max_trails=100, method_name='normal',
                       xyz_std=(3, 0, 3)),
    "shift_dense": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                        max_overlap_num_allowed=100000, max_overlap_rate=0.01,
                        max_trails=100, method_name='normal',
                        xyz_std=(0.7, 0, 0.7)),
}
NAMES = {"rot": "random_rotation_all", "flip": "random_flip_all",
         "shift": "random_box_shift", "shift100": "random_box_shift",
         "shift_none": "random_box_shift", "shift_dense": "random_box_shift"}


def _scene(seed):
    xyz = LO.synthetic_vertices(seed, k=5000)
    labels = LO.synthetic_labels(seed, xyz, n_boxes=18)
    attr = np.random.default_rng(seed).uniform(0, 1, (len(xyz), 1)
                                               ).astype(np.float32)
    return xyz, attr, labels


@pytest.mark.parametrize("seed,pipeline", CASES)
def test_augmentations_match_reference_fixture(seed, pipeline):
    import torch
    from pointgnn_amd import preprocess as PP
    from pointgnn_amd.kitti_dataset import Points
    fix = np.load(os.path.join(GOLD, "aug_kitti.npz"))
    xyz, attr, labels = _scene(seed)
    aug = PP.get_data_aug([{"method_name": NAMES[s], "method_kwargs": KW[s]}
                           for s in pipeline])
    pts = Points(xyz=torch.from_numpy(xyz).cuda(),
                 attr=torch.from_numpy(attr).cuda())
    labs = copy.deepcopy(labels)
    np.random.seed(seed)
    pts, labs = aug(pts, labs)
    assert pts.xyz.dtype == torch.float64          # like the reference's array
    out = PP.finish(pts)
    assert out.xyz.dtype == torch.float32 and out.attr is pts.attr
    pre = "case%d_" % seed
    assert np.random.uniform() == float(fix[pre + "rng_after"])   # same draws
    assert [l['name'] for l in labs] == list(fix[pre + "names"])
    got_l = np.array([[l[k] for k in LABEL_KEYS] for l in labs], np.float64)
    np.testing.assert_allclose(got_l, fix[pre + "labels"], rtol=1e-13,
                               atol=1e-13)
    got, ref = out.xyz.cpu().numpy(), fix[pre + "xyz"]
    ulp = np.abs(got.astype(np.float64) - ref) / np.spacing(
        np.maximum(np.abs(got), np.abs(ref)))
    print(seed, pipeline, "identical %.5f, max %.2f ulp" % (
        np.mean(ulp == 0), ulp.max()))
    if pipeline[0].startswith("shift"):
        # no rotation first: the reference's cloud is still float32 and a point
        # inside two boxes is rounded between its two shifts, here once at the
        # end -- an absolute float32 rounding of the intermediate (<= 70 m)
        assert np.abs(got - ref).max() <= 4e-6 and np.mean(ulp == 0) > 0.99
    else:
        assert ulp.max() <= 1.0
    # the caller's float32 cloud is untouched
    assert np.array_equal(xyz, _scene(seed)[0])


def test_raster_overlap_matches_reference_fixture_and_oracle():
    """nms.overlapped_boxes_3d (pgnn_overlapped_boxes_3d_raster: cv2.fillPoly's
    pixel counts in closed form per row) == the fixture written by the
    reference's nms.overlapped_boxes_3d (tests/golden/make_golden_aug.py), and
    == the literal restatement on fresh random footprints: rotated boxes,
    arbitrary (self-intersecting, degenerate, collinear) quadrilaterals, far
    apart pairs, single-pixel boxes.  Bit-exact: the counts are integers and
    the quotient is one float32 rounding + one float64 division."""
    from pointgnn_amd import nms
    from oracle import raster_oracle as RO
    fix = np.load(os.path.join(GOLD, "raster_overlap.npz"))
    for tag in ("a10", "a100", "quad"):
        got = nms.overlapped_boxes_3d(fix[tag + "_single"], fix[tag + "_list"])
        assert got.dtype == np.float64
        assert np.array_equal(got, fix[tag + "_overlap"], equal_nan=True), tag
    assert nms.overlapped_boxes_3d(fix["a10_single"],
                                   np.zeros((0, 8, 3), np.int32)).shape == (0,)
    assert nms.overlapped_boxes_3d(fix["a10_single"], np.array([])).shape == (0,)
    rng = np.random.default_rng(5)
    checked = positive = 0
    for rnd in range(12):
        n = 64
        hi = (6, 12, 40, 90, 200, 400)[rnd % 6]
        q = rng.integers(-hi // 4, hi, (n + 1, 8, 3)).astype(np.int32)
        if rnd % 2:
            # axis-aligned / partly degenerate footprints
            q[:, 1, 2] = q[:, 0, 2]
            q[:, 2, 0] = q[:, 1, 0]
            q[:, 3, 2] = q[:, 2, 2]
            q[:, 3, 0] = q[:, 0, 0]
        q[:, 4:] = q[:, :4]
        q[:, :4, 1] = rng.integers(-3, 3, (n + 1, 1))
        q[:, 4:, 1] = q[:, :4, 1] - rng.integers(0, 25, (n + 1, 1))
        if rnd == 3:
            q[1:9] += 10000          # disjoint in x / z: the early exits
        with np.errstate(all="ignore"):
            want = RO.overlapped_boxes_3d(q[0], q[1:])
        got = nms.overlapped_boxes_3d(q[0], q[1:])
        assert np.array_equal(got, want, equal_nan=True), rnd
        checked += n
        positive += int(np.sum(want > 0))
    print("raster overlap: %d random pairs identical (%d overlapping)" % (
        checked, positive))
    assert positive > 200


def test_box_shift_with_overlap_rate_keeps_boxes_apart():
    """Properties of the shipped configuration on top of the fixture equality
    above: objects end up pairwise separated under the raster overlap, every
    object's points move with it, DontCare goes last."""
    import torch
    from pointgnn_amd import preprocess as PP, nms
    from pointgnn_amd.kitti_dataset import Points
    from oracle import raster_oracle as RO
    xyz, attr, labels = _scene(11)
    np.random.seed(11)
    kw = KW["shift"]
    pts = Points(xyz=torch.from_numpy(xyz).cuda(), attr=None)
    before = copy.deepcopy(labels)
    out, labs = PP.random_box_shift(pts, copy.deepcopy(labels), **kw)
    assert len(labs) == len(before)
    n_dc = sum(l['name'] == 'DontCare' for l in before)
    assert all(l['name'] == 'DontCare' for l in labs[len(labs) - n_dc:])
    moved = [l for l in labs if l['name'] != 'DontCare']
    orig = [l for l in before if l['name'] != 'DontCare']
    shifted = 0
    for a, b in zip(orig, moved):
        assert a['name'] == b['name'] and a['yaw'] == b['yaw']
        assert a['y3d'] == b['y3d']                 # xyz_std = (3, 0, 3)
        shifted += (a['x3d'] != b['x3d'])
    assert shifted >= len(orig) // 2
    # every accepted move respected the rate against the boxes placed before it
    box = lambda l: [l['x3d'], l['y3d'], l['z3d'], l['length'], l['height'],  # noqa: E731
                     l['width'], l['yaw']]
    for i, (a, b) in enumerate(zip(orig, moved)):
        if i and a['x3d'] != b['x3d']:
            c_new = np.int32(10 * nms.boxes_3d_to_corners(np.array([box(b)])))
            c_old = np.int32(10 * nms.boxes_3d_to_corners(
                np.array([box(l) for l in moved[:i]])))
            ov = RO.overlapped_boxes_3d(c_new[0], c_old)
            assert np.all(ov < 0.01)
    assert out.xyz.shape == (len(xyz), 3)


def test_unimplemented_augmentations_raise():
    from pointgnn_amd import preprocess as PP
    assert PP.get_data_aug([])("p", "l") == ("p", "l")
    with pytest.raises(NotImplementedError):
        PP.aug_method_map['random_jitter'](None, None)
