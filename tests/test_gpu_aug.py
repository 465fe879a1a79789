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
         (2, ("rot", "flip", "shift")), (3, ("flip", "rot")), (7, ("shift",))]
KW = {
    "rot": dict(method_name='normal', yaw_std=0.39269908169872414,
                expend_factor=(1.0, 1.0, 1.0)),
    "flip": dict(flip_prob=0.5),
    "shift": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                  max_overlap_num_allowed=100, max_overlap_rate=None,
                  max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
}
NAMES = {"rot": "random_rotation_all", "flip": "random_flip_all",
         "shift": "random_box_shift"}


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
    if pipeline[0] == "shift":
        # no rotation first: the reference's cloud is still float32 and a point
        # inside two boxes is rounded between its two shifts, here once at the
        # end -- an absolute float32 rounding of the intermediate (<= 70 m)
        assert np.abs(got - ref).max() <= 4e-6 and np.mean(ulp == 0) > 0.99
    else:
        assert ulp.max() <= 1.0
    # the caller's float32 cloud is untouched
    assert np.array_equal(xyz, _scene(seed)[0])


def test_box_shift_with_overlap_rate_keeps_boxes_apart():
    """The overlap-rate test (exact polygon overlap here, cv2 raster in the
    reference: not pinned) -- properties: objects end up pairwise separated,
    every object's points move with it, DontCare goes last."""
    import torch
    from pointgnn_amd import preprocess as PP, nms
    from pointgnn_amd.kitti_dataset import Points
    xyz, attr, labels = _scene(11)
    np.random.seed(11)
    kw = dict(KW["shift"], max_overlap_rate=0.01)
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
            ov = nms.overlapped_boxes_3d_fast_poly(
                np.array(box(b)), np.array([box(l) for l in moved[:i]]),
                appr_factor=10.0)
            assert np.all(ov < 0.01)
    assert out.xyz.shape == (len(xyz), 3)


def test_unimplemented_augmentations_raise():
    from pointgnn_amd import preprocess as PP
    assert PP.get_data_aug([])("p", "l") == ("p", "l")
    with pytest.raises(NotImplementedError):
        PP.aug_method_map['random_jitter'](None, None)
