#!/usr/bin/env python
"""Generate tests/golden/aug_kitti.npz by running the reference's own
augmentation functions (models/preprocess.py) in the build container, with
NumPy's global RNG seeded, on seeded synthetic points and labels
(oracle.labels_oracle).  open3d / cv2 are stubbed with empty modules and
shapely.geometry.Polygon with oracle.detect_oracle.ConvexPolygon.
random_box_shift runs with the shipped configs' kwargs (max_overlap_rate=0.01,
configs/*_train_config), which routes every trial through
nms.overlapped_boxes_3d -> cv2.fillPoly / cv2.countNonZero
(preprocess.py:281-301, nms.py:29-62).  cv2 is not installable here, so the
`cv2` module the reference imports is oracle.raster_oracle.cv2_stub(): the
fillPoly algorithm of OpenCV 4.2 (drawing.cpp) restated in integers.  The
reference's own preprocess.py and nms.py run unmodified on top of it.

    python tests/golden/make_golden_aug.py
"""
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import detect_oracle as DO  # noqa: E402
from oracle import labels_oracle as LO  # noqa: E402
from oracle import raster_oracle as RO  # noqa: E402

LABEL_KEYS = ('x3d', 'y3d', 'z3d', 'yaw', 'length', 'height', 'width')
CASES = [  # (seed, pipeline)
    (0, ("rot", "flip", "shift")), (1, ("rot", "flip", "shift")),
    (2, ("rot", "flip", "shift")), (3, ("flip", "rot")), (7, ("shift",)),
    (4, ("rot", "shift100")), (5, ("flip", "shift_none")),
    (6, ("shift_dense",)),
]
KW = {
    "rot": dict(method_name='normal', yaw_std=0.39269908169872414,
                expend_factor=(1.0, 1.0, 1.0)),
    "flip": dict(flip_prob=0.5),
    # car_auto_T3_train_train_config:27-39, as shipped
    "shift": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                  max_overlap_num_allowed=100, max_overlap_rate=0.01,
                  max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
    # finer raster, looser rate
    "shift100": dict(appr_factor=100, expend_factor=(1.1, 1.1, 1.1),
                     max_overlap_num_allowed=100, max_overlap_rate=0.05,
                     max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
    # the branch without the raster test
    "shift_none": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                       max_overlap_num_allowed=100, max_overlap_rate=None,
                       max_trails=100, method_name='normal',
                       xyz_std=(3, 0, 3)),
    # small moves in a crowded scene: most trials are decided by the raster
    "shift_dense": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                        max_overlap_num_allowed=100000, max_overlap_rate=0.01,
                        max_trails=100, method_name='normal',
                        xyz_std=(0.7, 0, 0.7)),
}


def scene(seed):
    xyz = LO.synthetic_vertices(seed, k=5000)
    labels = LO.synthetic_labels(seed, xyz, n_boxes=18)
    attr = np.random.default_rng(seed).uniform(0, 1, (len(xyz), 1)
                                               ).astype(np.float32)
    return xyz, attr, labels


def main():
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    sys.modules["cv2"] = RO.cv2_stub()
    shp = types.ModuleType("shapely")
    geo = types.ModuleType("shapely.geometry")
    geo.Polygon = DO.ConvexPolygon
    shp.geometry = geo
    sys.modules["shapely"], sys.modules["shapely.geometry"] = shp, geo
    sys.path.insert(0, REF)
    try:
        from models import preprocess
        from dataset.kitti_dataset import Points
    finally:
        sys.path.remove(REF)
    from models import nms as ref_nms
    assert ref_nms.cv2 is sys.modules["cv2"]
    calls = {"n": 0, "pairs": 0, "hits": 0}
    real_overlap = ref_nms.overlapped_boxes_3d

    def counting_overlap(single_box, box_list):
        ov = real_overlap(single_box, box_list)
        calls["n"] += 1
        calls["pairs"] += len(box_list)
        calls["hits"] += int(np.sum(ov > 0))
        return ov
    preprocess.overlapped_boxes_3d = counting_overlap
    fns = {"rot": preprocess.random_rotation_all,
           "flip": preprocess.random_flip_all,
           "shift": preprocess.random_box_shift,
           "shift100": preprocess.random_box_shift,
           "shift_none": preprocess.random_box_shift,
           "shift_dense": preprocess.random_box_shift}
    out = {}
    for seed, pipeline in CASES:
        xyz, attr, labels = scene(seed)
        pts = Points(xyz=xyz.copy(), attr=attr)
        labs = copy.deepcopy(labels)
        np.random.seed(seed)
        for step in pipeline:
            pts, labs = fns[step](pts, labs, **KW[step])
        pre = "case%d_" % seed
        out[pre + "xyz"] = np.asarray(pts.xyz).astype(np.float32)
        out[pre + "names"] = np.array([l['name'] for l in labs])
        out[pre + "labels"] = np.array([[l[k] for k in LABEL_KEYS]
                                        for l in labs], np.float64)
        out[pre + "rng_after"] = np.array(np.random.uniform())
        moved = np.any(out[pre + "xyz"] != xyz, axis=1).mean()
        print(seed, pipeline, "points changed %.3f" % moved,
              "raster calls so far", calls)
    np.savez_compressed(os.path.join(HERE, "aug_kitti.npz"), **out)

    # raster_overlap.npz: nms.overlapped_boxes_3d itself (the reference's
    # function on the cv2 stand-in) on integer corner arrays: boxes of
    # car / pedestrian size around a common centre at both corner scales the
    # configs and the function's default use, plus arbitrary quadrilaterals
    # (self-intersecting, degenerate) as footprints.
    rng = np.random.default_rng(11)
    ro = {}
    for tag, appr, n in (("a10", 10, 160), ("a100", 100, 40)):
        boxes = np.empty((n + 1, 7))
        boxes[:, 0] = rng.uniform(-3, 3, n + 1)
        boxes[:, 1] = rng.uniform(1.0, 2.0, n + 1)
        boxes[:, 2] = rng.uniform(17, 23, n + 1)
        small = rng.uniform(size=n + 1) < 0.4
        boxes[:, 3] = np.where(small, rng.uniform(0.3, 1.2, n + 1),
                               rng.uniform(2.5, 5.0, n + 1))
        boxes[:, 4] = rng.uniform(1.2, 2.0, n + 1)
        boxes[:, 5] = np.where(small, rng.uniform(0.3, 0.9, n + 1),
                               rng.uniform(1.4, 2.2, n + 1))
        boxes[:, 6] = rng.uniform(-np.pi, np.pi, n + 1)
        corners = np.int32(appr * ref_nms.boxes_3d_to_corners(boxes))
        ro[tag + "_single"] = corners[0]
        ro[tag + "_list"] = corners[1:]
        ro[tag + "_overlap"] = real_overlap(corners[0], corners[1:])
    quads = rng.integers(0, 40, (121, 8, 3)).astype(np.int32)
    quads[:, 4:] = quads[:, :4]
    quads[:, :4, 1] = 0
    quads[:, 4:, 1] = -rng.integers(1, 20, (121, 1))
    with np.errstate(all="ignore"):
        ro["quad_single"], ro["quad_list"] = quads[0], quads[1:]
        ro["quad_overlap"] = real_overlap(quads[0], quads[1:])
    for k in ("a10", "a100", "quad"):
        ov = ro[k + "_overlap"]
        print(k, "pairs", len(ov), "overlapping", int(np.sum(ov > 0)),
              "max %.3f" % np.nanmax(ov))
    np.savez_compressed(os.path.join(HERE, "raster_overlap.npz"), **ro)


if __name__ == "__main__":
    main()
