#!/usr/bin/env python
"""Generate tests/golden/aug_kitti.npz by running the reference's own
augmentation functions (models/preprocess.py) in the build container, with
NumPy's global RNG seeded, on seeded synthetic points and labels
(oracle.labels_oracle).  open3d / cv2 are stubbed with empty modules and
shapely.geometry.Polygon with oracle.detect_oracle.ConvexPolygon -- none of them
is reached by the calls below: random_box_shift runs with
max_overlap_rate=None, the only mode that does not go through cv2.fillPoly
(nms.overlapped_boxes_3d), which this image cannot run.

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

LABEL_KEYS = ('x3d', 'y3d', 'z3d', 'yaw', 'length', 'height', 'width')
CASES = [  # (seed, pipeline)
    (0, ("rot", "flip", "shift")), (1, ("rot", "flip", "shift")),
    (2, ("rot", "flip", "shift")), (3, ("flip", "rot")), (7, ("shift",)),
]
KW = {
    "rot": dict(method_name='normal', yaw_std=0.39269908169872414,
                expend_factor=(1.0, 1.0, 1.0)),
    "flip": dict(flip_prob=0.5),
    # car_auto_T3_train_train_config minus max_overlap_rate (needs cv2)
    "shift": dict(appr_factor=10, expend_factor=(1.1, 1.1, 1.1),
                  max_overlap_num_allowed=100, max_overlap_rate=None,
                  max_trails=100, method_name='normal', xyz_std=(3, 0, 3)),
}


def scene(seed):
    xyz = LO.synthetic_vertices(seed, k=5000)
    labels = LO.synthetic_labels(seed, xyz, n_boxes=18)
    attr = np.random.default_rng(seed).uniform(0, 1, (len(xyz), 1)
                                               ).astype(np.float32)
    return xyz, attr, labels


def main():
    for name in ("open3d", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
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
    fns = {"rot": preprocess.random_rotation_all,
           "flip": preprocess.random_flip_all,
           "shift": preprocess.random_box_shift}
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
        print(seed, pipeline, "points changed %.3f" % moved)
    np.savez_compressed(os.path.join(HERE, "aug_kitti.npz"), **out)


if __name__ == "__main__":
    main()
