#!/usr/bin/env python
"""Generate tests/golden/labels_kitti.npz by running the reference's own
label / target code (dataset/kitti_dataset.py, models/box_encoding.py) in the
build container: `get_label` on a label file we write, the module-level
`box3d_to_normals` / `sel_xyz_in_box3d`, the three
`assign_classaware_*_label_to_points` methods, and the float64 encode + float32
cast of train.py:120-130.  open3d / cv2 are stubbed (unused on this slice); a
`KittiDataset` is created without `__init__`.

    python tests/golden/make_golden_labels.py
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import labels_oracle as LO  # noqa: E402


def main():
    for name in ("open3d", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    try:
        from dataset import kitti_dataset as kd
        from models import box_encoding
    finally:
        sys.path.remove(REF)
    out = {}
    xyz = LO.synthetic_vertices(0)
    labels = LO.synthetic_labels(0, xyz, n_boxes=60)
    ds = object.__new__(kd.KittiDataset)
    with tempfile.TemporaryDirectory() as tmp:
        LO.write_label_file(os.path.join(tmp, "000000.txt"), labels)
        ds._label_dir = tmp
        ds._file_list = ["000000"]
        for diff in (-100, 0, 1, 2):
            ds.difficulty = diff
            got = ds.get_label(0)
            out["label_count_d%d" % max(diff, -1)] = np.array(len(got))
            if diff == -100:
                assert len(got) == len(labels)
                for a, b in zip(got, labels):
                    assert a == b, (a, b)
    out["xyz_seed"] = np.array(0)
    for i in (0, 5, 11):
        n, lo, up = kd.box3d_to_normals(labels[i], (1.1, 1.2, 1.3))
        out["normals_%d" % i] = n
        out["lower_%d" % i] = lo
        out["upper_%d" % i] = up
        out["mask_%d" % i] = kd.sel_xyz_in_box3d(labels[i], xyz,
                                                 (1.1, 1.2, 1.3))
    methods = {"yaw": (8, "assign_classaware_label_to_points"),
               "Car": (4, "assign_classaware_car_label_to_points"),
               "Pedestrian_and_Cyclist": (
                   6, "assign_classaware_ped_and_cyc_label_to_points")}
    for method, (nc, fn) in methods.items():
        ds.num_classes = nc
        for tag, expend in (("e1", (1.0, 1.0, 1.0)), ("e2", (1.1, 1.1, 1.1))):
            cls, boxes, valid, lm = getattr(ds, fn)(labels, xyz, expend)
            enc = box_encoding.classaware_all_class_box_encoding(
                cls, xyz, boxes, lm).astype(np.float32)
            pre = "%s_%s_" % (method, tag)
            out[pre + "cls"] = cls
            out[pre + "boxes"] = boxes
            out[pre + "valid"] = valid
            out[pre + "encoded"] = enc
            print(method, tag, "labelled", int((cls > 0).sum()), "valid",
                  int(valid.sum()))
    np.savez_compressed(os.path.join(HERE, "labels_kitti.npz"), **out)


if __name__ == "__main__":
    main()
