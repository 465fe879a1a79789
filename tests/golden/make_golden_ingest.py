#!/usr/bin/env python
"""Generate tests/golden/ingest_kitti.npz by running the reference's own
`KittiDataset` methods (dataset/kitti_dataset.py) in the build container.

The module imports open3d and cv2 at the top; neither is used by the methods
on this slice, so both are stubbed with empty modules.  A `KittiDataset` is
created without `__init__` (which walks real KITTI directories); `get_calib`
reads a calib .txt we write to a temp dir, `velo_points_to_cam`,
`cam_points_to_image` and `rgb_to_cam_points` are called unmodified, and the
two mask lines of `get_cam_points_in_image_with_rgb` (:700-712) are applied
in between exactly as that method does (it cannot be called directly: it
reads the PNG through cv2).

    python tests/golden/make_golden_ingest.py
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

from oracle import ingest_oracle as IO  # noqa: E402


def reference_dataset_module():
    for name in ("open3d", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    try:
        from dataset import kitti_dataset
    finally:
        sys.path.remove(REF)
    return kitti_dataset


def main():
    kd = reference_dataset_module()
    ds = object.__new__(kd.KittiDataset)
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "000000.txt"), "w") as f:
            f.writelines(IO.CALIB_LINES)
        ds._calib_dir = tmp
        ds._file_list = ["000000"]
        calib = ds.get_calib(0)
    out = {}
    for k in ("P2", "R0_rect", "Tr_velo_to_cam", "velo_to_rect",
              "cam_to_image", "rect_to_cam", "velo_to_cam", "cam_to_velo",
              "velo_to_image"):
        out["calib_" + k] = calib[k]
    velo = IO.synthetic_velo_scan(0, n=60000)
    image = IO.synthetic_image(0)
    height, width = image.shape[:2]
    cam = ds.velo_points_to_cam(
        kd.Points(xyz=velo[:, :3], attr=velo[:, [3]]), calib)
    front = cam.xyz[:, 2] > 0.1
    front_pts = kd.Points(cam.xyz[front, :], cam.attr[front, :])
    img = ds.cam_points_to_image(front_pts, calib)
    inside = np.logical_and.reduce(
        [img.xyz[:, 0] > 0, img.xyz[:, 0] < width,
         img.xyz[:, 1] > 0, img.xyz[:, 1] < height])
    in_img = kd.Points(xyz=front_pts.xyz[inside, :],
                       attr=front_pts.attr[inside, :])
    with_rgb = ds.rgb_to_cam_points(in_img, image, calib)
    # inputs are seeded (oracle.ingest_oracle.synthetic_*): store their digests
    # instead of 2.4 MB of random bytes; the tests regenerate and verify them
    import hashlib
    out["velo_sha1"] = np.frombuffer(
        hashlib.sha1(velo.tobytes()).digest(), np.uint8)
    out["image_sha1"] = np.frombuffer(
        hashlib.sha1(image.tobytes()).digest(), np.uint8)
    out["cam_every16"] = cam.xyz[::16]
    out["xyz"] = in_img.xyz
    out["attr_rgb"] = with_rgb.attr
    out["kept"] = np.arange(len(velo), dtype=np.int32)[front][inside]
    print("scan", len(velo), "front", int(front.sum()), "in image",
          len(in_img.xyz))
    np.savez_compressed(os.path.join(HERE, "ingest_kitti.npz"), **out)


if __name__ == "__main__":
    main()
