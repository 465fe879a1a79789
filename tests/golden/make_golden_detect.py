#!/usr/bin/env python
"""Generate tests/golden/detect_*.npz by RUNNING THE REFERENCE'S OWN CODE in the
build container (needs /root/reference; the fixtures travel, this script's
imports do not):

  * models/box_encoding.py is pure NumPy and is imported unmodified;
  * models/nms.py imports cv2 and shapely, both absent from the image.  It is
    imported under an empty `cv2` stub (only the raster overlap, which run.py
    never selects, uses it) and a `shapely.geometry` stub whose `Polygon` is
    oracle.detect_oracle.ConvexPolygon.  The scan loops, sorting, median
    merge and score accumulation that produce these fixtures are therefore
    the reference's code; only the polygon-intersection area comes from our
    stand-in (checked against closed forms in tests/test_detect_cpu.py).

    python tests/golden/make_golden_detect.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import detect_oracle as DO  # noqa: E402


def reference_modules():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    shp = types.ModuleType("shapely")
    geo = types.ModuleType("shapely.geometry")
    geo.Polygon = DO.ConvexPolygon
    shp.geometry = geo
    sys.modules["shapely"] = shp
    sys.modules["shapely.geometry"] = geo
    sys.path.insert(0, REF)
    try:
        from models import box_encoding, nms
    finally:
        sys.path.remove(REF)
    return box_encoding, nms


LABEL_MAPS = {
    # run.py:244-250
    "car": {'Background': 0, 'Car': 1, 'DontCare': 3},
    "ped": {'Background': 0, 'Pedestrian': 1, 'Cyclist': 3, 'DontCare': 5},
}


# (seed, overlapped_thres, top_k, scene): the last two scenes pack the objects
# so that vote clusters of neighbouring objects chain into each other
CASES = [(0, 0.01, -1, {}), (1, 0.25, -1, {}), (2, 0.01, 120, {}),
         (3, 0.01, -1, dict(n_objects=25, half_width=7.0, depth=(5.0, 22.0))),
         (4, 0.1, -1, dict(n_objects=40, half_width=6.0, depth=(5.0, 18.0),
                           votes=(2, 12), noise_boxes=60))]


def main():
    box_encoding, nms = reference_modules()
    out = {}
    # ---- codec (box_encoding.py:231-299)
    for name, lm in LABEL_MAPS.items():
        rng = np.random.default_rng(len(name))
        nlab = max(lm.values()) + 1
        r = 500
        labels = rng.integers(0, nlab, (r, 1)).astype(np.int32)
        xyz = (rng.uniform(-40, 40, (r, 3))).astype(np.float32)
        enc = rng.normal(0, 0.6, (r, 1, 7)).astype(np.float32)
        dec = box_encoding.classaware_all_class_box_decoding(
            labels, xyz, enc, lm)
        boxes = np.concatenate([
            rng.uniform(-40, 40, (r, 1, 3)), rng.uniform(0.3, 6, (r, 1, 3)),
            rng.uniform(-np.pi, np.pi, (r, 1, 1))], axis=2).astype(np.float32)
        enc2 = box_encoding.classaware_all_class_box_encoding(
            labels, xyz, boxes, lm)
        out["codec_%s_labels" % name] = labels
        out["codec_%s_xyz" % name] = xyz
        out["codec_%s_encoded" % name] = enc
        out["codec_%s_decoded" % name] = dec
        out["codec_%s_boxes" % name] = boxes
        out["codec_%s_boxes_encoded" % name] = enc2
    np.savez_compressed(os.path.join(HERE, "detect_codec.npz"), **out)

    # ---- NMS (nms.py:241-300), the call shapes of run.py:291-323
    out = {}
    fns = {"plain": nms.nms_boxes_3d,
           "uncertainty": nms.nms_boxes_3d_uncertainty,
           "merge_only": nms.nms_boxes_3d_merge_only,
           "score_only": nms.nms_boxes_3d_score_only}
    for case, (seed, thres, top_k, kw) in enumerate(CASES):
        labels, boxes, scores = DO.synthetic_detections(seed, **kw)
        out["nms%d_labels" % case] = labels
        out["nms%d_boxes" % case] = boxes
        out["nms%d_scores" % case] = scores
        out["nms%d_params" % case] = np.array([thres, top_k], np.float64)
        for mode, fn in fns.items():
            cl, bx, sc, at = fn(
                labels.copy(), boxes.copy(), scores.copy(),
                overlapped_fn=nms.overlapped_boxes_3d_fast_poly,
                overlapped_thres=thres, appr_factor=100.0, top_k=top_k,
                attributes=np.arange(len(labels)))
            out["nms%d_%s_labels" % (case, mode)] = cl
            out["nms%d_%s_boxes" % (case, mode)] = bx
            out["nms%d_%s_scores" % (case, mode)] = sc
            out["nms%d_%s_attrs" % (case, mode)] = at
            print(case, mode, len(labels), "->", len(cl))
    # corners + pairwise overlap of the reference functions themselves
    labels, boxes, scores = DO.synthetic_detections(3, n_objects=4)
    corners = nms.boxes_3d_to_corners(boxes)
    out["geom_boxes"] = boxes
    out["geom_corners"] = corners
    out["geom_overlap_row0"] = nms.overlapped_boxes_3d_fast_poly(
        corners[0], corners[1:])
    np.savez_compressed(os.path.join(HERE, "detect_nms.npz"), **out)
    make_output_fixture(nms)


def make_output_fixture(nms):
    """run.py:360-412 (NMS output -> KITTI label tuples) composed from the
    reference's own functions in run.py's order: nms.boxes_3d_to_corners,
    KittiDataset.cam_points_to_image / sel_xyz_in_box3d / box3d_to_normals;
    run.py itself needs TensorFlow and cannot be imported, so the few lines
    between those calls (clipping, truncation test, `occlusion`, run.py:88-99)
    are restated here."""
    from oracle import ingest_oracle as IO
    for name in ("open3d",):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    try:
        from dataset import kitti_dataset as kd
    finally:
        sys.path.remove(REF)
    ds = object.__new__(kd.KittiDataset)
    calib = IO.get_calib(IO.CALIB_LINES)
    labels, boxes, scores = DO.synthetic_detections(
        5, n_objects=14, half_width=14.0, depth=(4.0, 45.0))
    rng = np.random.default_rng(5)
    cand_xyz = (boxes[:, :3] + rng.normal(0, 0.6, (len(boxes), 3))
                ).astype(np.float32)
    cl, bx, sc, at = nms.nms_boxes_3d_uncertainty(
        labels.copy(), boxes.copy(), scores.copy(),
        overlapped_fn=nms.overlapped_boxes_3d_fast_poly,
        overlapped_thres=0.01, appr_factor=100.0, top_k=-1,
        attributes=np.arange(len(labels)))
    names = ['Background', 'Car', 'Car', 'DontCare']
    corners_all = nms.boxes_3d_to_corners(bx)
    rows, kept = [], []
    for i in range(len(bx)):
        img = ds.cam_points_to_image(
            kd.Points(xyz=corners_all[i], attr=None), calib)
        xy = img.xyz[:, :2]
        xmin, ymin = np.amin(xy, axis=0)
        xmax, ymax = np.amax(xy, axis=0)
        cxmin, cymin = max(xmin, 0.0), max(ymin, 0.0)
        cxmax, cymax = min(xmax, 1242.0), min(ymax, 375.0)
        trunc = 1.0 - (cymax - cymin) * (cxmax - cxmin) / (
            (ymax - ymin) * (xmax - xmin))
        if trunc > 0.4:
            continue
        x3d, y3d, z3d, l, h, w, yaw = bx[i]
        tmp = {"x3d": x3d, "y3d": y3d, "z3d": z3d, "yaw": yaw, "height": h,
               "width": w, "length": l}
        inside = ds.sel_xyz_in_box3d(tmp, cand_xyz)
        pts = cand_xyz[inside]
        occ = 0
        if pts.shape[0]:
            normals, lower, upper = ds.box3d_to_normals(tmp)
            proj = np.matmul(pts, np.transpose(normals))
            occ = 1.0
            for k in range(3):
                occ *= (np.max(proj[:, k]) - np.min(proj[:, k])) / (
                    upper[k] - lower[k])
        rows.append([cxmin, cymin, cxmax, cymax, h, w, l, x3d, y3d, z3d, yaw,
                     (1 + occ) * sc[i], sc[i]])
        kept.append(i)
    print("kitti output:", len(bx), "boxes ->", len(rows), "lines")
    np.savez_compressed(
        os.path.join(HERE, "detect_output.npz"),
        labels=cl, boxes=bx, scores=sc, cand_xyz=cand_xyz,
        kept=np.array(kept), rows=np.array(rows, np.float64),
        names=np.array([names[int(cl[i])] for i in kept]))


if __name__ == "__main__":
    main()
