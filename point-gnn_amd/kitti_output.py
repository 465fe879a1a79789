"""Detections -> KITTI label lines, the tail of the reference's per-frame loop
(run.py:360-433): project each kept 3D box into the image, clip, drop boxes
truncated by more than 40 %, optionally rescore by the occlusion factor
(run.py:88-99, :397-406) and write `<frame>.txt`.

At most a few hundred boxes per frame: the projection and formatting are host
NumPy like the reference; the only per-point step -- which candidate vertices
lie inside a box (`sel_xyz_in_box3d`) -- runs on the device
(`pgnn_assign_box_labels`).
"""
import os

import numpy as np

from . import kitti_dataset

# run.py:370-382
CLASS_NAMES = {
    'yaw': ['Background', 'Car', 'Car', 'Pedestrian', 'Pedestrian',
            'Cyclist', 'Cyclist', 'DontCare'],
    'alpha': ['Background', 'Car', 'Car', 'Pedestrian', 'Pedestrian',
              'Cyclist', 'Cyclist', 'DontCare'],
    'Car': ['Background', 'Car', 'Car', 'DontCare'],
    'Pedestrian_and_Cyclist': ['Background', 'Pedestrian', 'Pedestrian',
                               'Cyclist', 'Cyclist', 'DontCare'],
}
# run.py:244-250
LABEL_MAPS = {
    'yaw': {'Background': 0, 'Car': 1, 'Pedestrian': 3, 'Cyclist': 5,
            'DontCare': 7},
    'Car': {'Background': 0, 'Car': 1, 'DontCare': 3},
    'Pedestrian_and_Cyclist': {'Background': 0, 'Pedestrian': 1,
                               'Cyclist': 3, 'DontCare': 5},
}


def _host(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def box_corners_in_image(box, calib):
    """nms.boxes_3d_to_corners (nms.py:9-27) of one (x,y,z,l,h,w,yaw) box +
    cam_points_to_image (kitti_dataset.py:1036-1052) -> [8,2] float64."""
    label = {'x3d': box[0], 'y3d': box[1], 'z3d': box[2], 'length': box[3],
             'height': box[4], 'width': box[5], 'yaw': box[6]}
    corners = kitti_dataset.box3d_to_cam_points(label).xyz
    homo = np.hstack([corners, np.ones((8, 1))])
    img = np.matmul(homo, np.transpose(calib['cam_to_image']))
    return (img / img[:, [2]])[:, :2]


def occlusion(label, xyz, _nlu=None):
    """run.py:88-99: product over the three box axes of the fraction of the
    box extent covered by the points inside it.  (_nlu: the label's
    box3d_to_normals when the caller has it already.)"""
    if xyz.shape[0] == 0:
        return 0
    normals, lower, upper = _nlu if _nlu is not None else \
        kitti_dataset.box3d_to_normals(label)
    projected = np.matmul(xyz, np.transpose(normals))
    rate = 1.0
    for k in range(3):
        rate = rate * ((np.max(projected[:, k]) - np.min(projected[:, k]))
                       / (upper[k] - lower[k]))
    return rate


def inside_box_host(label, xyz, _nlu=None, _xyz64=None):
    """kitti_dataset.sel_xyz_in_box3d on the HOST, bit for bit what the
    device kernel computes (csrc/labels.hip assign_labels_kernel: float32
    points widened, p = (x n0 + y n1) + z n2 in float64 without contraction,
    strictly inside): the frame loop with frames in flight has the candidates
    on the host already and must not go back to the device once per box.
    (_nlu / _xyz64: the label's box3d_to_normals / the widened columns when
    the caller has them already.)"""
    normals, lower, upper = _nlu if _nlu is not None else \
        kitti_dataset.box3d_to_normals(label)
    x, y, z = _xyz64 if _xyz64 is not None else \
        tuple(xyz[:, i].astype(np.float64) for i in range(3))
    inside = np.ones(xyz.shape[0], bool)
    for k in range(3):
        p = (x * normals[k, 0] + y * normals[k, 1]) + z * normals[k, 2]
        inside &= (p > lower[k]) & (p < upper[k])
    return inside


def detections_to_kitti_labels(class_labels, detection_boxes_3d, box_probs,
                               calib, label_method, candidate_xyz=None,
                               use_box_score=True,
                               image_size=(1242.0, 375.0), host_only=False):
    """run.py:360-412.  class_labels [M], detection_boxes_3d [M,7], box_probs
    [M] = the NMS outputs; candidate_xyz [n_candidates,3] = the vertices of
    ALL candidates that entered the NMS (`last_layer_points_xyz[box_indices]`),
    needed for the occlusion rescoring when use_box_score.  Returns the list
    of 16-field tuples the reference writes.  host_only: every argument is a
    NumPy array and nothing touches the device (inside_box_host)."""
    labels = _host(class_labels)
    boxes = _host(detection_boxes_3d)
    probs = _host(box_probs)
    names = CLASS_NAMES[label_method]
    width, height = float(image_size[0]), float(image_size[1])
    cand_dev = None
    xyz64 = None
    if use_box_score and host_only:
        if candidate_xyz is None:
            raise ValueError("use_box_score needs candidate_xyz")
        cand_host = np.asarray(candidate_xyz)
        xyz64 = tuple(cand_host[:, i].astype(np.float64) for i in range(3))
    elif use_box_score:
        if candidate_xyz is None:
            raise ValueError("use_box_score needs candidate_xyz")
        import torch
        cand_dev = candidate_xyz if isinstance(candidate_xyz, torch.Tensor) \
            else torch.as_tensor(np.asarray(candidate_xyz))
        if not cand_dev.is_cuda:
            cand_dev = cand_dev.cuda()
        cand_host = _host(cand_dev)
    out = []
    for i in range(len(boxes)):
        xy = box_corners_in_image(boxes[i], calib)
        xmin, ymin = np.amin(xy, axis=0)
        xmax, ymax = np.amax(xy, axis=0)
        clip_xmin, clip_ymin = max(xmin, 0.0), max(ymin, 0.0)
        clip_xmax, clip_ymax = min(xmax, width), min(ymax, height)
        truncation = 1.0 - (clip_ymax - clip_ymin) * (clip_xmax - clip_xmin) \
            / ((ymax - ymin) * (xmax - xmin))
        if truncation > 0.4:
            continue
        x3d, y3d, z3d, l, h, w, yaw = boxes[i]
        assert l > 0, str(i)
        score = probs[i]
        if use_box_score:
            tmp = {"x3d": x3d, "y3d": y3d, "z3d": z3d, "yaw": yaw,
                   "height": h, "width": w, "length": l}
            nlu = kitti_dataset.box3d_to_normals(tmp)
            inside = inside_box_host(tmp, cand_host, nlu, xyz64) \
                if host_only else \
                _host(kitti_dataset.sel_xyz_in_box3d(tmp, cand_dev))
            score = (1 + occlusion(tmp, cand_host[inside], nlu)) * score
        out.append((names[int(labels[i])], -1, -1, 0, clip_xmin, clip_ymin,
                    clip_xmax, clip_ymax, h, w, l, x3d, y3d, z3d, yaw, score))
    return out


def write_kitti_txt(filename, pred_labels):
    """run.py:423-433: one line per detection, fields separated (and
    followed) by a blank, an empty line at the end."""
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(filename, "w") as f:
        for pred in pred_labels:
            for field in pred:
                f.write(str(field) + ' ')
            f.write('\n')
        f.write('\n')
