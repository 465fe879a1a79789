"""Training augmentations of the reference (models/preprocess.py) that the
shipped train configs use -- `random_rotation_all`, `random_flip_all`,
`random_box_shift` -- with the per-point work on the GPU.

Same registry (`get_data_aug(aug_configs)`, `aug_method_map`), same call shape
`fn(cam_rgb_points, labels, **method_kwargs) -> (cam_rgb_points, labels)` and
the same consumption of NumPy's global random stream (one draw per decision, in
the reference's order), so a seeded run makes the same random choices.

Points live in a float64 CUDA tensor from the first augmentation on, like the
reference's array after its first `xyz.dot(R.T)` (preprocess.py:55);
`finish(points)` is train.py:124's cast back to float32.  Label dictionaries
are edited on the host exactly where the reference edits them.

`random_box_shift` tests a candidate position against the already placed boxes
with the reference's raster overlap (`nms.overlapped_boxes_3d`, nms.py:29-62:
cv2.fillPoly on the integer corner grid): `pgnn_overlapped_boxes_3d_raster`
produces cv2's pixel counts in closed form.  cv2 itself is absent from the
image; the fill rule is restated from OpenCV 4.2's drawing.cpp
(oracle/raster_oracle.py) and the fixtures come from the reference's own
preprocess.py / nms.py running on that restatement.  The other
registry entries raise NotImplementedError.
"""
import random
from copy import deepcopy

import numpy as np

from . import _lib
from . import kitti_dataset
from .kitti_dataset import Points


def _f64_points(cam_rgb_points):
    import torch
    xyz = cam_rgb_points.xyz
    if not isinstance(xyz, torch.Tensor):
        xyz = torch.as_tensor(np.asarray(xyz))
    if not xyz.is_cuda:
        xyz = xyz.cuda()
    if xyz.dtype != torch.float64:
        xyz = xyz.to(torch.float64)     # a copy: inputs stay untouched
    return xyz.contiguous()


def finish(cam_rgb_points):
    """train.py:124: the float32 cloud the graph builder receives."""
    import torch
    return Points(xyz=cam_rgb_points.xyz.to(torch.float32),
                  attr=cam_rgb_points.attr)


def _affine(xyz, rot=None, shift=None, select=None):
    import torch
    lib = _lib.load()
    r = np.ascontiguousarray(rot, np.float64) if rot is not None else None
    s = np.ascontiguousarray(shift, np.float64) if shift is not None else None
    with torch.cuda.device(xyz.device):
        _lib.check(lib.pgnn_points_affine_f64(
            _lib.ptr(xyz), int(xyz.shape[0]),
            r.ctypes.data if r is not None else None,
            s.ctypes.data if s is not None else None,
            _lib.ptr(select) if select is not None else None,
            _lib.stream_ptr()), "pgnn_points_affine_f64")


def _box_record(label, expend_factor):
    rec = np.zeros(24, np.float64)
    normals, lower, upper = kitti_dataset.box3d_to_normals(label, expend_factor)
    rec[0:9], rec[9:12], rec[12:15] = normals.reshape(-1), lower, upper
    rec[15] = 2.0
    return rec


def _in_box(xyz, label, expend_factor, exclude=None, want_mask=False):
    """(mask int32 [n] or None, count) of the points strictly inside the box
    (kitti_dataset.sel_xyz_in_box3d on the float64 cloud)."""
    import torch
    lib = _lib.load()
    rec = _box_record(label, expend_factor)
    n = int(xyz.shape[0])
    mask = torch.empty((n,), dtype=torch.int32, device=xyz.device) \
        if want_mask else None
    count = torch.zeros((1,), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(lib.pgnn_points_in_box_f64(
            _lib.ptr(xyz), n, rec.ctypes.data,
            _lib.ptr(exclude) if exclude is not None else None,
            _lib.ptr(mask) if mask is not None else None, _lib.ptr(count),
            _lib.stream_ptr()), "pgnn_points_in_box_f64")
    return mask, count


def _yaw_matrix(delta_yaw):
    c, s = np.cos(delta_yaw), np.sin(delta_yaw)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def random_rotation_all(cam_rgb_points, labels, method_name='normal',
                        yaw_std=0.3, expend_factor=(1.0, 1.1, 1.1)):
    """preprocess.py:44-66: one yaw rotation of the whole scene."""
    xyz = _f64_points(cam_rgb_points)
    if method_name == 'normal':
        delta_yaw = np.random.normal(scale=yaw_std)
    elif method_name == 'uniform':
        delta_yaw = np.random.uniform(low=-yaw_std, high=yaw_std)
    else:
        raise ValueError(method_name)
    rot = _yaw_matrix(delta_yaw)
    _affine(xyz, rot=rot)
    for label in labels:
        if label['name'] != 'DontCare':
            centre = np.array([[label['x3d'], label['y3d'], label['z3d']]])
            centre = centre.dot(np.transpose(rot))
            label['x3d'], label['y3d'], label['z3d'] = centre[0]
            label['yaw'] = label['yaw'] + delta_yaw
    return Points(xyz=xyz, attr=cam_rgb_points.attr), labels


def random_flip_all(cam_rgb_points, labels, flip_prob=0.5):
    """preprocess.py:68-77: mirror x with probability flip_prob."""
    xyz = _f64_points(cam_rgb_points)
    p = np.random.uniform()
    if p < flip_prob:
        _affine(xyz, rot=np.diag([-1.0, 1.0, 1.0]))
        for label in labels:
            if label['name'] != 'DontCare':
                label['x3d'] = -label['x3d']
                label['yaw'] = np.pi - label['yaw']
    return Points(xyz=xyz, attr=cam_rgb_points.attr), labels


_AUGMENT_LIST = ['Car', 'Pedestrian', 'Cyclist', 'Van', 'Truck', 'Misc',
                 'Tram', 'Person_sitting']


def _box7(label):
    return [label['x3d'], label['y3d'], label['z3d'], label['length'],
            label['height'], label['width'], label['yaw']]


def random_box_shift(cam_rgb_points, labels, max_overlap_num_allowed=0.1,
                     max_overlap_rate=None, max_trails=100, appr_factor=100,
                     method_name='normal', xyz_std=(1, 0, 1),
                     expend_factor=(1.0, 1.1, 1.1),
                     augment_list=_AUGMENT_LIST, shuffle=False):
    """preprocess.py:239-326: move each object (its box and the points inside
    it) by a random offset, retrying up to max_trails times until the new
    position swallows fewer than max_overlap_num_allowed foreign points and
    overlaps no already placed box by max_overlap_rate or more."""
    from . import nms
    xyz = _f64_points(cam_rgb_points)
    movable = [l for l in labels if l['name'] != 'DontCare']
    if shuffle:
        random.shuffle(movable)
    placed = []
    for label in movable:
        if label['name'] not in augment_list:
            placed.append(label)
            continue
        own, _ = _in_box(xyz, label, expend_factor, want_mask=True)
        accepted = None
        for _ in range(max_trails):
            if method_name == 'normal':
                delta = np.random.normal(scale=xyz_std)
            elif method_name == 'uniform':
                delta = np.random.uniform(low=-np.asarray(xyz_std),
                                          high=np.asarray(xyz_std))
            else:
                raise ValueError(method_name)
            moved = deepcopy(label)
            moved['x3d'] = moved['x3d'] + delta[0]
            moved['y3d'] = moved['y3d'] + delta[1]
            moved['z3d'] = moved['z3d'] + delta[2]
            _, extra = _in_box(xyz, moved, expend_factor, exclude=own)
            ok = int(extra.item()) < max_overlap_num_allowed
            if max_overlap_rate is not None and placed:
                # preprocess.py:281-301: integer corner grids (truncation of
                # appr_factor * corners), cv2 raster overlap against every
                # label placed so far -- shifted or not
                new_corners = np.int32(appr_factor * nms.boxes_3d_to_corners(
                    np.array([_box7(moved)])))
                placed_corners = np.int32(
                    appr_factor * nms.boxes_3d_to_corners(
                        np.array([_box7(l) for l in placed])))
                overlap = nms.overlapped_boxes_3d(new_corners[0],
                                                  placed_corners)
                ok = ok and bool(np.all(overlap < max_overlap_rate))
            if ok:
                _affine(xyz, shift=delta, select=own)
                accepted = moved
                break
        placed.append(accepted if accepted is not None else label)
    assert len(placed) == len(movable)
    placed.extend([l for l in labels if l['name'] == 'DontCare'])
    assert len(placed) == len(labels)
    return Points(xyz=xyz, attr=cam_rgb_points.attr), placed


def empty(cam_rgb_points, labels):
    return cam_rgb_points, labels


def _not_implemented(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            "%s: no shipped train config uses it; only random_rotation_all, "
            "random_flip_all and random_box_shift have a device path" % name)
    return fn


# preprocess.py:446-460
aug_method_map = {
    'random_rotation_all': random_rotation_all,
    'random_flip_all': random_flip_all,
    'random_box_shift': random_box_shift,
}
for _name in ('random_jitter', 'random_box_rotation', 'random_transition',
              'remove_background', 'random_drop', 'random_global_drop',
              'random_voxel_downsample', 'random_scale_all',
              'random_box_global_rotation', 'dilute_background'):
    aug_method_map[_name] = _not_implemented(_name)


def get_data_aug(aug_configs=[]):
    """preprocess.py:461-471."""
    if len(aug_configs) == 0:
        return empty

    def multiple_aug(cam_rgb_points, labels):
        for aug_config in aug_configs:
            method = aug_method_map[aug_config['method_name']]
            cam_rgb_points, labels = method(cam_rgb_points, labels,
                                            **aug_config['method_kwargs'])
        return cam_rgb_points, labels
    return multiple_aug
