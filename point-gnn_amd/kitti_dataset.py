"""KITTI frame ingest of the reference (dataset/kitti_dataset.py) with the
per-point work on the GPU: velodyne scan -> camera frame -> camera-image crop.

`KittiDataset` keeps the reference's method names for this slice
(`get_calib`, `get_velo_points`, `get_cam_points_in_image`,
`get_cam_points_in_image_with_rgb`, `get_filename`); file reads and the
calibration algebra stay on the host (a few 4x4 products per frame), the
per-point transform / projection / crop / ordered compaction is
`pgnn_kitti_cam_points_in_image`.  Results are `Points(xyz, attr)` with CUDA
tensors, ready for `graph_gen` and the model.

Not implemented here (raise or absent): labels, augmentation, voxel
down-sampling (`downsample_by_voxel_size` is null in every shipped config),
PNG decoding -- only the image SIZE is needed for the crop, read from the PNG
header; for 'irgb' features pass a decoded BGR array.
"""
import os
import struct
from collections import namedtuple

import numpy as np

from . import _lib

Points = namedtuple('Points', ['xyz', 'attr'])   # kitti_dataset.py:14


def parse_calib(path_or_lines):
    """kitti_dataset.py:483-522 `get_calib`: the dictionary of calibration
    matrices with the reference's dtypes (P2 / R0_rect / Tr_velo_to_cam are
    float32 as parsed; every derived matrix is float64)."""
    if isinstance(path_or_lines, str):
        with open(path_or_lines, 'r') as f:
            lines = f.readlines()
    else:
        lines = list(path_or_lines)
    calib = {}
    for line in lines:
        fields = line.split(' ')
        name = fields[0].rstrip(':')
        if not name.strip():
            continue
        calib[name] = np.array(fields[1:], dtype=np.float32)
    p2 = calib['P2'].reshape(3, 4)
    r0 = calib['R0_rect'].reshape(3, 3)
    tr = calib['Tr_velo_to_cam'].reshape(3, 4)
    calib['P2'], calib['R0_rect'], calib['Tr_velo_to_cam'] = p2, r0, tr
    bottom = np.array([[0.0, 0.0, 0.0, 1.0]])
    calib['velo_to_rect'] = np.concatenate([tr.astype(np.float64), bottom])
    calib['cam_to_image'] = np.concatenate(
        [p2[:, 0:3].astype(np.float64), np.zeros((3, 1))], axis=1)
    # rectified -> camera-2 frame: rotation R0 and the baseline shift
    # inv(P2[:, :3]) @ P2[:, 3], both evaluated in float32 like the reference
    shift = np.matmul(np.linalg.inv(p2[:, 0:3]), p2[:, [3]])
    rect_to_cam = np.concatenate([r0, shift], axis=1).astype(np.float64)
    calib['rect_to_cam'] = np.concatenate([rect_to_cam, bottom])
    calib['velo_to_cam'] = np.matmul(calib['rect_to_cam'],
                                     calib['velo_to_rect'])
    calib['cam_to_velo'] = np.linalg.inv(calib['velo_to_cam'])
    calib['velo_to_image'] = np.matmul(calib['cam_to_image'],
                                       calib['velo_to_cam'])
    return calib


def png_size(path):
    """(height, width) from a PNG's IHDR chunk (no decoder needed)."""
    with open(path, 'rb') as f:
        head = f.read(24)
    if len(head) < 24 or head[:8] != b'\x89PNG\r\n\x1a\n' or \
            head[12:16] != b'IHDR':
        raise ValueError("%s is not a PNG file" % path)
    width, height = struct.unpack('>II', head[16:24])
    return int(height), int(width)


def _stage_upload(arr, dev):
    """Host float32 array -> device tensor through pinned memory (an
    asynchronous copy-engine transfer instead of the runtime's pageable
    staging path).  The block comes from torch's caching host allocator, which
    hands it out again only after the copy has run."""
    import torch
    host = torch.empty(int(arr.size), dtype=torch.float32, pin_memory=True)
    host.numpy()[:] = arr.reshape(-1)
    return host.to(dev, non_blocking=True)


class PendingPoints(object):
    """cam_points_in_image(..., deferred=True): the crop is enqueued, the
    number of points inside the image is on its way to pinned memory, nothing
    has been waited for.  `event` orders a consumer stream behind the crop;
    result() waits for the count (not for the device) and returns Points --
    call it with the consuming stream current and after
    `stream.wait_event(pending.event)`."""

    def __init__(self, xyz, attr, count_host, event, pad_rgb):
        self.xyz, self.attr, self.count_host = xyz, attr, count_host
        self.event, self.pad_rgb = event, pad_rgb

    def result(self):
        import torch
        self.event.synchronize()
        m = int(self.count_host[0])
        xyz, attr = self.xyz[:m], self.attr[:m]
        if self.pad_rgb:
            zeros = torch.zeros((m, 3), dtype=torch.float32, device=attr.device)
            attr = torch.cat([attr, zeros], dim=1)
        return Points(xyz=xyz, attr=attr)


def cam_points_in_image(velo_data, calib, image_shape, image=None,
                        with_rgb=False, deferred=False):
    """velo_data [n,4] float32 (x,y,z,reflectance; NumPy or CUDA tensor) ->
    Points(xyz [m,3], attr [m,1] or [m,4]) as CUDA tensors:
    kitti_dataset.py:666-689 (`get_cam_points_in_image`) / :691-716 (`..._with_
    rgb`) after the file reads.  image_shape = (height, width); image = decoded
    BGR uint8 [H,W,3] (cv2.imread layout) when with_rgb."""
    import torch
    lib = _lib.load()
    dev = velo_data.device if isinstance(velo_data, torch.Tensor) and \
        velo_data.is_cuda else torch.device("cuda",
                                            torch.cuda.current_device())
    if isinstance(velo_data, np.ndarray) and velo_data.dtype == np.float32 \
            and velo_data.flags.c_contiguous and velo_data.size:
        v = _stage_upload(velo_data, dev)
    else:
        v = torch.as_tensor(velo_data).to(device=dev, dtype=torch.float32)
    v = v.reshape(-1, 4).contiguous()
    n = int(v.shape[0])
    height, width = int(image_shape[0]), int(image_shape[1])
    # kitti_dataset.py:1002-1005: the float32 casts of the transposed blocks
    rt = np.ascontiguousarray(calib['velo_to_cam'][:3, :4].astype(np.float32))
    p = np.ascontiguousarray(calib['cam_to_image'][:3, :3].astype(np.float64))
    attr_dim = 4 if with_rgb else 1
    img_t = None
    if with_rgb:
        if image is None:
            raise ValueError("with_rgb needs the decoded BGR image")
        img_t = torch.as_tensor(image).to(device=dev, dtype=torch.uint8)
        img_t = img_t.contiguous()
        if img_t.dim() != 3 or img_t.shape[2] != 3:
            raise ValueError("image must be [H, W, 3] uint8 (BGR)")
    out_xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    out_attr = torch.empty((n, attr_dim), dtype=torch.float32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = int(lib.pgnn_kitti_ingest_workspace_bytes(n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_kitti_cam_points_in_image(
            _lib.ptr(v), n, rt.ctypes.data, p.ctypes.data, float(width),
            float(height), _lib.ptr(img_t) if img_t is not None else None,
            int(img_t.shape[0]) if img_t is not None else 0,
            int(img_t.shape[1]) if img_t is not None else 0,
            _lib.ptr(ws), ws_bytes, _lib.ptr(out_xyz), _lib.ptr(out_attr),
            attr_dim, n, _lib.ptr(count), _lib.stream_ptr()),
            "pgnn_kitti_cam_points_in_image")
        if deferred:
            count_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            count_host.copy_(count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return PendingPoints(out_xyz, out_attr, count_host, ev, False)
        m = int(count.item())
    return Points(xyz=out_xyz[:m], attr=out_attr[:m])


class KittiDataset(object):
    """The inference-side slice of kitti_dataset.py:187-216: an index of
    frames under `image_dir` / `point_dir` / `calib_dir` (KITTI object layout:
    <name>.png, <name>.bin, <name>.txt)."""

    def __init__(self, image_dir, point_dir, calib_dir, label_dir=None,
                 index_filename=None, is_training=False, is_raw=False,
                 difficulty=-100, num_classes=8):
        if is_raw:
            raise NotImplementedError(
                "the raw-sequence layout is outside the ingest slice")
        if is_training and not label_dir:
            raise ValueError("is_training needs label_dir")
        self._image_dir = image_dir
        self._point_dir = point_dir
        self._calib_dir = calib_dir
        self._label_dir = label_dir
        if index_filename:
            with open(index_filename, 'r') as f:
                self._file_list = [l.rstrip('\n') for l in f if l.strip()]
        else:
            self._file_list = sorted(
                f[:-4] for f in os.listdir(point_dir) if f.endswith('.bin'))
        self.num_classes = num_classes
        self.difficulty = difficulty

    @property
    def num_files(self):
        return len(self._file_list)

    def get_filename(self, frame_idx):
        return self._file_list[frame_idx]

    def get_calib(self, frame_idx):
        return parse_calib(os.path.join(
            self._calib_dir, self._file_list[frame_idx]) + '.txt')

    def get_velo_points(self, frame_idx, xyz_range=None):
        """kitti_dataset.py:587-609 -> host Points (the crop runs later)."""
        data = np.fromfile(os.path.join(
            self._point_dir, self._file_list[frame_idx]) + '.bin',
            dtype=np.float32).reshape(-1, 4)
        if xyz_range is not None:
            m = np.ones(len(data), bool)
            for axis, (lo, hi) in enumerate(xyz_range):
                m &= (data[:, axis] > lo) & (data[:, axis] < hi)
            data = data[m]
        return Points(xyz=data[:, :3], attr=data[:, [3]])

    def _image_shape(self, frame_idx):
        return png_size(os.path.join(
            self._image_dir, self._file_list[frame_idx]) + '.png')

    def get_cam_points_in_image(self, frame_idx, downsample_voxel_size=None,
                                calib=None, xyz_range=None):
        if downsample_voxel_size is not None:
            raise NotImplementedError(
                "downsample_by_voxel_size is null in every shipped config")
        if calib is None:
            calib = self.get_calib(frame_idx)
        pts = self.get_velo_points(frame_idx, xyz_range=xyz_range)
        velo = np.concatenate([pts.xyz, pts.attr], axis=1)
        return cam_points_in_image(velo, calib, self._image_shape(frame_idx))

    def get_cam_points_in_image_with_rgb(self, frame_idx,
                                         downsample_voxel_size=None,
                                         calib=None, xyz_range=None,
                                         image=None, deferred=False):
        """With `image` (decoded BGR array) the attributes are
        [reflectance, r, g, b]; without one (PNG decoding is not part of this
        package) the colour channels are zero -- enough for the 'i' / 'i000' /
        '0' input features of the shipped configs (run.py:226-241).
        deferred (extension): enqueue only and return a PendingPoints."""
        if downsample_voxel_size is not None:
            raise NotImplementedError(
                "downsample_by_voxel_size is null in every shipped config")
        if calib is None:
            calib = self.get_calib(frame_idx)
        if xyz_range is None:
            # the file's [n,4] rows as they are (no split + re-join)
            velo = np.fromfile(os.path.join(
                self._point_dir, self._file_list[frame_idx]) + '.bin',
                dtype=np.float32).reshape(-1, 4)
        else:
            pts = self.get_velo_points(frame_idx, xyz_range=xyz_range)
            velo = np.concatenate([pts.xyz, pts.attr], axis=1)
        shape = image.shape[:2] if image is not None \
            else self._image_shape(frame_idx)
        if image is None:
            import torch
            p = cam_points_in_image(velo, calib, shape, deferred=deferred)
            if deferred:
                p.pad_rgb = True
                return p
            zeros = torch.zeros((p.attr.shape[0], 3), dtype=torch.float32,
                                device=p.attr.device)
            return Points(xyz=p.xyz, attr=torch.cat([p.attr, zeros], dim=1))
        return cam_points_in_image(velo, calib, shape, image=image,
                                   with_rgb=True, deferred=deferred)

    # ---- labels and training targets (kitti_dataset.py:703-751, 1132-1284)
    def get_label(self, frame_idx, no_orientation=False):
        """kitti_dataset.py:703-751: list of label dictionaries of a frame."""
        return read_label_file(os.path.join(
            self._label_dir, self._file_list[frame_idx]) + '.txt',
            self.difficulty)

    def sel_xyz_in_box3d(self, label, xyz, expend_factor=(1.0, 1.0, 1.0)):
        return sel_xyz_in_box3d(label, xyz, expend_factor)

    def box3d_to_cam_points(self, label, expend_factor=(1.0, 1.0, 1.0)):
        return box3d_to_cam_points(label, expend_factor)

    def box3d_to_normals(self, label, expend_factor=(1.0, 1.0, 1.0)):
        return box3d_to_normals(label, expend_factor)

    def assign_classaware_label_to_points(self, labels, xyz, expend_factor):
        """kitti_dataset.py:1132-1182 (label_method 'yaw', 8 classes)."""
        assert self.num_classes == 8
        return assign_label_to_points(labels, xyz, expend_factor, {
            'Background': 0, 'Car': 1, 'Pedestrian': 3, 'Cyclist': 5,
            'DontCare': 7})

    def assign_classaware_car_label_to_points(self, labels, xyz,
                                              expend_factor):
        """kitti_dataset.py:1184-1232 (label_method 'Car', 4 classes)."""
        assert self.num_classes == 4
        return assign_label_to_points(labels, xyz, expend_factor, {
            'Background': 0, 'Car': 1, 'DontCare': 3})

    def assign_classaware_ped_and_cyc_label_to_points(self, labels, xyz,
                                                      expend_factor):
        """kitti_dataset.py:1234-1284 ('Pedestrian_and_Cyclist', 6)."""
        assert self.num_classes == 6
        return assign_label_to_points(labels, xyz, expend_factor, {
            'Background': 0, 'Pedestrian': 1, 'Cyclist': 3, 'DontCare': 5})


_LABEL_FIELDS = (('truncation', float), ('occlusion', int), ('alpha', float),
                 ('xmin', float), ('ymin', float), ('xmax', float),
                 ('ymax', float), ('height', float), ('width', float),
                 ('length', float), ('x3d', float), ('y3d', float),
                 ('z3d', float), ('yaw', float))


def read_label_file(path, difficulty=-100):
    """kitti_dataset.py:703-751: KITTI label txt -> list of dicts, filtered by
    `difficulty` (0/1/2 = easy/moderate/hard limits; < 0 keeps everything)."""
    min_height = (40, 25, 25)
    max_occlusion = (0, 1, 2)
    max_truncation = (0.15, 0.3, 0.5)
    labels = []
    with open(path, 'r') as f:
        for line in f:
            fields = line.strip().split(' ')
            if fields == ['']:
                continue
            label = {'name': fields[0]}
            for (key, conv), text in zip(_LABEL_FIELDS, fields[1:15]):
                label[key] = conv(text)
            if len(fields) > 15:
                label['score'] = float(fields[15])
            if difficulty > -1:
                if label['truncation'] > max_truncation[difficulty] or \
                        label['occlusion'] > max_occlusion[difficulty] or \
                        label['ymax'] - label['ymin'] < min_height[difficulty]:
                    continue
            labels.append(label)
    return labels


def box3d_to_cam_points(label, expend_factor=(1.0, 1.0, 1.0)):
    """kitti_dataset.py:85-116: the 8 corners [8,3] float64 of a label box in
    camera coordinates (y down: the box bottom is at y3d), optionally
    expanded: height by expend_factor[0] (half above, half below), width and
    length by [1] and [2].  Returns Points(xyz, None)."""
    yaw = label['yaw']
    c, s = np.cos(yaw), np.sin(yaw)
    h = label['height']
    dh = h * (expend_factor[0] - 1)
    hw = label['width'] * expend_factor[1] / 2
    hl = label['length'] * expend_factor[2] / 2
    top, bottom = dh / 2, -h - dh / 2
    local = np.array([[hl, top, hw], [hl, top, -hw], [-hl, top, -hw],
                      [-hl, top, hw], [hl, bottom, hw], [hl, bottom, -hw],
                      [-hl, bottom, -hw], [-hl, bottom, hw]])
    rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    xyz = local.dot(rot.T) + np.array([label['x3d'], label['y3d'],
                                       label['z3d']])
    return Points(xyz=xyz, attr=None)


def box3d_to_normals(label, expend_factor=(1.0, 1.0, 1.0)):
    """kitti_dataset.py:118-141: edge directions of the box (rows wx, wy, wz)
    and the [lower, upper] bounds of a point's projections on them."""
    p = box3d_to_cam_points(label, expend_factor).xyz
    normals, lower, upper = [], [], []
    for far in (4, 1, 3):              # corner 0 against corners 4, 1, 3
        w = p[[0], :] - p[[far], :]
        normals.append(w)
        lower.append(np.matmul(w, p[far, :]))
        upper.append(np.matmul(w, p[0, :]))
    return (np.concatenate(normals, axis=0), np.concatenate(lower),
            np.concatenate(upper))


def _label_records(labels, expend_factor, label_map):
    """Host half of assign_*_label_to_points: one 24-double record per label
    (layout in include/pointgnn_hip.h)."""
    default = label_map['DontCare']
    rec = np.zeros((len(labels), 24), np.float64)
    for i, label in enumerate(labels):
        name = label['name']
        obj_cls = label_map.get(name, default)
        if 1 <= obj_cls <= default - 1:
            yaw = label['yaw']
            while yaw < -0.25 * np.pi:
                yaw += np.pi
            while yaw > 0.75 * np.pi:
                yaw -= np.pi
            action = 1.0
            cls = obj_cls if yaw < 0.25 * np.pi else obj_cls + 1
            box = (label['x3d'], label['y3d'], label['z3d'], label['length'],
                   label['height'], label['width'], yaw)
        elif name != 'DontCare':
            action, cls, box = 2.0, obj_cls, (0.0,) * 7
        else:
            continue                      # record stays action 0 (skipped)
        normals, lower, upper = box3d_to_normals(label, expend_factor)
        rec[i, 0:9] = normals.reshape(-1)
        rec[i, 9:12] = lower
        rec[i, 12:15] = upper
        rec[i, 15] = action
        rec[i, 16] = cls
        rec[i, 17:24] = box
    return rec


def _assign(xyz, rec, want_boxes=True):
    import torch
    lib = _lib.load()
    as_numpy = not isinstance(xyz, torch.Tensor)
    dev = xyz.device if not as_numpy and xyz.is_cuda else \
        torch.device("cuda", 0)
    p = torch.as_tensor(xyz)
    # float64 vertices stay float64 (train.py:100-118 passes the float64
    # vertex_coord_list of the augmented cloud; np.matmul then runs in float64
    # on the exact coordinates); everything else is the float32 cloud
    wide = p.dtype == torch.float64
    p = p.to(device=dev, dtype=torch.float64 if wide else torch.float32
             ).contiguous()
    n = int(p.shape[0])
    r = torch.from_numpy(np.ascontiguousarray(rec)).to(dev)
    cls = torch.empty((n,), dtype=torch.int32, device=dev)
    boxes = torch.empty((n, 7), dtype=torch.float64, device=dev) \
        if want_boxes else None
    valid = torch.empty((n,), dtype=torch.float32, device=dev)
    owner = torch.empty((n,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        entry = lib.pgnn_assign_box_labels_f64 if wide else \
            lib.pgnn_assign_box_labels
        _lib.check(entry(
            _lib.ptr(p), n, _lib.ptr(r), int(r.shape[0]), _lib.ptr(cls),
            _lib.ptr(boxes) if boxes is not None else None, _lib.ptr(valid),
            _lib.ptr(owner), _lib.stream_ptr()), "pgnn_assign_box_labels")
    return as_numpy, cls, boxes, valid, owner


def sel_xyz_in_box3d(label, xyz, expend_factor=(1.0, 1.0, 1.0)):
    """kitti_dataset.py:143-162: bool mask of the points strictly inside the
    (expanded) box.  NumPy in -> NumPy out, CUDA tensor in -> CUDA tensor."""
    rec = np.zeros((1, 24), np.float64)
    normals, lower, upper = box3d_to_normals(label, expend_factor)
    rec[0, 0:9], rec[0, 9:12], rec[0, 12:15] = normals.reshape(-1), lower, upper
    rec[0, 15] = 2.0
    as_numpy, _, _, _, owner = _assign(xyz, rec, want_boxes=False)
    mask = owner >= 0
    return mask.cpu().numpy() if as_numpy else mask


def assign_label_to_points(labels, xyz, expend_factor, label_map):
    """The shared body of kitti_dataset.py:1132-1284 -> (cls_labels [K,1],
    boxes_3d [K,1,7] float64, valid_boxes [K,1,1] float32, label_map).
    Class values: an object of class c gets c (|yaw| "horizontal") or c+1
    (vertical, yaw wrapped into (-pi/4, 3pi/4]); names outside the map get the
    DontCare value with valid 0; 'DontCare' boxes are ignored."""
    xyz_n = xyz.shape[0]
    assert xyz_n > 0, "No point No prediction"
    assert xyz.shape[1] == 3
    rec = _label_records(labels, expend_factor, label_map)
    as_numpy, cls, boxes, valid, _ = _assign(xyz, rec)
    cls = cls.reshape(-1, 1)
    boxes = boxes.reshape(-1, 1, 7)
    valid = valid.reshape(-1, 1, 1)
    if as_numpy:
        return (cls.cpu().numpy().astype(np.int64), boxes.cpu().numpy(),
                valid.cpu().numpy(), label_map)
    return cls, boxes, valid, label_map
