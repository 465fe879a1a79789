"""Box codec of the reference (models/box_encoding.py) on the GPU.

Same registry (`get_box_encoding_fn` / `get_box_decoding_fn` /
`get_encoding_len`, box_encoding.py:469-508) and the same call signature
`fn(cls_labels [R,1], points_xyz [R,3], boxes [R,B,7], label_map)`.  Only
`'classaware_all_class_box_encoding'` -- the method of every shipped config --
is implemented; the other registry keys raise NotImplementedError.

NumPy in -> NumPy out, CUDA tensors in -> CUDA tensors out; the arithmetic
runs in libpointgnn_hip.so (`pgnn_box_{decode,encode}_f32`), float32 like the
reference's NumPy code on its float32 network outputs.
"""
import math

import numpy as np

from . import _lib

# box_encoding.py:210-220 -- (l, h, w) medians of the KITTI classes
median_object_size_map = {
    'Cyclist': (1.76, 1.75, 0.6),
    'Van': (4.98, 2.13, 1.88),
    'Tram': (14.66, 3.61, 2.6),
    'Car': (3.88, 1.5, 1.63),
    'Misc': (2.52, 1.65, 1.51),
    'Pedestrian': (0.88, 1.77, 0.65),
    'Truck': (10.81, 3.34, 2.63),
    'Person_sitting': (0.75, 1.26, 0.59),
}


def class_table(label_map, dtype=np.float32):
    """[n,5] rows {l, h, w, yaw_offset, active} indexed by label value
    (box_encoding.py:239-262,268-291: label -> horizontal anchor, label+1 ->
    the same size turned by pi/2).  Later label_map entries overwrite earlier
    ones like the reference's loop does."""
    rows = {}
    for name, label in label_map.items():
        if name in ("Background", "DontCare"):
            continue
        l, h, w = median_object_size_map[name]
        rows[int(label)] = (l, h, w, 0.0)
        rows[int(label) + 1] = (l, h, w, 0.5 * math.pi)
    n = (max(rows) + 1) if rows else 0
    table = np.zeros((n, 5), dtype)
    for label, (l, h, w, yaw) in rows.items():
        table[label] = (l, h, w, yaw, 1.0)
    return table


_TABLES = {}


def _device_table(label_map, dtype, dev):
    """class_table(label_map) on `dev`, uploaded once per (map, dtype, device):
    the frame loop decodes every frame with the same table."""
    import torch
    key = (tuple(sorted(label_map.items())), np.dtype(dtype).str, str(dev))
    t = _TABLES.get(key)
    if t is None:
        t = _TABLES[key] = torch.from_numpy(class_table(label_map, dtype)).to(dev)
        torch.cuda.current_stream(dev).synchronize()   # used on any stream
    return t


def _run(entry, cls_labels, points_xyz, boxes, label_map):
    import torch
    lib = _lib.load()
    as_numpy = not isinstance(boxes, torch.Tensor)
    dev = boxes.device if not as_numpy else torch.device("cuda", 0)
    if dev.type != "cuda":
        raise _lib.PointGnnHipError(
            "box codec runs on the GPU only (no CPU fallback)")

    def to(x, dtype):
        return torch.as_tensor(np.asarray(x) if not isinstance(
            x, torch.Tensor) else x).to(device=dev, dtype=dtype).contiguous()

    # float64 ground-truth boxes (kitti_dataset.py:1199 np.zeros default) are
    # encoded in float64 and rounded once, like train.py:120-130
    wide = entry == "pgnn_box_encode_f32" and (
        (as_numpy and np.asarray(boxes).dtype == np.float64) or
        (not as_numpy and boxes.dtype == torch.float64))
    if wide:
        b = to(boxes, torch.float64)
        rows, per_row = int(b.shape[0]), int(b.shape[1])
        lab = to(cls_labels, torch.int32).reshape(-1)
        # float64 vertices (the training path, train.py:120-122) are used
        # as they are; float32 ones are widened inside the kernel
        xyz64 = (np.asarray(points_xyz).dtype == np.float64) if not isinstance(
            points_xyz, torch.Tensor) else points_xyz.dtype == torch.float64
        xyz = to(points_xyz, torch.float64 if xyz64 else torch.float32)
        table = _device_table(label_map, np.float64, dev)
        out = torch.empty(tuple(b.shape), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check((lib.pgnn_box_encode_f64_xyz64 if xyz64 else
                        lib.pgnn_box_encode_f64)(
                _lib.ptr(lab), _lib.ptr(xyz), _lib.ptr(b), _lib.ptr(table),
                int(table.shape[0]), rows, per_row, _lib.ptr(out),
                _lib.stream_ptr()), "pgnn_box_encode_f64")
        return out.cpu().numpy() if as_numpy else out
    b = to(boxes, torch.float32)
    if b.dim() != 3 or b.shape[2] != 7:
        raise ValueError("boxes must be [R, B, 7]")
    rows, per_row = int(b.shape[0]), int(b.shape[1])
    lab = to(cls_labels, torch.int32).reshape(-1)
    xyz = to(points_xyz, torch.float32)
    if lab.numel() != rows or tuple(xyz.shape) != (rows, 3):
        raise ValueError("cls_labels [R,1] / points_xyz [R,3] do not match boxes")
    table = _device_table(label_map, np.float32, dev)
    out = torch.empty_like(b)
    with torch.cuda.device(dev):
        _lib.check(getattr(lib, entry)(
            _lib.ptr(lab), _lib.ptr(xyz), _lib.ptr(b), _lib.ptr(table),
            int(table.shape[0]), rows, per_row, _lib.ptr(out),
            _lib.stream_ptr()), entry)
    return out.cpu().numpy() if as_numpy else out


def classaware_all_class_box_encoding(cls_labels, points_xyz, boxes_3d,
                                      label_map):
    """box_encoding.py:231-263."""
    return _run("pgnn_box_encode_f32", cls_labels, points_xyz, boxes_3d,
                label_map)


def classaware_all_class_box_decoding(cls_labels, points_xyz, encoded_boxes,
                                      label_map):
    """box_encoding.py:265-299."""
    return _run("pgnn_box_decode_f32", cls_labels, points_xyz, encoded_boxes,
                label_map)


_METHODS = ('direct_encoding', 'center_box_encoding', 'voxelnet_box_encoding',
            'classaware_voxelnet_box_encoding',
            'classaware_all_class_box_encoding',
            'classaware_all_class_box_canonical_encoding')


def _lookup(name, fn):
    if name not in _METHODS:
        raise KeyError(name)
    if name != 'classaware_all_class_box_encoding':
        raise NotImplementedError(
            "%s: only 'classaware_all_class_box_encoding' (the method of all "
            "shipped configs) has a HIP implementation" % name)
    return fn


def get_box_encoding_fn(encoding_method_name):
    return _lookup(encoding_method_name, classaware_all_class_box_encoding)


def get_box_decoding_fn(encoding_method_name):
    return _lookup(encoding_method_name, classaware_all_class_box_decoding)


def get_encoding_len(encoding_method_name):
    if encoding_method_name not in _METHODS:
        raise KeyError(encoding_method_name)
    return 7
