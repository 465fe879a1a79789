"""Rotated-box NMS of the reference (models/nms.py) on the GPU, plus the
decode -> select -> NMS stage of run.py:264-326 as one call (`detect_boxes`).

Same entry names and keyword arguments as the reference:
`nms_boxes_3d`, `nms_boxes_3d_uncertainty`, `nms_boxes_3d_merge_only`,
`nms_boxes_3d_score_only` (nms.py:241-300), each
`(class_labels, detection_boxes_3d, detection_scores, overlapped_thres=0.5,
overlapped_fn=..., appr_factor=10.0, top_k=-1, attributes=None) ->
(class_labels, detection_boxes_3d, detection_scores, attributes)`.

Differences a caller can observe:
  * the NMS entries take `overlapped_fn=overlapped_boxes_3d_fast_poly` only
    (what run.py passes, :295-322); that name is here a callable on
    (x,y,z,l,h,w,yaw) boxes -- the corner geometry is built on the device.
    The cv2 raster variant `overlapped_boxes_3d` (nms.py:29-62) exists as a
    standalone function on integer corner arrays, for its one caller on the
    path: random_box_shift (preprocess.py:281-301).
  * inputs are not modified (the reference edits `bboxes`/`scores` in place on
    its sorted copies only, so neither does it in effect);
  * boxes with EQUAL scores keep their input order (np.argsort(-scores),
    nms.py:93, leaves it unspecified).
NumPy in -> NumPy out; CUDA tensors in -> CUDA tensors out.
"""
import ctypes

import numpy as np

from . import _lib
from . import box_encoding

_MODE = {"plain": 0, "uncertainty": 1, "merge_only": 2, "score_only": 3}


def _device_of(*xs):
    import torch
    for x in xs:
        if isinstance(x, torch.Tensor) and x.device.type == "cuda":
            return x.device
    return torch.device("cuda", 0)


def _to(x, dtype, dev):
    import torch
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    return x.to(device=dev, dtype=dtype).contiguous()


def boxes_3d_to_corners(boxes_3d):
    """nms.py:9-27 on the host, float64 like the reference (its callers pass a
    handful of label boxes): [n,7] (x,y,z,l,h,w,yaw) -> [n,8,3].  The rotation
    goes through the same `corners.dot(R.T)` NumPy product, so the values the
    integer truncation of preprocess.py:291-298 sees are the reference's."""
    out = []
    for x3d, y3d, z3d, l, h, w, yaw in np.asarray(boxes_3d).reshape(-1, 7):
        c, s = np.cos(yaw), np.sin(yaw)
        rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        half = np.array([[l / 2, 0.0, w / 2], [l / 2, 0.0, -w / 2],
                         [-l / 2, 0.0, -w / 2], [-l / 2, 0.0, w / 2],
                         [l / 2, -h, w / 2], [l / 2, -h, -w / 2],
                         [-l / 2, -h, -w / 2], [-l / 2, -h, w / 2]])
        out.append(half.dot(np.transpose(rot)) + np.array([x3d, y3d, z3d]))
    return np.array(out)


def overlapped_boxes_3d_fast_poly(single_box, box_list, appr_factor=0.0):
    """nms.py:64-88 for boxes given as (x,y,z,l,h,w,yaw): overlap of
    `single_box` [7] with every row of `box_list` [n,7] -> [n] float64."""
    import torch
    lib = _lib.load()
    as_numpy = not isinstance(box_list, torch.Tensor)
    dev = _device_of(box_list, single_box)
    one = _to(single_box, torch.float32, dev).reshape(7)
    many = _to(box_list, torch.float32, dev).reshape(-1, 7)
    out = torch.empty((many.shape[0],), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_overlapped_boxes_3d(
            _lib.ptr(one), _lib.ptr(many), int(many.shape[0]),
            float(appr_factor), _lib.ptr(out), _lib.stream_ptr()),
            "pgnn_overlapped_boxes_3d")
    return out.cpu().numpy() if as_numpy else out


def overlapped_boxes_3d(single_box, box_list):
    """nms.py:29-62, the cv2.fillPoly raster overlap: `single_box` [8,3] and
    `box_list` [n,8,3] are INTEGER corner arrays
    (np.int32(appr_factor * boxes_3d_to_corners(.)), as the reference's callers
    pass them) -> [n] float64.  Pixel counts are those of cv2.fillPoly +
    cv2.countNonZero on the buffers the reference allocates
    (pgnn_overlapped_boxes_3d_raster; closed form per image row)."""
    import torch
    lib = _lib.load()
    as_numpy = not isinstance(box_list, torch.Tensor)
    dev = _device_of(box_list, single_box)
    many = _to(box_list, torch.int32, dev)
    n = int(many.shape[0]) if many.dim() == 3 else 0
    if as_numpy and n == 0:
        return np.zeros(0)
    one = _to(single_box, torch.int32, dev).reshape(8, 3)
    many = many.reshape(n, 8, 3)
    out = torch.empty((n,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_overlapped_boxes_3d_raster(
            _lib.ptr(one), _lib.ptr(many), n, _lib.ptr(out),
            _lib.stream_ptr()), "pgnn_overlapped_boxes_3d_raster")
    return out.cpu().numpy() if as_numpy else out


def _nms_launch(mode, class_labels, detection_boxes_3d, detection_scores,
                overlapped_thres, overlapped_fn, appr_factor, top_k,
                attributes):
    """Enqueue pgnn_nms_boxes_3d: -> (as_numpy, dtypes, n-row output tensors,
    the device count of kept rows); nothing is read back."""
    import torch
    if overlapped_fn is not overlapped_boxes_3d_fast_poly:
        raise NotImplementedError(
            "overlapped_fn must be pointgnn_amd.nms.overlapped_boxes_3d_fast_poly")
    lib = _lib.load()
    as_numpy = not isinstance(detection_boxes_3d, torch.Tensor)
    dev = _device_of(detection_boxes_3d, detection_scores, class_labels)
    label_dtype = (np.asarray(class_labels).dtype if as_numpy
                   else class_labels.dtype)
    boxes = _to(detection_boxes_3d, torch.float32, dev).reshape(-1, 7)
    n = int(boxes.shape[0])
    labels = _to(class_labels, torch.int32, dev).reshape(-1)
    scores = _to(detection_scores, torch.float32, dev).reshape(-1)
    if labels.numel() != n or scores.numel() != n:
        raise ValueError("labels / boxes / scores lengths differ")
    attrs = None
    attr_dtype = None
    if attributes is not None:
        attr_dtype = (np.asarray(attributes).dtype if not isinstance(
            attributes, torch.Tensor) else attributes.dtype)
        attrs = _to(attributes, torch.int32, dev).reshape(-1)
    o_lab = torch.empty((n,), dtype=torch.int32, device=dev)
    o_box = torch.empty((n, 7), dtype=torch.float32, device=dev)
    o_sco = torch.empty((n,), dtype=torch.float32, device=dev)
    o_att = torch.empty((n,), dtype=torch.int32, device=dev)
    o_cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = int(lib.pgnn_nms_workspace_bytes(n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_nms_boxes_3d(
            _lib.ptr(labels), _lib.ptr(boxes), _lib.ptr(scores),
            _lib.ptr(attrs) if attrs is not None else ctypes.c_void_p(0), n,
            float(overlapped_thres), _MODE[mode], float(appr_factor),
            int(top_k), _lib.ptr(ws), ws_bytes, _lib.ptr(o_lab),
            _lib.ptr(o_box), _lib.ptr(o_sco), _lib.ptr(o_att), _lib.ptr(o_cnt),
            _lib.stream_ptr()), "pgnn_nms_boxes_3d")
    return (as_numpy, label_dtype, attr_dtype,
            (o_lab, o_box, o_sco, o_att if attributes is not None else None),
            o_cnt)


def _nms(mode, class_labels, detection_boxes_3d, detection_scores,
         overlapped_thres, overlapped_fn, appr_factor, top_k, attributes):
    as_numpy, label_dtype, attr_dtype, out, o_cnt = _nms_launch(
        mode, class_labels, detection_boxes_3d, detection_scores,
        overlapped_thres, overlapped_fn, appr_factor, top_k, attributes)
    kept = int(o_cnt.item())
    res = tuple(None if t is None else t[:kept] for t in out)
    if as_numpy:
        return (res[0].cpu().numpy().astype(label_dtype),
                res[1].cpu().numpy(), res[2].cpu().numpy(),
                None if res[3] is None
                else res[3].cpu().numpy().astype(attr_dtype))
    return res


def nms_boxes_3d(class_labels, detection_boxes_3d, detection_scores,
                 overlapped_thres=0.5,
                 overlapped_fn=overlapped_boxes_3d_fast_poly,
                 appr_factor=10.0, top_k=-1, attributes=None):
    """nms.py:241-254 (integer corners scaled by appr_factor, :113-115)."""
    return _nms("plain", class_labels, detection_boxes_3d, detection_scores,
                overlapped_thres, overlapped_fn, appr_factor, top_k,
                attributes)


def nms_boxes_3d_uncertainty(class_labels, detection_boxes_3d,
                             detection_scores, overlapped_thres=0.5,
                             overlapped_fn=overlapped_boxes_3d_fast_poly,
                             appr_factor=10.0, top_k=-1, attributes=None):
    """nms.py:256-270: median merge + score accumulation."""
    return _nms("uncertainty", class_labels, detection_boxes_3d,
                detection_scores, overlapped_thres, overlapped_fn, appr_factor,
                top_k, attributes)


def nms_boxes_3d_merge_only(class_labels, detection_boxes_3d,
                            detection_scores, overlapped_thres=0.5,
                            overlapped_fn=overlapped_boxes_3d_fast_poly,
                            appr_factor=10.0, top_k=-1, attributes=None):
    """nms.py:272-285."""
    return _nms("merge_only", class_labels, detection_boxes_3d,
                detection_scores, overlapped_thres, overlapped_fn, appr_factor,
                top_k, attributes)


def nms_boxes_3d_score_only(class_labels, detection_boxes_3d,
                            detection_scores, overlapped_thres=0.5,
                            overlapped_fn=overlapped_boxes_3d_fast_poly,
                            appr_factor=10.0, top_k=-1, attributes=None):
    """nms.py:287-300."""
    return _nms("score_only", class_labels, detection_boxes_3d,
                detection_scores, overlapped_thres, overlapped_fn, appr_factor,
                top_k, attributes)


def select_candidates(probs):
    """run.py:264-290: (flat indices k*nc + c, merged class labels) of the
    entries with 0 < c < nc-1 and prob > 1/nc, as int32 CUDA tensors."""
    import torch
    lib = _lib.load()
    dev = _device_of(probs)
    p = _to(probs, torch.float32, dev)
    k, nc = int(p.shape[0]), int(p.shape[1])
    cap = k * nc
    idx = torch.empty((cap,), dtype=torch.int32, device=dev)
    lab = torch.empty((cap,), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_detection_candidates(
            _lib.ptr(p), k, nc, _lib.ptr(idx), _lib.ptr(lab), cap,
            _lib.ptr(cnt), _lib.stream_ptr()), "pgnn_detection_candidates")
        n = int(cnt.item())
    return idx[:n], lab[:n]


def select_candidates_dyn(probs, k_dev):
    """select_candidates for a capacity-form frame: probs [cap, nc] of which
    the first `k_dev` (device int32 [1]) rows exist.  Nothing is read back:
    -> (flat indices [cap*nc], labels [cap*nc], count [1]) on the device."""
    import torch
    lib = _lib.load()
    dev = _device_of(probs)
    p = _to(probs, torch.float32, dev)
    cap_rows, nc = int(p.shape[0]), int(p.shape[1])
    cap = cap_rows * nc
    idx = torch.empty((cap,), dtype=torch.int32, device=dev)
    lab = torch.empty((cap,), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pgnn_detection_candidates_dyn(
            _lib.ptr(p), cap_rows, _lib.ptr(k_dev), nc, _lib.ptr(idx),
            _lib.ptr(lab), cap, _lib.ptr(cnt), _lib.stream_ptr()),
            "pgnn_detection_candidates_dyn")
    return idx, lab, cnt


def detect_candidates_deferred(probs, box_encodings, last_layer_points_xyz,
                               idx, labels, label_map, overlapped_thres,
                               box_encoding_method=
                               'classaware_all_class_box_encoding',
                               use_box_merge=True, use_box_score=True,
                               appr_factor=100.0, top_k=-1):
    """The tail of detect_boxes for candidates that are already selected
    (idx, labels: the first n entries select_candidates[_dyn] wrote, n known
    to the host), with nothing read back: decode the candidates' boxes -- the
    codec is per box, so decoding the selection equals selecting from the
    decoded whole --, run the NMS variant, and return ((class_labels,
    boxes_3d, scores, nms_indices) with n rows each, kept count [1] on the
    device, candidate_xyz [n,3])."""
    import torch
    dev = _device_of(probs, box_encodings)
    p = _to(probs, torch.float32, dev)
    nc = int(p.shape[1])
    sel = idx.long()
    vert = sel // nc
    cand_xyz = _to(last_layer_points_xyz, torch.float32, dev)[vert]
    enc = _to(box_encodings, torch.float32, dev).reshape(-1, 1, 7)[sel]
    decode = box_encoding.get_box_decoding_fn(box_encoding_method)
    decoded = decode((sel % nc).to(torch.int32).reshape(-1, 1), cand_xyz, enc,
                     label_map)
    cand_scores = p.reshape(-1)[sel]
    mode = {(True, True): "uncertainty", (True, False): "merge_only",
            (False, True): "score_only", (False, False): "plain"}[
        (bool(use_box_merge), bool(use_box_score))]
    attrs = torch.arange(idx.numel(), dtype=torch.int32, device=dev)
    _, _, _, out, cnt = _nms_launch(
        mode, labels, decoded[:, 0], cand_scores, overlapped_thres,
        overlapped_boxes_3d_fast_poly, appr_factor, top_k, attrs)
    return out, cnt, cand_xyz


def detect_boxes(probs, box_encodings, last_layer_points_xyz, label_map,
                 overlapped_thres, box_encoding_method=
                 'classaware_all_class_box_encoding', use_box_merge=True,
                 use_box_score=True, appr_factor=100.0, top_k=-1):
    """run.py:264-326 on the device: decode every (vertex, class) box, keep the
    foreground candidates above 1/nc, merge the orientation classes and run
    the NMS variant selected by USE_BOX_MERGE / USE_BOX_SCORE (:291-323).

    probs [K,nc], box_encodings [K,nc,7], last_layer_points_xyz [K,3]
    -> (class_labels, detection_boxes_3d, detection_scores, nms_indices) as
    CUDA tensors; nms_indices index the candidate list like the reference's
    `attributes=np.arange(len(box_indices))`."""
    import torch
    dev = _device_of(probs, box_encodings)
    p = _to(probs, torch.float32, dev)
    k, nc = int(p.shape[0]), int(p.shape[1])
    enc = _to(box_encodings, torch.float32, dev).reshape(k * nc, 1, 7)
    xyz = _to(last_layer_points_xyz, torch.float32, dev)
    decode = box_encoding.get_box_decoding_fn(box_encoding_method)
    box_labels = torch.arange(nc, dtype=torch.int32, device=dev).repeat(k)
    centers = xyz.repeat_interleave(nc, dim=0)
    decoded = decode(box_labels.reshape(-1, 1), centers, enc, label_map)
    idx, labels = select_candidates(p)
    if idx.numel() == 0:
        e = torch.empty
        return (e((0,), dtype=torch.int32, device=dev),
                e((0, 7), dtype=torch.float32, device=dev),
                e((0,), dtype=torch.float32, device=dev),
                e((0,), dtype=torch.int32, device=dev))
    sel = idx.long()
    cand_boxes = decoded[sel, 0]
    cand_scores = p.reshape(-1)[sel]
    mode = {(True, True): "uncertainty", (True, False): "merge_only",
            (False, True): "score_only", (False, False): "plain"}[
        (bool(use_box_merge), bool(use_box_score))]
    attrs = torch.arange(idx.numel(), dtype=torch.int32, device=dev)
    return _nms(mode, labels, cand_boxes, cand_scores, overlapped_thres,
                overlapped_boxes_3d_fast_poly, appr_factor, top_k, attrs)
