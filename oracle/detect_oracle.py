"""CPU restatement of the reference's detection post-processing -- TEST
INFRASTRUCTURE ONLY (imported by tests/, never by the product package).

Follows, function by function:
  * box codec `classaware_all_class_box_{encoding,decoding}`
    (models/box_encoding.py:231-299, size table :210-220);
  * candidate selection of run.py:264-290;
  * `boxes_3d_to_corners` (models/nms.py:9-27),
    `overlapped_boxes_3d_fast_poly` (:64-88), `bboxes_sort` (:90-106) and the
    four scan loops `bboxes_nms` (:108-131), `bboxes_nms_uncertainty`
    (:133-170), `bboxes_nms_merge_only` (:172-202), `bboxes_nms_score_only`
    (:204-239) behind `nms_boxes_3d*` (:241-300).

Pinning: the codec and the four scan loops are pinned against the reference's
own code run in the build container (tests/golden/make_golden_detect.py imports
models/box_encoding.py unmodified and models/nms.py under an empty `cv2` stub
and a `shapely.geometry.Polygon` stub backed by `ConvexPolygon` below).
shapely itself is absent from the image, so the polygon intersection AREA is
"parity unpinned" against GEOS; it is checked against closed-form cases
instead (tests/test_detect_cpu.py).  The intersection here is computed by a
different algorithm (vertex collection + angular sort) than the HIP kernel
(Sutherland-Hodgman clipping) so that the two check each other.
"""
import numpy as np

# models/box_encoding.py:210-220 (l, h, w)
MEDIAN_OBJECT_SIZE = {
    'Cyclist': (1.76, 1.75, 0.6),
    'Van': (4.98, 2.13, 1.88),
    'Tram': (14.66, 3.61, 2.6),
    'Car': (3.88, 1.5, 1.63),
    'Misc': (2.52, 1.65, 1.51),
    'Pedestrian': (0.88, 1.77, 0.65),
    'Truck': (10.81, 3.34, 2.63),
    'Person_sitting': (0.75, 1.26, 0.59),
}
_SKIP = ("Background", "DontCare")
_QUARTER_PI = np.pi * 0.25
_HALF_PI = np.pi * 0.5


def box_decoding(cls_labels, points_xyz, encoded_boxes, label_map):
    """box_encoding.py:265-299.  cls_labels [R,1], points_xyz [R,3],
    encoded_boxes [R,B,7] -> decoded [R,B,7] (only box column 0 is class
    scaled, :274-291; the xyz offset goes to all columns, :293-298)."""
    enc = np.asarray(encoded_boxes)
    out = enc.copy()
    lab = np.asarray(cls_labels)[:, 0]
    for name, base in label_map.items():
        if name in _SKIP:
            continue
        l, h, w = MEDIAN_OBJECT_SIZE[name]
        for label, yaw0 in ((base, None), (base + 1, _HALF_PI)):
            m = lab == label
            sub = enc[m, 0, :]
            dec = np.empty_like(sub)
            dec[:, 0] = sub[:, 0] * l
            dec[:, 1] = sub[:, 1] * h
            dec[:, 2] = sub[:, 2] * w
            dec[:, 3] = np.exp(sub[:, 3]) * l
            dec[:, 4] = np.exp(sub[:, 4]) * h
            dec[:, 5] = np.exp(sub[:, 5]) * w
            dec[:, 6] = sub[:, 6] * _QUARTER_PI
            if yaw0 is not None:
                dec[:, 6] = dec[:, 6] + yaw0
            out[m, 0, :] = dec
    out[:, :, 0:3] = out[:, :, 0:3] + np.asarray(points_xyz)[:, None, :]
    return out


def box_encoding(cls_labels, points_xyz, boxes_3d, label_map):
    """box_encoding.py:231-263 (label values of the map must be distinct, as in
    every shipped label_map; the reference's in-place divide would compound
    otherwise)."""
    boxes = np.asarray(boxes_3d)
    out = boxes.copy()
    out[:, :, 0:3] = boxes[:, :, 0:3] - np.asarray(points_xyz)[:, None, :]
    lab = np.asarray(cls_labels)[:, 0]
    for name, base in label_map.items():
        if name in _SKIP:
            continue
        l, h, w = MEDIAN_OBJECT_SIZE[name]
        for label, yaw0 in ((base, None), (base + 1, _HALF_PI)):
            m = lab == label
            src = boxes[m, 0, :]
            enc = np.empty_like(src)
            enc[:, 0] = out[m, 0, 0] / l
            enc[:, 1] = out[m, 0, 1] / h
            enc[:, 2] = out[m, 0, 2] / w
            enc[:, 3] = np.log(src[:, 3] / l)
            enc[:, 4] = np.log(src[:, 4] / h)
            enc[:, 5] = np.log(src[:, 5] / w)
            yaw = src[:, 6] if yaw0 is None else src[:, 6] - yaw0
            enc[:, 6] = yaw / _QUARTER_PI
            out[m, 0, :] = enc
    return out


def select_candidates(probs):
    """run.py:264-290 -> (flat indices, merged labels) for probs [K,nc]."""
    probs = np.asarray(probs)
    k, nc = probs.shape
    labels = np.tile(np.arange(nc), k)
    flat = probs.reshape(-1)
    mask = (labels > 0) & (labels < nc - 1) & (flat > 1. / nc)
    idx = np.nonzero(mask)[0]
    lab = labels[idx].copy()
    for even in (2, 4, 6):
        lab[lab == even] = even - 1
    return idx, lab


# ---------------------------------------------------------------------------
# geometry
def boxes_3d_to_corners(boxes_3d):
    """nms.py:9-27 -> [n,8,3] float64.  With float32 boxes NumPy evaluates
    cos/sin, l/2, w/2 and -h in float32 and the rotation in float64; the
    casts below make that explicit so float64 inputs behave the same way the
    reference does with them."""
    b = np.asarray(boxes_3d)
    n = b.shape[0]
    ft = b.dtype if b.dtype.kind == 'f' else np.float64
    yaw = b[:, 6].astype(ft)
    c = np.cos(yaw).astype(np.float64)
    s = np.sin(yaw).astype(np.float64)
    hl = (b[:, 3].astype(ft) / 2).astype(np.float64)
    hw = (b[:, 5].astype(ft) / 2).astype(np.float64)
    nh = (-b[:, 4].astype(ft)).astype(np.float64)
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float64)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], np.float64)
    top = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.float64)
    cx = hl[:, None] * sx
    cz = hw[:, None] * sz
    cy = nh[:, None] * top
    out = np.empty((n, 8, 3), np.float64)
    out[:, :, 0] = (cx * c[:, None] + cz * s[:, None]) + b[:, None, 0]
    out[:, :, 1] = cy + b[:, None, 1]
    out[:, :, 2] = (cx * -s[:, None] + cz * c[:, None]) + b[:, None, 2]
    return out


def _shoelace(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


class ConvexPolygon(object):
    """Minimal stand-in for shapely.geometry.Polygon on CONVEX rings: `.area`
    and `.intersection(other).area`."""

    def __init__(self, pts):
        self.pts = np.asarray(pts, np.float64).reshape(-1, 2)

    @property
    def area(self):
        return _shoelace(self.pts) if len(self.pts) >= 3 else 0.0

    def _contains(self, q, eps=0.0):
        p = self.pts
        d = np.roll(p, -1, axis=0) - p
        cr = d[:, 0] * (q[1] - p[:, 1]) - d[:, 1] * (q[0] - p[:, 0])
        return bool(np.all(cr >= -eps) or np.all(cr <= eps))

    def intersection(self, other):
        if len(self.pts) < 3 or len(other.pts) < 3 or \
                self.area == 0.0 or other.area == 0.0:
            return ConvexPolygon(np.zeros((0, 2)))
        cand = [q for q in self.pts if other._contains(q)]
        cand += [q for q in other.pts if self._contains(q)]
        a, b = self.pts, other.pts
        for i in range(len(a)):
            p1, p2 = a[i], a[(i + 1) % len(a)]
            r = p2 - p1
            for j in range(len(b)):
                q1, q2 = b[j], b[(j + 1) % len(b)]
                s = q2 - q1
                den = r[0] * s[1] - r[1] * s[0]
                if den == 0.0:
                    continue
                t = ((q1[0] - p1[0]) * s[1] - (q1[1] - p1[1]) * s[0]) / den
                u = ((q1[0] - p1[0]) * r[1] - (q1[1] - p1[1]) * r[0]) / den
                if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
                    cand.append(p1 + t * r)
        if len(cand) < 3:
            return ConvexPolygon(np.zeros((0, 2)))
        pts = np.array(cand)
        ctr = pts.mean(axis=0)
        order = np.argsort(np.arctan2(pts[:, 1] - ctr[1], pts[:, 0] - ctr[0]),
                           kind="stable")
        return ConvexPolygon(pts[order])


def overlapped_boxes_3d_fast_poly(single_box, box_list):
    """nms.py:64-88 on corner arrays: single_box [8,3], box_list [n,8,3]."""
    single_box = np.asarray(single_box)
    box_list = np.asarray(box_list)
    n = len(box_list)
    overlap = np.zeros(n)
    if n == 0:
        return overlap
    hi0, lo0 = single_box.max(axis=0), single_box.min(axis=0)
    hi, lo = box_list.max(axis=1), box_list.min(axis=1)
    apart = np.any((hi0 < lo) | (lo0 > hi), axis=1)
    p1 = ConvexPolygon(single_box[:4][:, [0, 2]])
    a1 = p1.area
    for i in np.nonzero(~apart)[0]:
        p2 = ConvexPolygon(box_list[i][:4][:, [0, 2]])
        shared = p1.intersection(p2).area
        sy = min(hi[i, 1], hi0[1]) - max(lo[i, 1], lo0[1])
        inter = sy * shared
        union = (hi[i, 1] - lo[i, 1]) * p2.area + (hi0[1] - lo0[1]) * a1
        with np.errstate(divide='ignore', invalid='ignore'):
            overlap[i] = np.float32(inter) / (union - inter)
    return overlap


MODES = {"plain": 0, "uncertainty": 1, "merge_only": 2, "score_only": 3}


def nms_boxes_3d(class_labels, boxes_3d, scores, overlapped_thres=0.5,
                 mode="uncertainty", appr_factor=10.0, top_k=-1,
                 attributes=None):
    """nms.py:241-300 for the four variants -> (labels, boxes, scores,
    attributes) of the kept boxes in score order.  Ordering: descending score,
    ties by input order (the reference's `np.argsort(-scores)` leaves tie
    order unspecified, nms.py:93)."""
    labels = np.asarray(class_labels)
    order = np.argsort(-np.asarray(scores), kind="stable")
    if top_k > 0:
        order = order[:top_k]
    labels = labels[order]
    scores = np.array(scores)[order]
    boxes = np.array(boxes_3d)[order]
    attrs = np.arange(len(np.asarray(class_labels)))[order] \
        if attributes is None else np.asarray(attributes)[order]
    merge = mode in ("uncertainty", "merge_only")
    rescore = mode in ("uncertainty", "score_only")
    corners = boxes_3d_to_corners(boxes)
    if mode == "plain":
        corners = np.int32(corners * appr_factor)           # nms.py:115
    m = len(scores)
    keep = np.ones(m, bool)
    for i in range(m - 1):
        if not keep[i]:
            continue
        later = np.nonzero(keep[i + 1:])[0] + i + 1          # still valid
        ov = overlapped_boxes_3d_fast_poly(corners[i], corners[later])
        hit = later[(ov > overlapped_thres) & (labels[later] == labels[i])]
        if merge:
            boxes[i] = np.median(np.concatenate([boxes[hit], boxes[[i]]]),
                                 axis=0)
        if rescore:
            mean_corners = boxes_3d_to_corners(boxes[[i]])[0]
            ov2 = overlapped_boxes_3d_fast_poly(mean_corners, corners[hit])
            scores[i] += np.sum(scores[hit] * ov2)
        keep[hit] = False
    return labels[keep], boxes[keep], scores[keep], attrs[keep]


def synthetic_detections(seed, n_objects=12, votes=(3, 40), noise_boxes=30,
                         num_labels=(1, 3), half_width=30.0, depth=(5.0, 60.0)):
    """Seeded detection-like input: clusters of jittered votes around objects,
    plus isolated boxes.  Scores are distinct (no sort ties)."""
    rng = np.random.default_rng(seed)
    boxes, labels = [], []
    for _ in range(n_objects):
        ctr = np.array([rng.uniform(-half_width, half_width),
                        rng.uniform(0.5, 2.0), rng.uniform(*depth)])
        size = np.array([3.9, 1.5, 1.6]) * rng.uniform(0.8, 1.2, 3)
        yaw = rng.uniform(-np.pi, np.pi)
        lab = int(rng.choice(num_labels))
        for _ in range(int(rng.integers(votes[0], votes[1]))):
            b = np.concatenate([ctr + rng.normal(0, 0.15, 3),
                                size * rng.uniform(0.9, 1.1, 3),
                                [yaw + rng.normal(0, 0.08)]])
            boxes.append(b)
            labels.append(lab)
    for _ in range(noise_boxes):
        boxes.append(np.array([rng.uniform(-half_width, half_width),
                               rng.uniform(0.5, 2.0), rng.uniform(*depth),
                               3.9, 1.5, 1.6,
                               rng.uniform(-np.pi, np.pi)]))
        labels.append(int(rng.choice(num_labels)))
    boxes = np.array(boxes, np.float32)
    labels = np.array(labels, np.int64)
    scores = rng.permutation(len(boxes)).astype(np.float32)
    scores = (0.26 + 0.7 * (scores + rng.uniform(0.1, 0.9, len(boxes))) /
              len(boxes)).astype(np.float32)
    assert len(np.unique(scores)) == len(scores)
    perm = rng.permutation(len(boxes))
    return labels[perm], boxes[perm], scores[perm]


# ---------------------------------------------------------------------------
# run.py:360-412: NMS output -> KITTI label tuples
CLASS_NAMES = {
    'Car': ['Background', 'Car', 'Car', 'DontCare'],
    'Pedestrian_and_Cyclist': ['Background', 'Pedestrian', 'Pedestrian',
                               'Cyclist', 'Cyclist', 'DontCare'],
}


def kitti_labels(class_labels, boxes_3d, probs, cam_to_image, label_method,
                 candidate_xyz, use_box_score=True):
    """Restates run.py:360-412 with the helpers above and
    oracle.labels_oracle (box normals / inside test, run.py:88-99 occlusion).
    """
    from oracle import labels_oracle as LO
    out = []
    corners = boxes_3d_to_corners(boxes_3d)
    for i in range(len(boxes_3d)):
        homo = np.hstack([corners[i], np.ones((8, 1))])
        img = homo @ cam_to_image.T
        xy = (img / img[:, 2:3])[:, :2]
        lo, hi = xy.min(axis=0), xy.max(axis=0)
        cl = np.maximum(lo, 0.0)
        ch = np.minimum(hi, [1242.0, 375.0])
        trunc = 1.0 - (ch[1] - cl[1]) * (ch[0] - cl[0]) / (
            (hi[1] - lo[1]) * (hi[0] - lo[0]))
        if trunc > 0.4:
            continue
        x, y, z, l, h, w, yaw = boxes_3d[i]
        score = probs[i]
        if use_box_score:
            lab = {"x3d": x, "y3d": y, "z3d": z, "yaw": yaw, "height": h,
                   "width": w, "length": l}
            inside = candidate_xyz[LO.sel_xyz_in_box3d(lab, candidate_xyz)]
            occ = 0
            if len(inside):
                n, lower, upper = LO.box_normals(lab)
                p = inside @ n.T
                occ = np.prod((p.max(axis=0) - p.min(axis=0)) /
                              (upper - lower))
            score = (1 + occ) * score
        out.append((CLASS_NAMES[label_method][int(class_labels[i])], -1, -1,
                    0, cl[0], cl[1], ch[0], ch[1], h, w, l, x, y, z, yaw,
                    score))
    return out
