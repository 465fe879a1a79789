"""CPU restatement of the reference's training-target construction -- TEST
INFRASTRUCTURE ONLY (imported by tests/, never by the product package).

Follows dataset/kitti_dataset.py: `get_label` (:703-751),
`box3d_to_cam_points` (:85-116), `box3d_to_normals` (:118-141),
`sel_xyz_in_box3d` (:143-162) and the three
`assign_classaware_*_label_to_points` (:1132-1284); then train.py:120-130
(encode in float64, cast to float32).

Pinned against the reference's own functions run in the build container
(tests/golden/make_golden_labels.py)."""
import numpy as np

LABEL_MAPS = {
    "yaw": ({'Background': 0, 'Car': 1, 'Pedestrian': 3, 'Cyclist': 5,
             'DontCare': 7}, 8),
    "Car": ({'Background': 0, 'Car': 1, 'DontCare': 3}, 4),
    "Pedestrian_and_Cyclist": ({'Background': 0, 'Pedestrian': 1,
                                'Cyclist': 3, 'DontCare': 5}, 6),
}
_KEYS = ('truncation', 'occlusion', 'alpha', 'xmin', 'ymin', 'xmax', 'ymax',
         'height', 'width', 'length', 'x3d', 'y3d', 'z3d', 'yaw')


def get_label(path, difficulty=-100):
    """:703-751."""
    limits = {0: (0.15, 0, 40), 1: (0.3, 1, 25), 2: (0.5, 2, 25)}
    out = []
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        f = line.split(' ')
        lab = {'name': f[0]}
        for k, v in zip(_KEYS, f[1:15]):
            lab[k] = int(v) if k == 'occlusion' else float(v)
        if len(f) > 15:
            lab['score'] = float(f[15])
        if difficulty > -1:
            t, o, h = limits[difficulty]
            if lab['truncation'] > t or lab['occlusion'] > o or \
                    (lab['ymax'] - lab['ymin']) < h:
                continue
        out.append(lab)
    return out


def box_corners(label, expend=(1.0, 1.0, 1.0)):
    """:85-116 -> [8,3] float64."""
    yaw, h = label['yaw'], label['height']
    dh = h * (expend[0] - 1)
    w = label['width'] * expend[1]
    l = label['length'] * expend[2]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * (l / 2)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * (w / 2)
    sy = np.array([dh / 2] * 4 + [-h - dh / 2] * 4)
    local = np.stack([sx, sy, sz], axis=1)
    rot = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0],
                    [-np.sin(yaw), 0, np.cos(yaw)]])
    return local.dot(rot.T) + np.array([label['x3d'], label['y3d'],
                                        label['z3d']])


def box_normals(label, expend=(1.0, 1.0, 1.0)):
    """:118-141 -> (normals [3,3], lower [3], upper [3])."""
    p = box_corners(label, expend)
    w = np.stack([p[0] - p[4], p[0] - p[1], p[0] - p[3]])
    lower = np.array([w[0] @ p[4], w[1] @ p[1], w[2] @ p[3]])
    upper = w @ p[0]
    return w, lower, upper


def sel_xyz_in_box3d(label, xyz, expend=(1.0, 1.0, 1.0)):
    """:143-162."""
    w, lower, upper = box_normals(label, expend)
    proj = np.matmul(xyz, w.T)
    return np.all((proj > lower) & (proj < upper), axis=1)


def assign_labels(labels, xyz, expend, label_method):
    """:1132-1284 -> (cls_labels [K,1] int64, boxes_3d [K,1,7] float64,
    valid_boxes [K,1,1] float32, label_map)."""
    label_map, _ = LABEL_MAPS[label_method]
    dont_care = label_map['DontCare']
    k = xyz.shape[0]
    cls = np.zeros((k, 1), np.int64)
    boxes = np.zeros((k, 1, 7))
    valid = np.zeros((k, 1, 1), np.float32)
    for lab in labels:
        c = label_map.get(lab['name'], dont_care)
        if 1 <= c < dont_care:
            m = sel_xyz_in_box3d(lab, xyz, expend)
            yaw = lab['yaw']
            while yaw < -0.25 * np.pi:
                yaw += np.pi
            while yaw > 0.75 * np.pi:
                yaw -= np.pi
            cls[m, :] = c if yaw < 0.25 * np.pi else c + 1
            boxes[m, 0, :] = (lab['x3d'], lab['y3d'], lab['z3d'],
                              lab['length'], lab['height'], lab['width'], yaw)
            valid[m, 0, :] = 1
        elif lab['name'] != 'DontCare':
            m = sel_xyz_in_box3d(lab, xyz, expend)
            cls[m, :] = c
            valid[m, 0, :] = 0.0
    return cls, boxes, valid, label_map


def synthetic_vertices(seed, k=6000):
    """Keypoint-like vertices [k,3] float32 in the camera frame."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-12, 12, k), rng.uniform(0.2, 1.8, k),
                     rng.uniform(4, 40, k)], axis=1).astype(np.float32)


def synthetic_labels(seed, xyz, n_boxes=24, names=(
        'Car', 'Car', 'Car', 'Van', 'Pedestrian', 'Cyclist', 'DontCare',
        'Truck', 'Person_sitting')):
    """Label dictionaries centred on random points of `xyz` (so that boxes
    contain vertices), with deliberate overlaps: every third box reuses the
    previous centre with another class/yaw, which exercises the overwrite
    order."""
    rng = np.random.default_rng(seed)
    sizes = {'Car': (3.9, 1.5, 1.6), 'Van': (5.0, 2.1, 1.9),
             'Pedestrian': (0.9, 1.8, 0.7), 'Cyclist': (1.8, 1.7, 0.6),
             'DontCare': (4.0, 1.5, 1.6), 'Truck': (10.0, 3.3, 2.6),
             'Person_sitting': (0.8, 1.3, 0.6)}
    out, ctr = [], None
    for i in range(n_boxes):
        name = names[int(rng.integers(len(names)))]
        l, h, w = np.array(sizes[name]) * rng.uniform(0.85, 1.2, 3)
        if i % 3 != 2 or ctr is None:
            ctr = xyz[int(rng.integers(len(xyz)))].astype(np.float64)
        c = ctr + rng.normal(0, 0.3, 3)
        out.append({
            'name': name, 'truncation': float(rng.uniform(0, 0.6)),
            'occlusion': int(rng.integers(0, 4)),
            'alpha': float(rng.uniform(-3, 3)),
            'xmin': 100.0, 'ymin': 100.0, 'xmax': 200.0,
            'ymax': float(100 + rng.uniform(10, 80)),
            'height': float(h), 'width': float(w), 'length': float(l),
            'x3d': float(c[0]), 'y3d': float(c[1] + h / 2), 'z3d': float(c[2]),
            'yaw': float(rng.uniform(-2 * np.pi, 2 * np.pi))})
    return out


def write_label_file(path, labels):
    with open(path, 'w') as f:
        for lab in labels:
            f.write(' '.join([lab['name']] + [repr(lab[k]) for k in _KEYS])
                    + '\n')
        f.write('\n')
