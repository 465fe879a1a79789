"""CPU restatement of the reference's KITTI frame ingest -- TEST INFRASTRUCTURE
ONLY (imported by tests/, never by the product package).

Follows dataset/kitti_dataset.py: `get_calib` (:483-522),
`velo_points_to_cam` (:998-1006), `cam_points_to_image` (:1036-1052),
`get_cam_points_in_image` (:666-689), `rgb_to_cam_points` (:990-996).

Pinned against the reference's own methods run in the build container
(tests/golden/make_golden_ingest.py calls them on a `KittiDataset` instance
created without `__init__`, under empty `open3d` / `cv2` stubs -- neither is
touched by these methods)."""
import numpy as np

# a KITTI object-detection calibration file (training sample 000000 style
# values; the numbers only need to be plausible, the fixture pins them)
CALIB_LINES = [
    "P0: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 "
    "7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 "
    "1.000000e+00 0.000000e+00\n",
    "P1: 7.215377e+02 0.000000e+00 6.095593e+02 -3.875744e+02 0.000000e+00 "
    "7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 "
    "1.000000e+00 0.000000e+00\n",
    "P2: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 "
    "7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 "
    "1.000000e+00 2.745884e-03\n",
    "P3: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 "
    "7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 "
    "1.000000e+00 2.729905e-03\n",
    "R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 "
    "9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01\n",
    "Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 "
    "1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 "
    "7.523790e-03 1.480755e-02 -2.717806e-01\n",
    "Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 "
    "-7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 "
    "1.482454e-02 9.998881e-01 -7.997231e-01\n",
]
IMAGE_SHAPE = (375, 1242)


def get_calib(lines):
    """kitti_dataset.py:483-522 from the lines of a calib .txt."""
    c = {}
    for line in lines:
        parts = line.split(' ')
        c[parts[0].rstrip(':')] = np.array(parts[1:], dtype=np.float32)
    p2 = c['P2'] = c['P2'].reshape(3, 4)
    r0 = c['R0_rect'] = c['R0_rect'].reshape(3, 3)
    tr = c['Tr_velo_to_cam'] = c['Tr_velo_to_cam'].reshape(3, 4)
    last = np.array([[0, 0, 0, 1]])
    c['velo_to_rect'] = np.vstack([tr, last])                       # :505
    c['cam_to_image'] = np.hstack([p2[:, :3], np.zeros((3, 1), int)])  # :506
    t = np.matmul(np.linalg.inv(p2[:, :3]), p2[:, 3:4])             # :509-510
    c['rect_to_cam'] = np.vstack([np.hstack([r0, t]), last])        # :507-512
    c['velo_to_cam'] = np.matmul(c['rect_to_cam'], c['velo_to_rect'])
    c['cam_to_velo'] = np.linalg.inv(c['velo_to_cam'])
    c['velo_to_image'] = np.matmul(c['cam_to_image'], c['velo_to_cam'])
    return c


def velo_to_cam(xyz, calib):
    """:998-1006 (float32 matmul + float32 translation)."""
    m = calib['velo_to_cam'].T
    out = np.matmul(xyz, m[:3, :3].astype(np.float32))
    out += m[3:4, :3].astype(np.float32)
    return out


def cam_to_image(xyz, calib):
    """:1036-1052 -> [n,3] float64 (u, v, 1)."""
    h = np.hstack([xyz, np.ones((len(xyz), 1))])
    img = np.matmul(h, calib['cam_to_image'].T)
    return img / img[:, 2:3]


def cam_points_in_image(velo_data, calib, image_shape, image=None):
    """:666-689 (+ :990-996 when `image` is given) -> (xyz [m,3] float32,
    attr [m,1] or [m,4] float32, kept scan indices)."""
    velo_data = np.asarray(velo_data, np.float32).reshape(-1, 4)
    cam = velo_to_cam(velo_data[:, :3], calib)
    attr = velo_data[:, 3:4]
    idx = np.arange(len(cam))
    front = cam[:, 2] > 0.1
    cam, attr, idx = cam[front], attr[front], idx[front]
    uv = cam_to_image(cam, calib)
    h, w = image_shape[0], image_shape[1]
    inside = (uv[:, 0] > 0) & (uv[:, 0] < w) & (uv[:, 1] > 0) & (uv[:, 1] < h)
    cam, attr, idx, uv = cam[inside], attr[inside], idx[inside], uv[inside]
    if image is not None:
        rgb = image[np.int32(uv[:, 1]), np.int32(uv[:, 0]), ::-1]
        attr = np.hstack([attr, rgb.astype(np.float32) / 255])
    return cam, attr, idx


def synthetic_velo_scan(seed, n=120000):
    """A 360-degree velodyne-frame scan [n,4] float32: ground returns plus
    random structure, reflectance in [0,1) (x forward, y left, z up)."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(-np.pi, np.pi, n)
    rad = rng.uniform(2.5, 75.0, n) ** 0.85 * 1.9
    z = np.where(rng.uniform(size=n) < 0.6, -1.73 + rng.normal(0, 0.03, n),
                 rng.uniform(-1.7, 1.5, n))
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), z,
                    rng.uniform(0, 1, n)], axis=1)
    return pts.astype(np.float32)


def synthetic_image(seed, shape=IMAGE_SHAPE):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (shape[0], shape[1], 3), dtype=np.uint8)
