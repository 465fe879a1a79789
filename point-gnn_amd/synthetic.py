"""Seeded synthetic KITTI-shaped LiDAR clouds.

There is no KITTI data (and no network) on the build or GPU boxes, so every
test and benchmark runs on a ray-cast stand-in for a camera-FOV-cropped
HDL-64E scan: 64 beams (elevation +2 deg .. -24.8 deg), fixed azimuth step over
the camera field of view, sensor 1.73 m above a ground plane, random
car-sized boxes, far building-sized boxes, optional semi-transparent
vegetation volumes ("canopy": a ray is stopped at a uniformly random depth
inside the volume with probability 1-exp(-density*path)), 2-70 m range gate, 0.2 % multiplicative range noise.  Output is in the KITTI *camera* frame
(x right, y down, z forward) as float32, which is what the reference feeds to
its graph generator (run.py:219-222 passes ``cam_rgb_points.xyz``).

This module is host-side input synthesis only; nothing here is on the
measured path.
"""
import numpy as np

__all__ = ["synthetic_cloud", "CLOUD_PRESETS"]

# Scene = flat ground + groups of yawed boxes.  A group is
# (count, r_lo, r_hi, footprint_lo, footprint_hi, height_lo, height_hi): boxes
# placed uniformly in range/azimuth inside the field of view, resting on the
# ground.  Far, tall "building" boxes intercept the upper beams and spread the
# returns over many voxels, which is what sets K (keypoints) and E1.
_CARS = (12, 6.0, 60.0, 1.6, 4.5, 1.4, 1.9)
_FAR16 = (16, 40.0, 68.0, 10.0, 20.0, 6.0, 14.0)

# name -> kwargs.  Shapes measured with the reference's own graph_gen.py at the
# shipped inference kwargs (seed 0): see DESIGN.md "Workloads".
CLOUD_PRESETS = {
    # car_auto_T3 inference shape: N=20k, K~2.9k, E0~350k, E1~500k
    "car": dict(n_points=20000, fov_deg=81.0, az_step_deg=0.1728,
                groups=(_FAR16, _CARS)),
    # north-star shape (BASELINE.json: ~20k points / ~600k level-1 edges):
    # the `car` scene plus ten tree canopies, whose volumetric returns occupy
    # many voxels inside one 4 m ball (mean over seeds 0..7: K~2.9k, E0~377k,
    # E1~602k; E1 ranges 417k..918k)
    "car_600k": dict(n_points=20000, fov_deg=81.0, az_step_deg=0.1728,
                     groups=(_FAR16, _CARS),
                     canopy=((10, 10.0, 45.0, 4.0, 8.0, 4.0, 8.0, 1.5, 0.3),)),
    # config-5 stress: 150 deg FOV, ~50k points, ped_cyl radii/voxel
    "ped_dense": dict(n_points=50000, fov_deg=150.0, az_step_deg=0.15,
                      groups=((28, 35.0, 68.0, 10.0, 20.0, 6.0, 14.0),
                              (30, 5.0, 40.0, 0.4, 0.9, 1.5, 1.9), _CARS)),
    # unit-test size
    "tiny": dict(n_points=1500, fov_deg=40.0, az_step_deg=0.7,
                 groups=((3, 15.0, 30.0, 4.0, 8.0, 3.0, 6.0),
                         (4, 5.0, 25.0, 1.6, 4.5, 1.4, 1.9))),
    "small": dict(n_points=5000, fov_deg=60.0, az_step_deg=0.35,
                  groups=((6, 25.0, 50.0, 6.0, 12.0, 4.0, 8.0),
                          (6, 5.0, 40.0, 1.6, 4.5, 1.4, 1.9))),
}


def _ray_boxes(origin, dirs, centers, sizes, yaws):
    """Slab test of R rays against B yawed boxes.  Returns [R] nearest hit t
    (inf when no box is hit).  Lidar frame: x forward, y left, z up."""
    t_best = np.full(dirs.shape[0], np.inf)
    for c, s, yaw in zip(centers, sizes, yaws):
        cy, sy = np.cos(yaw), np.sin(yaw)
        # rotate ray into the box frame (rotation about z)
        rot = np.array([[cy, sy, 0.0], [-sy, cy, 0.0], [0.0, 0.0, 1.0]])
        o = (origin - c) @ rot.T
        d = dirs @ rot.T
        half = s * 0.5
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            t0 = (-half - o) * inv
            t1 = (half - o) * inv
        tmin = np.minimum(t0, t1).max(axis=1)
        tmax = np.maximum(t0, t1).min(axis=1)
        hit = (tmax >= np.maximum(tmin, 0.0)) & np.isfinite(tmin)
        t = np.where(hit & (tmin > 0), tmin, np.inf)
        t_best = np.minimum(t_best, t)
    return t_best


def _ray_box_span(origin, dirs, center, size, yaw):
    """Entry/exit parameters of R rays through ONE yawed box: (hit, t_in,
    t_out)."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    rot = np.array([[cy, sy, 0.0], [-sy, cy, 0.0], [0.0, 0.0, 1.0]])
    o = (origin - center) @ rot.T
    d = dirs @ rot.T
    half = size * 0.5
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (-half - o) * inv
        t1 = (half - o) * inv
    tmin = np.minimum(t0, t1).max(axis=1)
    tmax = np.maximum(t0, t1).min(axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & np.isfinite(tmin) & (tmin > 0)
    return hit, tmin, tmax


def synthetic_cloud(seed=0, n_points=20000, fov_deg=81.0, az_step_deg=0.1728,
                    groups=(_FAR16, _CARS), preset=None, noise=0.002,
                    canopy=()):
    """Returns (xyz float32 [N,3] camera frame, intensity float32 [N,1]).

    Deterministic in ``seed``.  N = min(n_points, number of valid returns).
    """
    if preset is not None:
        return synthetic_cloud(seed=seed, **CLOUD_PRESETS[preset])
    rng = np.random.default_rng(seed)
    sensor_h = 1.73
    elev = np.deg2rad(np.linspace(2.0, -24.8, 64))
    n_az = int(np.floor(fov_deg / az_step_deg)) + 1
    az = np.deg2rad(-0.5 * fov_deg + az_step_deg * np.arange(n_az))
    el, a = np.meshgrid(elev, az, indexing="ij")
    el = el.ravel()
    a = a.ravel()
    # lidar frame: x forward, y left, z up
    dirs = np.stack([np.cos(el) * np.cos(a), np.cos(el) * np.sin(a),
                     np.sin(el)], axis=1)
    origin = np.zeros(3)
    with np.errstate(divide="ignore"):
        t_hit = np.where(dirs[:, 2] < 0, -sensor_h / dirs[:, 2], np.inf)
    half_fov = np.deg2rad(0.5 * fov_deg)
    for (cnt, r_lo, r_hi, f_lo, f_hi, h_lo, h_hi) in groups:
        if cnt <= 0:
            continue
        rad = r_lo + (r_hi - r_lo) * rng.random(cnt)
        ang = (rng.random(cnt) * 2.0 - 1.0) * half_fov
        sizes = np.stack([f_lo + (f_hi - f_lo) * rng.random(cnt),
                          f_lo + (f_hi - f_lo) * rng.random(cnt),
                          h_lo + (h_hi - h_lo) * rng.random(cnt)], axis=1)
        centers = np.stack([rad * np.cos(ang), rad * np.sin(ang),
                            -sensor_h + 0.5 * sizes[:, 2]], axis=1)
        yaws = rng.random(cnt) * np.pi
        t_hit = np.minimum(t_hit, _ray_boxes(origin, dirs, centers, sizes,
                                             yaws))
    # vegetation: (count, r_lo, r_hi, footprint_lo, footprint_hi, height_lo,
    # height_hi, base height above ground, density 1/m).  Drawn after the solid
    # groups, so presets without canopy keep their random stream.
    for (cnt, r_lo, r_hi, f_lo, f_hi, h_lo, h_hi, base, dens) in canopy:
        rad = r_lo + (r_hi - r_lo) * rng.random(cnt)
        ang = (rng.random(cnt) * 2.0 - 1.0) * half_fov
        sizes = np.stack([f_lo + (f_hi - f_lo) * rng.random(cnt),
                          f_lo + (f_hi - f_lo) * rng.random(cnt),
                          h_lo + (h_hi - h_lo) * rng.random(cnt)], axis=1)
        centers = np.stack([rad * np.cos(ang), rad * np.sin(ang),
                            -sensor_h + base + 0.5 * sizes[:, 2]], axis=1)
        yaws = rng.random(cnt) * np.pi
        for c, s, yaw in zip(centers, sizes, yaws):
            hit, t_in, t_out = _ray_box_span(origin, dirs, c, s, yaw)
            path = np.where(hit, t_out - t_in, 0.0)
            stop = hit & (rng.random(hit.shape[0]) < 1.0 - np.exp(-dens * path))
            t = np.where(stop, t_in + rng.random(hit.shape[0]) * (t_out - t_in),
                         np.inf)
            t_hit = np.minimum(t_hit, t)
    valid = np.isfinite(t_hit) & (t_hit >= 2.0) & (t_hit <= 70.0)
    t = t_hit[valid] * (1.0 + noise * rng.standard_normal(valid.sum()))
    pts = dirs[valid] * t[:, None]
    # lidar (x fwd, y left, z up) -> camera (x right, y down, z fwd)
    cam = np.stack([-pts[:, 1], -pts[:, 2], pts[:, 0]], axis=1)
    rng2 = np.random.default_rng(seed + 1)
    n = min(n_points, cam.shape[0])
    sel = np.sort(rng2.choice(cam.shape[0], size=n, replace=False))
    xyz = np.ascontiguousarray(cam[sel].astype(np.float32))
    intensity = rng2.random((n, 1)).astype(np.float32)
    return xyz, intensity


# ---- synthetic KITTI-object directories (the frame loop's input format) --------
def synthetic_calib(xyz_cam, margin_px=8.0, focal=721.5377):
    """A KITTI-style calibration under which every point of the camera-frame
    cloud `xyz_cam` projects inside the image: (calib file lines, (height,
    width)).  P2 has KITTI's focal length; the principal point and the image
    size are fitted to the cloud (the synthetic scene spans the HDL-64E's
    26.8 deg of elevation, more than a 375-row KITTI image sees, and a frame
    cropped to 375 rows would not be the BASELINE workload any more);
    R0_rect = I; Tr_velo_to_cam is KITTI's axis permutation (x_cam = -y_velo,
    y_cam = -z_velo, z_cam = x_velo) with KITTI's nominal lever arm."""
    x, y, z = (xyz_cam[:, i].astype(np.float64) for i in range(3))
    u, v = focal * x / z, focal * y / z
    cx = float(np.ceil(margin_px - u.min()))
    cy = float(np.ceil(margin_px - v.min()))
    width = int(np.ceil(u.max() + cx + margin_px))
    height = int(np.ceil(v.max() + cy + margin_px))
    p2 = [focal, 0.0, cx, 0.0, 0.0, focal, cy, 0.0, 0.0, 0.0, 1.0, 0.0]
    r0 = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    tr = [0.0, -1.0, 0.0, -0.004, 0.0, 0.0, -1.0, -0.076, 1.0, 0.0, 0.0, -0.272]
    fmt = lambda vals: " ".join("%.12e" % t for t in vals)
    lines = ["P0: %s\n" % fmt(p2), "P1: %s\n" % fmt(p2), "P2: %s\n" % fmt(p2),
             "P3: %s\n" % fmt(p2), "R0_rect: %s\n" % fmt(r0),
             "Tr_velo_to_cam: %s\n" % fmt(tr),
             "Tr_imu_to_velo: %s\n" % fmt(tr)]
    return lines, (height, width)


def write_png_header(path, height, width):
    """A file that starts like a PNG (signature + IHDR): all the frame loop
    ever reads of the image without an `image_reader` is its size."""
    import struct
    import zlib
    ihdr = struct.pack(">IIBBBBB", int(width), int(height), 8, 2, 0, 0, 0)
    chunk = b"IHDR" + ihdr
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + struct.pack(">I", len(ihdr)) + chunk +
                struct.pack(">I", zlib.crc32(chunk) & 0xffffffff))


def write_kitti_frames(root, seeds, preset="car_600k", behind_points=95000):
    """A KITTI-object directory tree under `root` (image_2/, velodyne/,
    calib/; frame names %06d of the seeds) holding, per seed, the synthetic
    scene of `synthetic_cloud(seed, preset)` as a velodyne scan: the camera-FOV
    returns mapped into the lidar frame plus `behind_points` returns outside
    the image (a real HDL-64E sweep has ~120 k points of which ~20 k fall in
    the camera image: the file size and the crop's work are then KITTI's).
    Returns (image_dir, point_dir, calib_dir)."""
    import os
    dirs = [os.path.join(root, d) for d in ("image_2", "velodyne", "calib")]
    for d in dirs:
        os.makedirs(d, exist_ok=True)
    for seed in seeds:
        xyz_cam, inten = synthetic_cloud(seed=seed, preset=preset)
        lines, (height, width) = synthetic_calib(xyz_cam)
        # cam -> velo: inverse of Tr (R0 = I): x_v = z_c + 0.272, y_v = -x_c
        # - 0.004, z_v = -y_c - 0.076
        xc, yc, zc = (xyz_cam[:, i].astype(np.float64) for i in range(3))
        front = np.stack([zc + 0.272, -xc - 0.004, -yc - 0.076], axis=1)
        rng = np.random.default_rng(100000 + seed)
        n_b = int(behind_points)
        az = rng.uniform(0.75 * np.pi, 1.25 * np.pi, n_b)   # behind the car
        el = np.deg2rad(rng.uniform(-24.8, 2.0, n_b))
        r = np.minimum(rng.uniform(3.0, 70.0, n_b),
                       1.73 / np.maximum(np.sin(-el), 1e-3))
        back = np.stack([r * np.cos(el) * np.cos(az),
                         r * np.cos(el) * np.sin(az), r * np.sin(el)], axis=1)
        scan = np.vstack([
            np.hstack([front, inten[:, :1].astype(np.float64)]),
            np.hstack([back, rng.random((n_b, 1))])]).astype(np.float32)
        scan = scan[rng.permutation(len(scan))]
        name = "%06d" % seed
        scan.tofile(os.path.join(dirs[1], name + ".bin"))
        with open(os.path.join(dirs[2], name + ".txt"), "w") as f:
            f.write("".join(lines))
        write_png_header(os.path.join(dirs[0], name + ".png"), height, width)
    return tuple(dirs)
