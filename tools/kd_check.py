#!/usr/bin/env python
"""Quick device check of the kd-tree replica (csrc/kdtree.hip) against the
real scikit-learn + timing of the build.  Run under `timeout`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import graph_gen  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
from sklearn.neighbors import KDTree  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
clouds = {"n1": rng.random((1, 3)), "n31": rng.random((31, 3)),
          "n62": rng.random((62, 3)), "n500": rng.random((500, 3)),
          "n3000": rng.random((3000, 3)), "n9000": rng.random((9000, 3))}
for name in ("tiny", "small", "car", "car_600k", "ped_dense"):
    clouds[name] = synthetic_cloud(seed=0, preset=name)[0]
if len(sys.argv) > 1:      # kd_check.py car ped_dense ...
    clouds = {k: v for k, v in clouds.items() if k in sys.argv[1:]}
for name, xyz in clouds.items():
    xyz = np.ascontiguousarray(xyz, np.float32)
    idx, bounds, status = graph_gen.kdtree_replica(xyz)
    ref = KDTree(xyz.astype(np.float64), leaf_size=30).get_arrays()
    ok = np.array_equal(idx, ref[1]) and np.array_equal(bounds[:, :3], ref[3][0]) \
        and np.array_equal(bounds[:, 3:], ref[3][1])
    x = torch.from_numpy(xyz).to(dev)
    graph_gen.kdtree_replica(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        graph_gen.kdtree_replica(x)
    torch.cuda.synchronize()
    print("%-10s n %6d equal %s status %d  build %.1f us" % (
        name, len(xyz), ok, status, (time.perf_counter() - t0) / 10 * 1e6),
        flush=True)
