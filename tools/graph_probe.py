#!/usr/bin/env python
"""[needs a diagnostic build: python tools/build_variant.py diag core.hip,gnn.hip,graph.hip -DPGNN_DIAG; PGNN_LIB=ab/libdiag.so]
Where the 'gen graph' phase goes (GPU box): wall time of the sequential
graph build of one frame, with / without the kd-tree replica, and the time
the host alone spends inside the calls (measured by running the same calls
while the stream is blocked behind a long-running kernel, so that no host
read has anything to wait for ... not possible with the three count reads,
hence: host share = wall - GPU-busy share from rocprofv3 when run under it)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, graph_gen  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402

dev = torch.device("cuda", 0)
preset = sys.argv[1] if len(sys.argv) > 1 else "car_600k"
cfgname = "ped_cyl_auto_T3" if preset.startswith("ped") else "car_auto_T3"
cfg = configs.get_config(cfgname)
xyz = torch.from_numpy(synthetic_cloud(seed=0, preset=preset)[0]).to(dev)
fn = graph_gen.get_graph_generate_fn(cfg["graph_gen_method"])
kw = cfg["runtime_graph_gen_kwargs"]


def wall(reps=40):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(xyz, **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts[5:])
    return float(np.median(ts)), float(ts.min()), out


for label, dbg in (("kd-tree replica on", 0), ("kd-tree replica off", 1),
                   ("kd-tree replica on", 0)):
    _lib.set_tunable("graph_debug", dbg)
    med, mn, out = wall()
    coords, kps, edges = out
    print("%-22s gen graph median %.0f us  min %.0f us   K %d E0 %d E1 %d" % (
        label, med, mn, coords[1].shape[0], edges[0].shape[0],
        edges[1].shape[0]), flush=True)
_lib.set_tunable("graph_debug", 0)
# the keypoint stage alone and one radius-graph level alone
for name, call in (
        ("keypoints (center)", lambda: graph_gen.keypoints_device(
            xyz, kw["base_voxel_size"] * kw["level_configs"][0]["graph_scale"]
            if False else 0.4 if cfgname.startswith("car") else 0.2, "center")),
        ("radius graph L0", lambda: graph_gen.radius_graph_device(
            coords[0], coords[1],
            kw["level_configs"][0]["graph_gen_kwargs"]["radius"])),
        ("radius graph L1", lambda: graph_gen.radius_graph_device(
            coords[1], coords[2],
            kw["level_configs"][1]["graph_gen_kwargs"]["radius"]))):
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    print("%-22s median %.0f us  min %.0f us" % (
        name, float(np.median(ts[5:])), float(np.min(ts[5:]))), flush=True)
