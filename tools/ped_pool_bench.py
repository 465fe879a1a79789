#!/usr/bin/env python
"""The ped_cyl pooling and edge kernels standalone on the ped_dense frame
(bench.roofline_pool_kernel / roofline_edge_kernel), for A/B runs of library
variants (PGNN_LIB=ab/lib<name>.so).

    python tools/ped_pool_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = configs.get_config("ped_cyl_auto_T3")
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                      device=dev)
xyz, inten = synthetic_cloud(seed=0, preset="ped_dense")
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
eng.run_frame(x, f)
coords, kps, edges = eng.last_graph
pl = bench.roofline_pool_kernel(torch, eng, frame=(x, f))
ed = bench.roofline_edge_kernel(torch, eng, edges[1], int(coords[1].shape[0]),
                                frame=(x, f))
print("lib %s" % os.environ.get("PGNN_LIB", "(tree)"))
print("  ped pooling %8.1f us  frac %.3f" % (pl["avg_launch_us"], pl["frac"]))
print("  ped edge    %8.1f us  frac %.3f" % (ed["avg_launch_us"], ed["frac"]))
