#!/usr/bin/env python
"""Per-tile phase timeline of the fused POOLING kernel (same stamps as
tools/tile_timeline.py: t0 tile start, t1 after the prologue barrier, t2 after
the epilogue barrier)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights, gnn, graph_gen, models  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402

dev = torch.device("cuda", 0)
cfg = configs.get_config("car_auto_T3")
params = weights.init_params(cfg, seed=0, bias_scale=0.05)
xyz, inten = synthetic_cloud(seed=0, preset="car")
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
fn = graph_gen.get_graph_generate_fn(cfg["graph_gen_method"])
coords, kps, edges = fn(x, **cfg["runtime_graph_gen_kwargs"])
model = models.get_model(cfg["model_name"])(
    num_classes=4, box_encoding_len=7, mode="test",
    **cfg["model_kwargs"]).load_state_dict(params)
lib = _lib.load()
for a in sys.argv[1:]:
    if a.startswith("--tune="):
        k, v = a[len("--tune="):].split("=")
        _lib.set_tunable(k, int(v))
kw = cfg["model_kwargs"]["layer_configs"][0]["kwargs"]


def run():
    with gnn.parameters(model._store), gnn.variable_scope("layer1"):
        return gnn.PointSetPooling().apply_regular(f, coords[0], kps[0],
                                                   edges[0], **kw)


for _ in range(3):
    run()
buf = torch.zeros(1024 * 32 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib.pgnn_set_debug_buffer(_lib.ptr(buf))
# only the fused pooling kernel stamps; the output MLP that follows is the
# 8-wave rows kernel (no stamps)
run()
torch.cuda.synchronize()
lib.pgnn_set_debug_buffer(None)
grid = 512
ts = buf.cpu().numpy().reshape(1024, 32, 8)[:grid]
valid = ts[:, :, 0] > 0
t0, t1, t2 = ts[:, :, 0], ts[:, :, 1], ts[:, :, 2]


def stat(name, a):
    a = a[valid]
    print("%-28s mean %7.0f  p50 %7.0f  p90 %7.0f" % (
        name, a.mean(), np.median(a), np.percentile(a, 90)))


print("E0 %d, tiles stamped %d" % (int(edges[0].shape[0]), valid.sum()))
stat("prologue (gather -> LDS)", t1 - t0)
stat("4 layers + epilogue", t2 - t1)
per = (t2[:, 1:] - t2[:, :-1])[valid[:, 1:] & valid[:, :-1]]
print("%-28s mean %7.0f  p50 %7.0f" % ("tile period", per.mean(), np.median(per)))
print("MFMA floor per tile and wave: 784 MFMAs x 32 cycles = 25088 cycles")
t7, kind = ts[:, :, 7], ts[:, :, 3]
stat("  3 hidden layers", t7 - t1)
for name, m in (("fast (one segment)", kind == 1), ("few runs (registers)", kind == 2),
                ("general epilogue", kind == 0)):
    a = (t2 - t7)[valid & m]
    if a.size:
        print("  last layer, %-18s n %5d  mean %7.0f  p50 %7.0f  p90 %7.0f" % (
            name, a.size, a.mean(), np.median(a), np.percentile(a, 90)))
