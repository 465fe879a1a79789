#!/usr/bin/env python
"""Per-tile phase timeline of the fused edge kernel (s_memtime stamps written by
the kernel itself through pgnn_set_debug_buffer): average cycles per phase and
how the two co-resident workgroups of a CU (blocks b and b + grid/2, verified
with HW_ID once) overlap."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = configs.get_config("car_auto_T3")
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                      device=dev)
xyz, inten = synthetic_cloud(seed=0, preset="car")
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
eng.run_frame(x, f)
coords, kps, edges = eng.last_graph
n_k = int(coords[1].shape[0])
if "--local-src" in sys.argv:
    # experiment: what if keypoints were numbered spatially (src close to dst)?
    e1 = edges[1].clone()
    e1[:, 0] = torch.clamp(e1[:, 1] + (e1[:, 0] % 128) - 64, 0, n_k - 1)
    edges = [edges[0], e1.contiguous()]
lib = _lib.load()
for a in sys.argv[1:]:
    if a.startswith("--tune="):
        k, v = a[len("--tune="):].split("=")
        _lib.set_tunable(k, int(v))
buf = torch.zeros(1024 * 32 * 8, dtype=torch.int64, device=dev)
bench.roofline_edge_kernel(torch, eng, edges[1], n_k, reps=3)   # warm
lib.pgnn_set_debug_buffer(_lib.ptr(buf))
r = bench.roofline_edge_kernel(torch, eng, edges[1], n_k, reps=1)
torch.cuda.synchronize()
lib.pgnn_set_debug_buffer(None)
grid = 512
ts = buf.cpu().numpy().reshape(1024, 32, 8)[:grid]
valid = ts[:, :, 0] > 0
t0, t1, t2, t3 = ts[:, :, 0], ts[:, :, 1], ts[:, :, 2], ts[:, :, 3]


def stat(name, a):
    a = a[valid]
    print("%-28s mean %7.0f  p50 %7.0f  p90 %7.0f" % (
        name, a.mean(), np.median(a), np.percentile(a, 90)))


print("kernel us %.1f, tiles stamped %d" % (r["avg_launch_us"], valid.sum()))
stat("indices -> LDS (cycles)", t3 - t0)
t4, t5 = ts[:, :, 4], ts[:, :, 5]
stat("  index load latency", t4 - t3)
stat("  wave0 gather+LDS writes", t5 - t4)
stat("  barrier wait (others)", t1 - t5)
stat("gather P,Q -> LDS", t1 - t3)
stat("GEMM + epilogue", t2 - t1)
per = (t2[:, 1:] - t2[:, :-1])[valid[:, 1:] & valid[:, :-1]]
print("%-28s mean %7.0f  p50 %7.0f" % ("tile period", per.mean(), np.median(per)))
ov = []
for a in range(grid // 2):
    b = a + grid // 2
    ia = [(t1[a, i], t2[a, i]) for i in range(32) if valid[a, i]]
    ib = [(t1[b, i], t2[b, i]) for i in range(32) if valid[b, i]]
    tot = sum(e - s for s, e in ia)
    both = sum(max(0, min(e, e2) - max(s, s2)) for s, e in ia for s2, e2 in ib)
    ov.append(both / max(tot, 1))
print("fraction of a WG's GEMM+epilogue time overlapping its CU partner's: "
      "%.2f" % np.mean(ov))

# shader clock during the kernel: cycle counter vs the constant 100 MHz counter
t6 = ts[:, :, 6]
clk = []
for w in range(grid):
    idx = [i for i in range(32) if valid[w, i]]
    if len(idx) >= 4 and t6[w, idx[-1]] > t6[w, idx[0]]:
        clk.append((t0[w, idx[-1]] - t0[w, idx[0]]) /
                   (t6[w, idx[-1]] - t6[w, idx[0]]) * 0.1)
print("cycle-counter rate during the kernel: %.3f GHz (p10 %.3f, p90 %.3f)"
      % (np.mean(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
