#!/usr/bin/env python
"""Phase stamps of the fused per-vertex kernel (vertex_update_pre_edge_kernel):
shader-clock at entry / tile loaded / each front layer / y written / tail done,
per workgroup (pgnn_set_debug_buffer).

    python tools/krow_timeline.py [--k 3352]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
import torch  # noqa: E402
import runpy  # noqa: E402

ns = runpy.run_path(os.path.join(ROOT, "tools", "krow_bench.py"))
lib, _lib, k = ns["lib"], ns["_lib"], ns["k"]
dev = torch.device("cuda", 0)
n_wg = (k + 15) // 16
buf = torch.zeros(n_wg * 16, dtype=torch.int64, device=dev)
lib.pgnn_set_debug_buffer(_lib.ptr(buf))
ns["run_fused"]()
torch.cuda.synchronize()
lib.pgnn_set_debug_buffer(None)
ts = buf.cpu().numpy().reshape(n_wg, 16)
st = ts[:, :7].astype(np.float64)
names = ["tile load", "front layer 1", "front layer 2", "y + tile refill",
         "offset chain + Q + P"]
print("per-workgroup phase cycles (shader clock), %d workgroups" % n_wg)
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print("  %-24s p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (
        nm, np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
tot = st[:, 5] - st[:, 0]
print("  %-24s p10 %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % (
    "entry -> end", np.percentile(tot, 10), np.median(tot),
    np.percentile(tot, 90), tot.max()))
rt0, rt1 = ts[:, 15], ts[:, 14]
print("entry spread %.2f us; first entry -> last end %.2f us; per-workgroup "
      "entry->end us p50 %.2f max %.2f" % (
          (rt0.max() - rt0.min()) / 100.0, (rt1.max() - rt0.min()) / 100.0,
          np.median(rt1 - rt0) / 100.0, (rt1 - rt0).max() / 100.0))
