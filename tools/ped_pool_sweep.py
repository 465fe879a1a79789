#!/usr/bin/env python
"""Sweep of the launch-shape tunables of `ped_cyl`'s pooling kernel (the
LDS-tile fused_mlp_kernel<POOL>: gather + point MLP 4->32->64->128->256->512 +
scatter-max) on the `ped_dense` frame: row sub-tiles per workgroup
(pool_msub), workgroups per CU (mlp_blocks_per_cu), dynamic tile pool share
(mlp_pool_pct), hidden layers through the LDS tile (mlp_debug 1024) -- every
combination in ONE process, same frame, same inputs (profiles/r05_ped_pool_sweep.txt).

    python tools/ped_pool_sweep.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights  # noqa: E402
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


def measure():
    pl = bench.roofline_pool_kernel(torch, eng, reps=6, frame=(x, f))
    return pl["avg_launch_us"], pl["frac"]


base = measure()
print("default (pool_msub auto, mlp_blocks_per_cu 4, mlp_pool_pct 12): "
      "%.1f us  frac %.3f" % base)
print("%-9s %-7s %-9s %-7s %10s %7s" % ("pool_msub", "per_cu", "pool_pct",
                                         "hid_lds", "us", "frac"))
rows = []
for hid in (0, 1024):
    for msub in (0, 1, 2, 3, 4):
        for per_cu in (1, 2, 3, 4):
            for pct in (0, 12, 25):
                _lib.set_tunable("mlp_debug", hid)
                _lib.set_tunable("pool_msub", msub)
                _lib.set_tunable("mlp_blocks_per_cu", per_cu)
                _lib.set_tunable("mlp_pool_pct", pct)
                try:
                    us, frac = measure()
                except Exception as exc:      # a shape the launcher refuses
                    print("%-9d %-7d %-9d %-7d %s" % (msub, per_cu, pct, hid,
                                                     str(exc)[:60]))
                    continue
                rows.append((us, msub, per_cu, pct, hid, frac))
                print("%-9d %-7d %-9d %-7d %10.1f %7.3f" % (msub, per_cu, pct,
                                                            hid, us, frac))
rows.sort()
print("best five:")
for us, msub, per_cu, pct, hid, frac in rows[:5]:
    print("  pool_msub %d per_cu %d pool_pct %d hid_lds %d: %.1f us frac %.3f"
          % (msub, per_cu, pct, hid, us, frac))
_lib.set_tunable("mlp_debug", 0)
_lib.set_tunable("pool_msub", 0)
_lib.set_tunable("mlp_blocks_per_cu", 4)
_lib.set_tunable("mlp_pool_pct", 12)
print("default again: %.1f us  frac %.3f" % measure())
