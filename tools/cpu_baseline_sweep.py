#!/usr/bin/env python
"""Thread-count x chunk sweep of the cpu_baseline's GNN port
(oracle/gnn_oracle_torch.py) on THIS host: the whole car_600k seed-0 frame,
1 warm-up + best of 3 per cell, next to the NumPy sgemm probe bench.py's
cpu_baseline leg reports.  -> profiles/r06_cpu_baseline_sweep.txt

    python tools/cpu_baseline_sweep.py [out.txt]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
from oracle import graph_oracle as go  # noqa: E402
from oracle import gnn_oracle_torch as gn  # noqa: E402


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    cfg = configs.get_config("car_auto_T3")
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    xyz, inten = synthetic_cloud(seed=0, preset="car_600k")
    coords, kps, edges = go.multi_level_graph(
        xyz, **cfg['runtime_graph_gen_kwargs'])
    n_k, e0, e1 = coords[1].shape[0], len(edges[0]), len(edges[1])
    flops = bench.algorithmic_flops_per_frame(cfg, n_k, e0, e1)
    allowed = len(os.sched_getaffinity(0))
    a = np.random.default_rng(0).standard_normal((4096, 304)).astype(np.float32)
    b = np.random.default_rng(1).standard_normal((304, 304)).astype(np.float32)
    a @ b
    t = time.perf_counter()
    for _ in range(10):
        a @ b
    probe = 10 * 2 * 4096 * 304 * 304 / (time.perf_counter() - t) / 1e9
    print("host: %s" % bench._host_description(), file=out)
    print("allowed CPUs %d; frame K %d E0 %d E1 %d = %.1f GFLOP (algorithmic); "
          "NumPy sgemm probe [4096x304]x[304x304]: %.0f GFLOP/s"
          % (allowed, n_k, e0, e1, flops / 1e9, probe), file=out)
    print("%8s %8s %8s %9s %9s %7s" % ("workers", "intraop", "chunk", "best_s",
                                       "GFLOP/s", "/probe"), file=out)
    cells = []
    for intra in (32, 64):
        cells.append((0, intra, 1 << 17))
    for w in (16, 32, 64, 96, 128, 192, 256):
        if w > allowed:
            continue
        for chunk in (1024, 2048, 4096, 8192):
            cells.append((w, 1, chunk))
    best = None
    for w, intra, chunk in cells:
        torch.set_num_threads(min(intra, allowed))
        gn.WORKERS, gn.CHUNK_ROWS = w, chunk
        gn.predict(params, cfg, inten, coords, kps, edges)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            gn.predict(params, cfg, inten, coords, kps, edges)
            ts.append(time.perf_counter() - t)
        g = flops / min(ts) / 1e9
        print("%8d %8d %8d %9.3f %9.0f %7.3f" % (w, intra, chunk, min(ts), g,
                                                 g / probe), file=out)
        out.flush()
        if best is None or g > best[0]:
            best = (g, w, intra, chunk)
    print("best: %.0f GFLOP/s = %.3f of the probe at workers %d, intra-op %d, "
          "chunk %d" % (best[0], best[0] / probe, best[1], best[2], best[3]),
          file=out)


if __name__ == "__main__":
    main()
