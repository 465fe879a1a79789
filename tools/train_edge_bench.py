#!/usr/bin/env python
"""The GNN stage's fused edge kernel at the TRAINING step's shape (two frames,
training graph kwargs, merged: ~70k level-1 edges on ~1.6k vertices), standalone:

    inference entry (maxima only)            pgnn_edge_mlp_scatter_max_fwd
    training entry, rows                     pgnn_edge_mlp_scatter_max_rows_fwd
    training entry, rows + H1                (what csrc/trainer.hip launches)

One HIP-event pair round `reps` launches each.  --tune=key=value as elsewhere.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, gnn, graph_gen, train, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = configs.get_config("car_auto_T3")
eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                      device=dev)
for a in sys.argv[1:]:
    if a.startswith("--tune="):
        k, v = a[len("--tune="):].split("=")
        _lib.set_tunable(k, int(v))
x0, f0 = synthetic_cloud(seed=0, preset="car")
eng.run_frame(torch.from_numpy(x0).to(dev), torch.from_numpy(f0).to(dev))
fn = graph_gen.get_graph_generate_fn(cfg['graph_gen_method'])
lib = _lib.load()
store = eng.model._store
key = [k for k in store._cache if k[0] == 'edge'][0]
c, p_chain, wx_dev, rest = store._cache[key]
wq = int(wx_dev.shape[1])
wa = gnn.padded_width(rest.n_out)
for seeds in ((0, 1), (2, 3)):
    np.random.seed(99)
    frs = []
    for sd in seeds:
        xx, ff = synthetic_cloud(seed=sd, preset="car")
        xx, ff = torch.from_numpy(xx).to(dev), torch.from_numpy(ff).to(dev)
        cs, ks, es = fn(xx, **cfg['graph_gen_kwargs'])
        z = torch.zeros((int(cs[1].shape[0]), 1), device=dev)
        frs.append((ff, cs, ks, es, z, z, z))
    merged = train.batch_data(frs)
    e1 = merged[3][1].contiguous()
    n_k, n_e = int(merged[1][1].shape[0]), int(e1.shape[0])
    p = torch.randn((n_k, wq), device=dev)
    q = torch.randn((n_k, wq), device=dev) * 0.1
    p[:, c:] = 0
    q[:, c:] = 0
    agg = torch.zeros((n_k, wa), device=dev)
    rows = torch.empty((n_e, wa), device=dev)
    h1 = torch.empty((n_e, wq), device=dev)
    srt = 1 if getattr(e1, "_pgnn_sorted", 0) == 1 else 0

    def infer():
        _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd(
            _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e1), n_e,
            n_k, rest.array, rest.n, srt | 2, _lib.ptr(agg), agg.stride(0),
            _lib.ptr(_lib.sched_ws()), _lib.stream_ptr()), "edge")

    def train_rows(with_h1):
        def run():
            _lib.check(lib.pgnn_edge_mlp_scatter_max_rows_fwd(
                _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e1),
                n_e, n_k, rest.array, srt, _lib.ptr(agg), agg.stride(0),
                _lib.ptr(rows), rows.stride(0),
                _lib.ptr(h1) if with_h1 else None, _lib.stream_ptr()),
                "edge rows")
        return run
    flops = 2.0 * rest.k_in * rest.n_out * n_e
    print("frames %s: K %d E1 %d (%.1f edges per vertex, %d row tiles, sorted "
          "%d); MFMA floor %.1f us" % (seeds, n_k, n_e, n_e / n_k,
                                       (n_e + 15) // 16, srt,
                                       flops / 157.3e12 * 1e6))
    for name, fnc in (("inference entry", infer),
                      ("training entry, rows", train_rows(False)),
                      ("training entry, rows + H1", train_rows(True))):
        t = [bench.time_kernel(fnc, 10, torch) * 1e6 for _ in range(3)]
        print("  %-28s %s us" % (name, " ".join("%.1f" % v for v in t)))
