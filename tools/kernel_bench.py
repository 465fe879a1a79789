#!/usr/bin/env python
"""Micro-benchmarks of single kernels on a real frame graph (GPU box).

    python tools/kernel_bench.py scatter [--sweep]   # standalone scatter-max
    python tools/kernel_bench.py edge                # fused edge kernel
    python tools/kernel_bench.py frame               # a few whole frames

Used under rocprofv3 --pmc ... to attribute counters to one kernel, and alone
for in-process A/B sweeps of tunables (interleaved rounds, median)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["scatter", "edge", "frame"])
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--config", default="car_auto_T3")
    ap.add_argument("--preset", default="car")
    ap.add_argument("--tune", action="append", default=[],
                    help="key=value tunable (repeatable)")
    args = ap.parse_args()
    import torch
    import pointgnn_amd  # noqa
    from pointgnn_amd import _lib, configs, weights, gnn
    from pointgnn_amd.engine import InferenceEngine
    from pointgnn_amd.synthetic import synthetic_cloud
    import bench
    dev = torch.device("cuda", 0)
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.set_tunable(k, int(v))
    cfg = configs.get_config(args.config)
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    xyz, inten = synthetic_cloud(seed=0, preset=args.preset)
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
    eng.run_frame(x, f)
    coords, kps, edges = eng.last_graph
    n_k = int(coords[1].shape[0])
    width = cfg['model_kwargs']['layer_configs'][0]['kwargs'][
        'output_MLP_depth_list'][-1]
    if args.what == "scatter":
        if args.sweep:
            res = {}
            for rnd in range(3):
                for rpw in (64, 128, 256, 512, 1024, 2048):
                    _lib.set_tunable("scatter_rows_per_wave", rpw)
                    r = bench.roofline_scatter_max(torch, edges[1], n_k, width,
                                                   reps=args.reps)
                    res.setdefault(rpw, []).append(r["achieved"])
            for rpw, v in res.items():
                print("rows_per_wave %4d: GB/s %s median %.0f" % (
                    rpw, ["%.0f" % a for a in v], float(np.median(v))))
        else:
            r = bench.roofline_scatter_max(torch, edges[1], n_k, width,
                                           reps=args.reps)
            print(json.dumps(r))
    elif args.what == "edge":
        r = bench.roofline_edge_kernel(torch, eng, edges[1], n_k,
                                       reps=args.reps)
        print(json.dumps(r))
    else:
        for _ in range(args.reps):
            eng.run_frame(x, f)
        torch.cuda.synchronize()
        print("frames done", n_k, [int(e.shape[0]) for e in edges])


if __name__ == "__main__":
    main()
