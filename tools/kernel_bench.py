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
    ap.add_argument("what", choices=["scatter", "edge", "frame", "detect"])
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
            r["workload"] = {"E": int(edges[1].shape[0]), "C": int(width),
                             "K": n_k}
            print(json.dumps(r))
    elif args.what == "edge":
        r = bench.roofline_edge_kernel(torch, eng, edges[1], n_k,
                                       reps=args.reps)
        print(json.dumps(r))
    elif args.what == "detect":
        # run.py:264-326 after the frame: decode + candidates + NMS.  Two
        # inputs: the frame's own outputs under seeded weights (near-uniform
        # probabilities: a stress case with thousands of candidates), and
        # detection-like votes clustered on objects.
        import time
        from pointgnn_amd import nms
        from oracle import detect_oracle as DO   # input generator only
        logits, boxes = eng.run_frame(x, f)
        probs = torch.softmax(logits, dim=1)
        label_map = {'Background': 0, 'Car': 1, 'DontCare': 3}
        xyz_k = coords[-1]

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                out = fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return out, 1e3 * float(np.median(ts))
        out, ms = timed(lambda: nms.detect_boxes(probs, boxes, xyz_k,
                                                 label_map, 0.01))
        n_c = int(nms.select_candidates(probs)[0].numel())
        print(json.dumps({"case": "frame outputs, seeded weights",
                          "K": n_k, "candidates": n_c,
                          "kept": int(out[0].numel()), "ms": ms}))
        lab, bx, sc = DO.synthetic_detections(0, n_objects=30,
                                              votes=(20, 120))
        tl, tb, tsc = (torch.from_numpy(lab).to(dev),
                       torch.from_numpy(bx).to(dev),
                       torch.from_numpy(sc).to(dev))
        out, ms = timed(lambda: nms.nms_boxes_3d_uncertainty(
            tl, tb, tsc, overlapped_thres=0.01, appr_factor=100.0))
        print(json.dumps({"case": "30 objects x 20-120 votes",
                          "candidates": len(lab), "kept": int(out[0].numel()),
                          "ms": ms}))
    else:
        for _ in range(args.reps):
            eng.run_frame(x, f)
        torch.cuda.synchronize()
        print("frames done", n_k, [int(e.shape[0]) for e in edges])


if __name__ == "__main__":
    main()
