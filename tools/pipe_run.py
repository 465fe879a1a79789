#!/usr/bin/env python
"""The bench's steady-state frame pipeline and nothing else (for
`rocprofv3 --kernel-trace`): warm-up, then N pipelined frames of a preset.

    python tools/pipe_run.py [--frames 12] [--preset car_600k] [--host-sized]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--pool", type=int, default=8)
    ap.add_argument("--preset", default="car_600k")
    ap.add_argument("--config", default="car_auto_T3")
    ap.add_argument("--compute-streams", type=int, default=2)
    ap.add_argument("--host-sized", action="store_true")
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    from pointgnn_amd import _lib
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.set_tunable(k, int(v))
    dev = torch.device("cuda")
    cfg = configs.get_config(args.config)
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    pool = []
    for s in range(args.pool):
        xyz, inten = synthetic_cloud(seed=s, preset=args.preset)
        pool.append((torch.from_numpy(xyz).to(dev),
                     torch.from_numpy(inten).to(dev)))
    fr = [pool[i % len(pool)] for i in range(args.frames)]
    for _ in range(2):
        eng.run_frames_pipelined(fr[:8], compute_streams=args.compute_streams,
                                 deferred=not args.host_sized)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_frames_pipelined(fr, compute_streams=args.compute_streams,
                             deferred=not args.host_sized)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("frames/s %.1f  (%.3f ms/frame)" % (len(fr) / dt, dt / len(fr) * 1e3))
    print("shapes", eng.frame_shapes[-len(fr):])


if __name__ == "__main__":
    main()
