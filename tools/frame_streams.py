#!/usr/bin/env python
"""Alternative schedule: every frame (graph build + GNN) wholly on ONE stream,
N such streams round-robin -- no builder / compute split, no events between
streams.  Needs the capacity form (no host wait inside a frame)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine, concurrent_streams  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4, 6])
    ap.add_argument("--plain", action="store_true",
                    help="ordinary torch streams instead of probed ones")
    args = ap.parse_args()
    dev = torch.device("cuda")
    cfg = configs.get_config("car_auto_T3")
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    pool = []
    for s in range(8):
        xyz, inten = synthetic_cloud(seed=s, preset="car_600k")
        pool.append((torch.from_numpy(xyz).to(dev),
                     torch.from_numpy(inten).to(dev)))
        eng.run_frame(*pool[-1])
    fr = [pool[(i + 5) % 8] for i in range(args.frames)]
    torch.cuda.synchronize()
    for n in args.streams:
        streams = [torch.cuda.Stream() for _ in range(n)] if args.plain else \
            list(concurrent_streams(n))
        for rep in range(2):
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = []
            for i, (x, f) in enumerate(fr):
                with torch.cuda.stream(streams[i % n]):
                    outs.append(eng.run_frame_deferred(x, f))
            torch.cuda.synchronize()
            res = [o.result() for o in outs]
            dt = time.perf_counter() - t0
        print("%d frame stream(s): %.3f ms/frame = %.1f frames/s (overflows %d)"
              % (n, dt / len(fr) * 1e3, len(fr) / dt, eng.deferred_overflows))


if __name__ == "__main__":
    main()
