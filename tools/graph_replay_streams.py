#!/usr/bin/env python
"""Throughput of replayed whole-frame hipGraphs: N captured frames (one per
stream), the bench's frame pool round-robin through them."""
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
    n_frames = 64
    fr = [pool[(i + 5) % 8] for i in range(n_frames)]
    for n in (1, 2, 3, 4):
        caps = [eng.capture_frame(*pool[0]) for _ in range(n)]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = []
            for i, (x, f) in enumerate(fr):
                outs.append(caps[i % n].replay(x, f))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        k = outs[-1].counts.tensor.tolist()
        print("%d replayed graph stream(s): %.3f ms/frame = %.1f frames/s  "
              "(last frame K %d)" % (n, dt / n_frames * 1e3, n_frames / dt, k[0]))
        ref = eng.run_frames_on_streams(fr[:8], max(1, n))
    # note: replays of one graph overwrite its outputs; a consumer would take
    # result() (or copy) before the graph's next replay


if __name__ == "__main__":
    main()
