#!/usr/bin/env python
"""run_frames_on_streams with the message passing on high-priority partner
streams (gnn_priority=True): same logits / boxes as the plain schedule?"""
import os
import sys

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
    frames = []
    for s in range(6):
        xyz, inten = synthetic_cloud(seed=s, preset="car_600k")
        frames.append((torch.from_numpy(xyz).to(dev),
                       torch.from_numpy(inten).to(dev)))
    frames = frames * 3
    a = eng.run_frames_on_streams(frames, 3)
    for rep in range(3):
        b = eng.run_frames_on_streams(frames, 3, gnn_priority=True)
        torch.cuda.synchronize()
        same = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])
                   for x, y in zip(a, b))
        print("gnn_priority rep %d: %d frames, identical = %s" % (
            rep, len(b), same))
        assert same


if __name__ == "__main__":
    main()
