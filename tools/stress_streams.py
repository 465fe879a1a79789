#!/usr/bin/env python
"""Stress of the frame-stream schedule: a few hundred frames of mixed sizes,
results compared bit for bit with frame-at-a-time execution; outputs of one
batch are dropped while the next batch runs (allocator reuse across streams)."""
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
    pool = []
    for s in range(12):
        xyz, inten = synthetic_cloud(
            seed=s, preset=("car", "small", "car_600k", "tiny")[s % 4])
        pool.append((torch.from_numpy(xyz).to(dev),
                     torch.from_numpy(inten).to(dev)))
    ref = [tuple(t.clone() for t in eng.run_frame(x, f)) for x, f in pool]
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(0)
    bad = 0
    total = 0
    for batch in range(12):
        order = torch.randint(0, len(pool), (32,), generator=g).tolist()
        outs = eng.run_frames_on_streams([pool[i] for i in order],
                                         1 + batch % 4)
        # use the results on the current stream, then drop them
        for i, (lg, bx) in zip(order, outs):
            total += 1
            if not (torch.equal(lg, ref[i][0]) and torch.equal(bx, ref[i][1])):
                bad += 1
        del outs
    torch.cuda.synchronize()
    print("frames %d, mismatches %d, overflow rebuilds %d" % (
        total, bad, eng.deferred_overflows))
    assert bad == 0


if __name__ == "__main__":
    main()
