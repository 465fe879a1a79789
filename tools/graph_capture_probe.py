#!/usr/bin/env python
"""Can a whole frame (capacity form: graph build + GNN, ~110 launches through
the C ABI) be captured into ONE hipGraph and replayed?  Prints the
enqueue-to-result latency of eager vs replayed frames and checks that the
replay's logits equal the eager ones."""
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="car_auto_T3")
    ap.add_argument("--preset", default="car_600k")
    ap.add_argument("--reps", type=int, default=9)
    args = ap.parse_args()
    dev = torch.device("cuda")
    cfg = configs.get_config(args.config)
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    xyz, inten = synthetic_cloud(seed=0, preset=args.preset)
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
    lg0, bx0 = eng.run_frame(x, f)
    print("shapes", eng.frame_shapes[-1], flush=True)
    k = lg0.shape[0]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # warm every cache on the capture stream
        for _ in range(3):
            eng.run_frame_deferred(x, f).result()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    xs, fs = x.clone(), f.clone()          # static inputs of the graph
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        d = eng.run_frame_deferred(xs, fs)
    torch.cuda.synchronize()
    print("captured", flush=True)
    lat = {"eager": [], "replay": []}
    for rep in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_frame_deferred(x, f).result()
        torch.cuda.synchronize()
        lat["eager"].append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        counts = d.counts.tensor.tolist()
        torch.cuda.synchronize()
        lat["replay"].append((time.perf_counter() - t0) * 1e3)
        print("rep", rep, "ok", counts[:2], flush=True)
    print("K", counts[0], "status", counts[1], "edges", counts[2:])
    print("latency ms: eager %.3f  replay %.3f" % (
        sorted(lat["eager"])[len(lat["eager"]) // 2],
        sorted(lat["replay"])[len(lat["replay"]) // 2]))
    print("replay == eager:", torch.equal(d.logits[:k], lg0),
          torch.equal(d.boxes[:k], bx0))
    # another cloud of the same size through the same graph
    xyz2, inten2 = synthetic_cloud(seed=3, preset=args.preset)
    lg2, bx2 = eng.run_frame(torch.from_numpy(xyz2).to(dev),
                             torch.from_numpy(inten2).to(dev))
    xs.copy_(torch.from_numpy(xyz2).to(dev))
    fs.copy_(torch.from_numpy(inten2).to(dev))
    g.replay()
    c2 = d.counts.tensor.tolist()
    k2 = c2[0]
    print("second cloud: K", k2, "overflow", c2[3] > c2[2] or c2[5] > c2[4],
          "equal", torch.equal(d.logits[:k2], lg2), torch.equal(d.boxes[:k2], bx2))


if __name__ == "__main__":
    main()
