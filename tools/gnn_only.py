#!/usr/bin/env python
"""Upper bound of the frame pipeline: the GNN of the bench's frame pool with
the graphs prebuilt (no builder on the device at all), on 1 and 2 streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine, concurrent_streams  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def main():
    dev = torch.device("cuda")
    cfg = configs.get_config("car_auto_T3")
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    pool = []
    for s in range(8):
        xyz, inten = synthetic_cloud(seed=s, preset="car_600k")
        x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
        eng.run_frame(x, f)
        pool.append((f, eng.last_graph))
    torch.cuda.synchronize()
    streams = concurrent_streams(5)
    n = 64
    for ns in (1, 2, 3, 4):
        for rep in range(2):
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                f, g = pool[(i + 5) % 8]
                with torch.cuda.stream(streams[1 + i % ns]):
                    eng.model.predict(f, *g, is_training=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("GNN only, %d stream(s): %.3f ms/frame = %.1f frames/s" % (
            ns, dt / n * 1e3, n / dt))


if __name__ == "__main__":
    main()
