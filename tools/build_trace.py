#!/usr/bin/env python
"""The graph build of ONE frame alone on an idle device, in capacity form:
in order on one stream, then with its independent parts on side streams
(graph_gen `overlap_build`).  Run under `rocprofv3 --kernel-trace` and dump
the last milliseconds with tools/trace_dump.py to see which kernels really
overlapped (tools/sessions/r03_s30.sh)."""
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
    xyz, inten = synthetic_cloud(seed=0, preset="car_600k")
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
    eng.run_frame(x, f)
    torch.cuda.synchronize()
    main = torch.cuda.Stream() if "--stream" in sys.argv else \
        torch.cuda.current_stream()
    print("main stream: %s" % ("a torch stream" if "--stream" in sys.argv
                               else "the default stream"))
    with torch.cuda.stream(main):
        run(eng, x)


def run(eng, x):
    order = [False] * 6 + [True] * 6     # the trace ends with overlapped builds
    for overlap in order:
        torch.cuda.synchronize()
        t = time.perf_counter()
        g = eng.build_graph_deferred(x, overlap=overlap)
        t_enq = time.perf_counter()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        del g
        print("overlap %d: enqueue %.3f ms, done %.3f ms" % (
            overlap, (t_enq - t) * 1e3, (t_end - t) * 1e3))
        time.sleep(0.002)


if __name__ == "__main__":
    main()
