#!/usr/bin/env python
"""Step latency of four kinds of dependent chains (VALU, barrier, L2 pointer
chase, LDS) alone and beside the GNN kernels of a frame, at wave priority 0
and 3 (tools/micro/corun_probe.hip; build: see tools/r03_s7.sh)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine, concurrent_streams  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "ab", "libprobe.so"))
    lib.probe_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
    dev = torch.device("cuda")
    cfg = configs.get_config("car_auto_T3")
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    xyz, inten = synthetic_cloud(seed=0, preset="car_600k")
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
    eng.run_frame(x, f)
    graph = eng.last_graph
    sa, sb = concurrent_streams(2)
    n = 1 << 16
    nxt = torch.randperm(n, device=dev).to(torch.int32)
    out = torch.zeros(4, device=dev)
    torch.cuda.synchronize()
    for s in (sa, sb):
        s.wait_stream(torch.cuda.current_stream())

    def load(k):
        with torch.cuda.stream(sa):
            for _ in range(k):
                eng.model.predict(f, *graph, is_training=False)

    cases = [("VALU chain, 1 wave", 0, 1, 64, 20000),
             ("VALU chain, 256 thr x 8 wg", 0, 8, 256, 20000),
             ("barrier chain, 256 thr", 1, 1, 256, 2000),
             ("barrier chain, 1024 thr", 1, 1, 1024, 2000),
             ("barrier chain, 256 thr x 10 wg", 1, 10, 256, 2000),
             ("L2 pointer chase, 1 wave", 2, 1, 64, 2000),
             ("L2 pointer chase, 256 thr x 8 wg", 2, 8, 256, 2000),
             ("LDS chain, 1 wave", 3, 1, 64, 20000)]
    print("%-36s %12s %12s %12s %12s" % ("ns per step", "alone p0", "alone p3",
                                         "beside p0", "beside p3"))
    for name, kind, blocks, threads, iters in cases:
        row = []
        for beside in (False, True):
            for pr in (0, 1):
                if beside:
                    load(6)
                ts = []
                with torch.cuda.stream(sb):
                    for rep in range(6):
                        e0 = torch.cuda.Event(enable_timing=True)
                        e1 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                        lib.probe_launch(kind, pr, blocks, threads, iters,
                                         nxt.data_ptr(), out.data_ptr(),
                                         sb.cuda_stream)
                        e1.record()
                        ts.append((e0, e1))
                torch.cuda.synchronize()
                t = sorted(a.elapsed_time(b) for a, b in ts)
                row.append(t[len(t) // 2] * 1e6 / iters)
        print("%-36s %12.1f %12.1f %12.1f %12.1f" % ((name,) + tuple(row)))


if __name__ == "__main__":
    main()
