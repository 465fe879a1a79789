#!/usr/bin/env python
"""How long does the graph build take beside the GNN kernels of another
frame?  Stream A runs model.predict on a prebuilt graph back to back (the
persistent MFMA kernels), stream B builds graphs; the build is timed with
events alone and beside the load, and the load alone and beside the builds.

    python tools/corun.py [--preset car_600k] [--builds 12]
"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="car_600k")
    ap.add_argument("--config", default="car_auto_T3")
    ap.add_argument("--builds", type=int, default=12)
    ap.add_argument("--host-sized", action="store_true")
    ap.add_argument("--graph-cus", type=int, default=0,
                    help="builder on that many reserved CUs, GNN on the rest "
                         "(CU-masked streams)")
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
    xyz, inten = synthetic_cloud(seed=0, preset=args.preset)
    x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
    eng.run_frame(x, f)
    graph = eng.last_graph
    torch.cuda.synchronize()
    if args.graph_cus > 0:
        sb, scs = eng._pipeline_streams(args.graph_cus)
        sa = scs[0]
    else:
        sa, sb = concurrent_streams(2)

    def load(n):
        with torch.cuda.stream(sa):
            for _ in range(n):
                eng.model.predict(f, *graph, is_training=False)

    def build():
        return eng.build_graph(x) if args.host_sized else \
            eng.build_graph_deferred(x)

    def builds(n):
        evs = []
        with torch.cuda.stream(sb):
            for _ in range(n):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                build()
                e1.record()
                evs.append((e0, e1))
        return evs

    def med(evs):
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)
        return t[len(t) // 2], t[0], t[-1]

    for s in (sa, sb):
        s.wait_stream(torch.cuda.current_stream())
    builds(3)
    load(2)
    torch.cuda.synchronize()
    alone = med(builds(args.builds))
    t0 = time.perf_counter()
    load(8)
    torch.cuda.synchronize()
    gnn_alone = (time.perf_counter() - t0) / 8 * 1e3
    n_load = max(8, int(args.builds * 3.0 / gnn_alone) + 4)
    t0 = time.perf_counter()
    load(n_load)
    time.sleep(0.002)
    evs = builds(args.builds)
    torch.cuda.synchronize()
    both = (time.perf_counter() - t0)
    beside = med(evs)
    print("graph build alone   : median %.3f ms (min %.3f max %.3f)" % alone)
    print("graph build beside  : median %.3f ms (min %.3f max %.3f)" % beside)
    print("GNN alone %.3f ms/frame; %d GNN frames + %d builds together: %.3f ms "
          "(GNN alone would take %.3f)" % (gnn_alone, n_load, args.builds,
                                           both * 1e3, gnn_alone * n_load))


if __name__ == "__main__":
    main()
