#!/usr/bin/env python
"""Event timeline of the frame pipeline (no profiler): per frame, when the
graph build and the GNN started / ended on their streams, and what the GNN
stream waited for.

    python tools/pipe_events.py [--frames 24] [--tune key=value ...]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--preset", default="car_600k")
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--ahead", type=int, default=1,
                    help="graphs enqueued ahead of the GNN")
    ap.add_argument("--graph-streams", type=int, default=1)
    args = ap.parse_args()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.set_tunable(k, int(v))
    dev = torch.device("cuda")
    cfg = configs.get_config("car_auto_T3")
    eng = InferenceEngine(cfg, weights.init_params(cfg, seed=0, bias_scale=0.05),
                          device=dev)
    pool = []
    for s in range(8):
        xyz, inten = synthetic_cloud(seed=s, preset=args.preset)
        pool.append((torch.from_numpy(xyz).to(dev),
                     torch.from_numpy(inten).to(dev)))
    fr = [pool[(i + 5) % 8] for i in range(args.frames)]
    for x, f in pool:
        eng.run_frame(x, f)
    from pointgnn_amd.engine import concurrent_streams
    st = concurrent_streams(1 + args.graph_streams)
    sc, sgs = st[0], st[1:]
    cur = torch.cuda.current_stream()
    torch.cuda.synchronize()

    def ev():
        return torch.cuda.Event(enable_timing=True)

    for rep in range(2):
        for s in [sc] + list(sgs):
            s.wait_stream(cur)
        G0, G1, C0, C1, built = [], [], [], [], []
        origin = ev()
        origin.record()
        t0 = time.perf_counter()

        def build(i):
            with torch.cuda.stream(sgs[i % len(sgs)]):
                a, b = ev(), ev()
                a.record()
                g = eng.build_graph_deferred(fr[i][0])
                b.record()
            G0.append(a)
            G1.append(b)
            built.append(g)

        for i in range(min(args.ahead, len(fr))):
            build(i)
        for i in range(len(fr)):
            sc.wait_event(G1[i])
            with torch.cuda.stream(sc):
                a, b = ev(), ev()
                a.record()
                coords, kps, edges = built[i]
                for t in list(coords) + list(kps) + list(edges):
                    t.record_stream(sc)
                edges[0]._pgnn_count.frame.tensor.record_stream(sc)
                eng.model.predict(fr[i][1], coords, kps, edges,
                                  is_training=False)
                b.record()
            C0.append(a)
            C1.append(b)
            if i + args.ahead < len(fr):
                build(i + args.ahead)
        host_done = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    print("host enqueued everything after %.2f ms; all done after %.2f ms = "
          "%.3f ms/frame" % (host_done, wall, wall / len(fr)))
    print("%3s %9s %9s %9s | %9s %9s %9s | %s" % (
        "i", "G start", "G end", "G dur", "C start", "C end", "C dur",
        "C start - max(G end, prev C end)"))
    prev_c1 = 0.0
    for i in range(len(fr)):
        g0, g1 = origin.elapsed_time(G0[i]), origin.elapsed_time(G1[i])
        c0, c1 = origin.elapsed_time(C0[i]), origin.elapsed_time(C1[i])
        print("%3d %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f | %+.3f  (%s)" % (
            i, g0, g1, g1 - g0, c0, c1, c1 - c0, c0 - max(g1, prev_c1),
            "waited for graph" if g1 > prev_c1 else "back to back"))
        prev_c1 = c1


if __name__ == "__main__":
    main()
