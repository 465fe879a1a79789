#!/usr/bin/env python
"""Time of the kd-tree replica alone (pgnn_kdtree_replica: kd_init + kd_top
levels + kd_subtree) on the headline clouds, HIP events round back-to-back
builds; prints one line per cloud.  PGNN_LIB selects an A/B build.

    python tools/kd_bench.py [--preset car_600k] [--seeds 4]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="car_600k")
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    tot = []
    for seed in range(args.seeds):
        xyz, _ = synthetic_cloud(seed=seed, preset=args.preset)
        p = torch.from_numpy(xyz).to(dev)
        n = int(p.shape[0])
        lv, nodes = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(lib.pgnn_kdtree_shape(n, ctypes.byref(lv),
                                         ctypes.byref(nodes)), "shape")
        ws_bytes = lib.pgnn_kdtree_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        bounds = torch.empty((nodes.value, 6), dtype=torch.float64, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)

        def run():
            _lib.check(lib.pgnn_kdtree_replica(
                _lib.ptr(p), n, _lib.ptr(ws), ws_bytes, _lib.ptr(idx),
                _lib.ptr(bounds), _lib.ptr(status), _lib.stream_ptr()), "kd")
        for _ in range(3):
            run()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        tot.append(us)
        print("seed %d n %d levels %d: %.1f us per build, status %d, "
              "idx checksum %d" % (seed, n, lv.value, us, int(status.item()),
                                   int((idx.long() * torch.arange(
                                       n, device=dev)).sum().item() % 1000003)))
    print("mean %.1f us (%s)" % (sum(tot) / len(tot),
                                 os.environ.get("PGNN_LIB", "default build")))


if __name__ == "__main__":
    main()
