#!/usr/bin/env python
"""What does each part of the graph build cost the GNN kernels it runs beside?
The GNN of the bench's frame pool runs on two streams with prebuilt graphs; a
third stream runs, once per frame, ONE part of the builder (nothing / the
kd-tree replica / the whole keypoint stage / the two radius graphs / all of
it) on a fixed cloud.  Reported: ms per GNN frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, graph_gen as G, weights  # noqa: E402
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
        pool.append((x, f, eng.last_graph))
    x0 = pool[0][0]
    coords0 = pool[0][2][0]
    kw = cfg['runtime_graph_gen_kwargs']
    lv = kw['level_configs']
    voxel = float(kw['base_voxel_size']) * lv[0]['graph_scale']
    counts = torch.zeros(8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s1, s2, sb = concurrent_streams(3)

    def part_none():
        pass

    def part_kd():
        G.kdtree_replica(x0)            # device tensors in: no host read ...
    # (kdtree_replica reads the status back: use the raw entry instead)
    lib = _lib.load()
    import ctypes
    n = int(x0.shape[0])
    lvl, nodes = ctypes.c_int32(), ctypes.c_int32()
    lib.pgnn_kdtree_shape(n, ctypes.byref(lvl), ctypes.byref(nodes))
    kd_ws = torch.empty(lib.pgnn_kdtree_workspace_bytes(n), dtype=torch.uint8,
                        device=dev)
    kd_idx = torch.empty(n, dtype=torch.int32, device=dev)
    kd_bounds = torch.empty((nodes.value, 6), dtype=torch.float64, device=dev)
    kd_status = torch.zeros(1, dtype=torch.int32, device=dev)

    def part_kd():  # noqa: F811
        _lib.check(lib.pgnn_kdtree_replica(
            _lib.ptr(x0), n, _lib.ptr(kd_ws), kd_ws.numel(), _lib.ptr(kd_idx),
            _lib.ptr(kd_bounds), _lib.ptr(kd_status), _lib.stream_ptr()), "kd")

    def part_keypoints():
        G.keypoints_device(x0, voxel, 'center', num_out=counts[0:2], k_hint=3000)

    def part_radius():
        for l, c in enumerate(lv):
            G.radius_graph_dyn_device(
                coords0[c['graph_level']], coords0[c['graph_level'] + 1],
                c['graph_gen_kwargs']['radius'], None, 1 << 21,
                counts[2 + 2 * l:4 + 2 * l], 0)

    def part_all():
        eng.build_graph_deferred(x0)

    n_frames = 48
    for name, part in (("nothing", part_none), ("kd-tree replica", part_kd),
                       ("keypoints (voxel hash + kd + 1-NN)", part_keypoints),
                       ("two radius graphs", part_radius),
                       ("whole build", part_all), ("nothing", part_none)):
        for rep in range(2):
            cur = torch.cuda.current_stream()
            for s in (s1, s2, sb):
                s.wait_stream(cur)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_frames):
                _, f, g = pool[(i + 5) % 8]
                with torch.cuda.stream(sb):
                    part()
                with torch.cuda.stream((s1, s2)[i % 2]):
                    eng.model.predict(f, *g, is_training=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_frames * 1e3
        print("GNN on 2 streams + per frame %-36s: %.3f ms/frame" % (name, dt))


if __name__ == "__main__":
    main()
