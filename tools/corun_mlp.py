#!/usr/bin/env python
"""Would K-row MLP kernels small enough to sit BESIDE the persistent MFMA
kernels pay?  A stand-in (tools/micro/corun_probe.hip, probe_mlp: 190
workgroups x 256 threads, 24 KB LDS, 57 K-group steps of 5 column tiles per
wave = three 300-wide layers) is timed alone and beside the GNN, and the GNN is
timed with 8 such launches per frame running beside it on another stream."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine, concurrent_streams  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "ab", "libprobe.so"))
    lib.probe_mlp_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
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
    w = torch.randn(19 * 19 * 256 + 512, device=dev) * 0.01
    out = torch.zeros(4, device=dev)
    s1, s2, sb = concurrent_streams(3)
    torch.cuda.synchronize()

    def probes(n, stream):
        for _ in range(n):
            lib.probe_mlp_launch(0, 190, 57, w.data_ptr(), out.data_ptr(),
                                 stream.cuda_stream)

    # the stand-in alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probes(4, sb)
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        e0.record()
        probes(16, sb)
        e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 16 * 1e3
    n_frames = 48
    res = {}
    for per_frame in (0, 8, 0, 8):
        for rep in range(2):
            cur = torch.cuda.current_stream()
            for s in (s1, s2, sb):
                s.wait_stream(cur)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pe0 = torch.cuda.Event(enable_timing=True)
            pe1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sb):
                pe0.record()
            for i in range(n_frames):
                f, g = pool[(i + 5) % 8]
                probes(per_frame, sb)
                with torch.cuda.stream((s1, s2)[i % 2]):
                    eng.model.predict(f, *g, is_training=False)
            with torch.cuda.stream(sb):
                pe1.record()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_frames * 1e3
        beside = pe0.elapsed_time(pe1) / max(1, per_frame * n_frames) * 1e3
        print("GNN on 2 streams + %d stand-in launches per frame: %.3f ms/frame"
              "%s" % (per_frame, dt, "" if not per_frame else
                      "   (stand-in: %.1f us alone, %.1f us beside)" % (alone, beside)))


if __name__ == "__main__":
    main()
