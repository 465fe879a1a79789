#!/usr/bin/env python
"""fp32-MFMA vs split-bf16 edge stage on a bench frame's real per-vertex inputs
(first GNN iteration of seed 0): kernel-only time by per-launch HIP events,
and the two outputs' distance.

    python tools/bf16x3_bench.py [--preset car_600k] [--config car_auto_T3]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, configs, gnn, weights  # noqa: E402
from pointgnn_amd.engine import InferenceEngine  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
import bench  # noqa: E402


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


preset, config = opt("--preset", "car_600k"), opt("--config", "car_auto_T3")
dev = torch.device("cuda", 0)
cfg = configs.get_config(config)
params = weights.init_params(cfg, seed=0, bias_scale=0.05)
eng = InferenceEngine(cfg, params, device=dev)
xyz, inten = synthetic_cloud(seed=0, preset=preset)
x, f = torch.from_numpy(xyz).to(dev), torch.from_numpy(inten).to(dev)
gnn.EDGE_INPUT_TAP = []
eng.run_frame(x, f)
p, q = [t.clone() for t in gnn.EDGE_INPUT_TAP[0]]
gnn.EDGE_INPUT_TAP = None
coords, kps, edges = eng.last_graph
e1 = edges[1]
n_k, n_e = int(coords[1].shape[0]), int(e1.shape[0])
lib = _lib.load()
store = eng.model._store
key = [k for k in store._cache if k[0] == 'edge'][0]
c, p_chain, wx_dev, rest = store._cache[key]
wq = int(wx_dev.shape[1])
w, b = store.mlp(key[1], len(key[2]))[1]
host = np.empty(lib.pgnn_packed_fc_bf16x3_bytes(*w.shape), np.uint8)
_lib.check(lib.pgnn_pack_fc_bf16x3(np.ascontiguousarray(w).ctypes.data,
                                   np.ascontiguousarray(b).ctypes.data,
                                   w.shape[0], w.shape[1], host.ctypes.data))
image = torch.from_numpy(host).to(dev)
lowest = float(np.finfo(np.float32).min)
out = {a: torch.full((n_k, gnn.padded_width(rest.n_out)), lowest, device=dev)
       for a in ("f32", "bf16x3")}


def run_f32():
    _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd(
        _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e1), n_e, n_k,
        rest.array, rest.n, 1 | 2, _lib.ptr(out["f32"]), out["f32"].stride(0),
        _lib.ptr(_lib.sched_ws()), _lib.stream_ptr()))


host2 = np.empty(lib.pgnn_packed_fc_f16x2_bytes(*w.shape), np.uint8)
_lib.check(lib.pgnn_pack_fc_f16x2(np.ascontiguousarray(w).ctypes.data,
                                  np.ascontiguousarray(b).ctypes.data,
                                  w.shape[0], w.shape[1], host2.ctypes.data))
image2 = torch.from_numpy(host2).to(dev)
out["f16x2"] = torch.full((n_k, gnn.padded_width(rest.n_out)), lowest, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)


def run_f16():
    _lib.check(lib.pgnn_edge_mlp_scatter_max_f16x2_fwd(
        _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e1), n_e, n_k,
        _lib.ptr(image2), int(rest.n_out), 0, 1 | 2, _lib.ptr(out["f16x2"]),
        out["f16x2"].stride(0), _lib.ptr(status), None, None, _lib.stream_ptr()))


def run_b16():
    _lib.check(lib.pgnn_edge_mlp_scatter_max_bf16x3_fwd(
        _lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e1), n_e, n_k,
        _lib.ptr(image), int(rest.n_out), 0, 1 | 2, _lib.ptr(out["bf16x3"]),
        out["bf16x3"].stride(0), None, None, _lib.stream_ptr()))


t32 = bench.time_kernel(run_f32, 10, torch)
t16 = bench.time_kernel(run_b16, 10, torch)
flops = 2.0 * rest.k_in * rest.n_out * n_e
a, bb = out["f32"][:, :rest.n_out], out["bf16x3"][:, :rest.n_out]
print("%s/%s E1 %d K %d C %d" % (config, preset, n_e, n_k, rest.n_out))
print("  fp32-MFMA edge kernel %8.1f us  %6.1f TFLOP/s" % (t32 * 1e6, flops / t32 / 1e12))
print("  split-bf16 (6 terms)  %8.1f us  %6.1f TFLOP/s-equivalent (%.2fx)" % (
    t16 * 1e6, flops / t16 / 1e12, t32 / t16))
import hashlib
print("  sha256 of the bf16x3 output: %s" % hashlib.sha256(
    bb.contiguous().cpu().numpy().tobytes()).hexdigest()[:16])
print("  max |bf16x3 - fp32| %.3g (|out|max %.3g)" % (
    float((a - bb).abs().max()), float(a.abs().max())))

t_f16 = bench.time_kernel(run_f16, 10, torch)
cc = out["f16x2"][:, :rest.n_out]
print("  fp16 x2 (3 terms)     %8.1f us  %6.1f TFLOP/s-equivalent (%.2fx); "
      "max |f16x2 - fp32| %.3g; range flag %d" % (
          t_f16 * 1e6, flops / t_f16 / 1e12, t32 / t_f16,
          float((a - cc).abs().max()), int(status.item())))
