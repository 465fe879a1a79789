#!/usr/bin/env python
"""Standalone timing of the per-vertex (K-row) kernels at the bench frame's K:
the separate launches (pgnn_mlp_fwd update chain, pgnn_vertex_pre_edge_fwd) and
the one-launch forms (pgnn_vertex_update_pre_edge_fwd, pgnn_mlp2_fwd), HIP
events round 50 back-to-back launches each.

    [PGNN_LIB=ab/lib<variant>.so] python tools/krow_bench.py [--k 3352] [--c 300]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import _lib, gnn  # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


k, c = arg("--k", 3352), arg("--c", 300)
dev = torch.device("cuda", 0)
lib = _lib.load()
rng = np.random.default_rng(0)
wq = gnn.padded_width(c)
store = gnn.ParamStore({}, dev)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rnd(*shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


upd = gnn.Chain(store, [(rnd(c, c, scale=c ** -0.5), rnd(c, scale=0.1), 0),
                        (rnd(c, c, scale=c ** -0.5), rnd(c, scale=0.1), c)])
off = gnn.Chain(store, [(rnd(c, 64, scale=0.1), rnd(64), 0),
                        (rnd(64, 3, scale=0.1), rnd(3), 3)])
w1 = rnd(c + 3, c, scale=0.1)
p_chain = gnn.Chain(store, [(w1, rnd(c), c)])
heads = gnn.Chain(store, [(rnd(c, 320, scale=c ** -0.5), rnd(320, scale=0.1), 0),
                          (rnd(320, 272, scale=0.05), rnd(272, scale=0.1), 16),
                          (rnd(272, 48, scale=0.05), rnd(48, scale=0.1), 48)])
wx = np.zeros((3, wq), np.float32)
wx[:, :c] = w1[c:]
wx_dev = T(wx)
agg_in = T(np.pad(rnd(k, c), ((0, 0), (0, wq - c))))
h_prev = T(np.pad(rnd(k, c), ((0, 0), (0, wq - c))))
x = T(rng.uniform(-20, 20, (k, 3)).astype(np.float32))
y = torch.empty((k, wq), device=dev)
P = torch.empty((k, wq), device=dev)
Q = torch.empty((k, wq), device=dev)
agg = torch.empty((k, wq), device=dev)
o = torch.empty((k, 48), device=dev)
st = _lib.stream_ptr()
pre = (off.array, off.n, p_chain.array, _lib.ptr(wx_dev), k)


def run_update():
    _lib.check(lib.pgnn_mlp_fwd(_lib.ptr(agg_in), agg_in.stride(0), c, None, 0,
                                0, k, upd.array, upd.n, _lib.ptr(h_prev),
                                h_prev.stride(0), _lib.ptr(y), y.stride(0), st))


def run_pre():
    _lib.check(lib.pgnn_vertex_pre_edge_fwd(
        _lib.ptr(y), y.stride(0), c, _lib.ptr(x), *pre, _lib.ptr(P),
        _lib.ptr(Q), wq, _lib.ptr(agg), wq, st))


def run_heads():
    _lib.check(lib.pgnn_mlp_fwd(_lib.ptr(y), y.stride(0), c, None, 0, 0, k,
                                heads.array, heads.n, None, 0, _lib.ptr(o),
                                o.stride(0), st))


def run_fused():
    _lib.check(lib.pgnn_vertex_update_pre_edge_fwd(
        _lib.ptr(agg_in), agg_in.stride(0), c, upd.array, upd.n,
        _lib.ptr(h_prev), h_prev.stride(0), _lib.ptr(y), y.stride(0), c,
        _lib.ptr(x), *pre, _lib.ptr(P), _lib.ptr(Q), wq, _lib.ptr(agg), wq, st))


def run_mlp2():
    _lib.check(lib.pgnn_mlp2_fwd(
        _lib.ptr(agg_in), agg_in.stride(0), c, upd.array, upd.n,
        _lib.ptr(h_prev), h_prev.stride(0), _lib.ptr(y), y.stride(0), c,
        heads.array, heads.n, _lib.ptr(o), o.stride(0), k, st))


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


flops = {"update": 4 * c * c, "pre_edge": 2 * (c * 64 + 64 * 3 + (c + 3) * c),
         "heads": 2 * (c * 320 + 320 * 272 + 272 * 48)}
print("lib %s  K %d C %d" % (os.environ.get("PGNN_LIB", "(tree)"), k, c))
for name, fn, fl in (
        ("update (pgnn_mlp_fwd)", run_update, flops["update"]),
        ("pre-edge (pgnn_vertex_pre_edge_fwd)", run_pre, flops["pre_edge"]),
        ("heads (pgnn_mlp_fwd)", run_heads, flops["heads"]),
        ("update + pre-edge, one launch", run_fused,
         flops["update"] + flops["pre_edge"]),
        ("update + heads, one launch", run_mlp2,
         flops["update"] + flops["heads"])):
    us = timeit(fn)
    print("  %-40s %7.1f us  %6.1f TFLOP/s (%.2f of fp32-MFMA peak)" % (
        name, us, fl * k / us / 1e6, fl * k / us / 1e6 / 157.3))
