"""The SECONDARY split-bf16 arithmetic of the per-edge layer
(csrc/edge_ws_bf16.h, model.edge_arith = 'bf16x3'): both operands of the
300x300 / 256x256 product are split exactly into three bf16 parts and the six
products of combined order <= 2 accumulate in fp32 on the bf16 matrix pipe.

It is NOT bit-identical to the fp32-MFMA kernel (another summation), so the
bars are: (i) within fp32 rounding noise of the fp32 kernel, (ii) no further
from a float64 evaluation than the fp32 kernel is -- at the layer and for whole
BASELINE-size frames (both distances are printed), (iii) every other property
of the stage (foreign ids, unsorted lists, ragged counts, capacity form)
unchanged.  The fp32 kernel stays the default and the parity reference."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs, weights
from pointgnn_amd.synthetic import synthetic_cloud
from oracle import gnn_oracle as gn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_TOL = 2e-4


def gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from pointgnn_amd import _lib
    _lib.load()
    return torch.device("cuda")


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _stage(dev, c, edges, k, seed, arith, unsorted=False):
    """One edge stage (gather, layer, scatter-max) through the C ABI with the
    given arithmetic; returns (out [k, c], inputs for the float64 reference)."""
    import torch
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    wq = gnn.padded_width(c)
    p = np.zeros((k, wq), np.float32)
    q = np.zeros((k, wq), np.float32)
    p[:, :c] = rng.standard_normal((k, c))
    q[:, :c] = 0.3 * rng.standard_normal((k, c))
    w = (rng.standard_normal((c, c)) / np.sqrt(c)).astype(np.float32)
    b = (0.1 * rng.standard_normal(c)).astype(np.float32)
    store = gnn.ParamStore({}, device=dev)
    chain = gnn.Chain(store, [(w, b, 0)])
    pd, qd, ed = T(p, dev), T(q, dev), T(edges, dev)
    out = torch.empty((k, wq), dtype=torch.float32, device=dev)
    flag = 0 if unsorted else 1
    if arith == "f32":
        _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd(
            _lib.ptr(pd), _lib.ptr(qd), wq, c, _lib.ptr(ed), len(edges), k,
            chain.array, 1, flag, _lib.ptr(out), wq,
            _lib.ptr(_lib.sched_ws()), _lib.stream_ptr()), "f32 edge stage")
    else:
        host = np.empty(lib.pgnn_packed_fc_bf16x3_bytes(c, c), np.uint8)
        _lib.check(lib.pgnn_pack_fc_bf16x3(w.ctypes.data, b.ctypes.data, c, c,
                                           host.ctypes.data))
        image = T(host, dev)
        _lib.check(lib.pgnn_edge_mlp_scatter_max_bf16x3_fwd(
            _lib.ptr(pd), _lib.ptr(qd), wq, c, _lib.ptr(ed), len(edges), k,
            _lib.ptr(image), c, 0, flag, _lib.ptr(out), wq, None, None,
            _lib.stream_ptr()), "bf16x3 edge stage")
    return out.cpu().numpy()[:, :c], (p, q, w, b)


def _stage_f64(p, q, w, b, edges, k, c):
    src, dst = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    ok = (dst >= 0) & (dst < k)
    out = np.full((k, c), np.finfo(np.float32).min, np.float64)
    w64, b64 = w.astype(np.float64), b.astype(np.float64)
    for lo in range(0, len(edges), 65536):
        s, d, m = src[lo:lo + 65536], dst[lo:lo + 65536], ok[lo:lo + 65536]
        h1 = np.maximum(p[s, :c] - q[np.where(m, d, 0), :c], 0)   # fp32, as the kernel
        rows = np.maximum(h1.astype(np.float64) @ w64 + b64, 0)
        np.maximum.at(out, d[m], rows[m])
    return out


@pytest.mark.parametrize("c", [300, 256])
def test_bf16x3_edge_stage_is_as_close_to_float64_as_the_fp32_kernel(dev, c):
    g = gold("graph_small.npz")
    edges = g["ref_edges1"].astype(np.int32)
    k = g["kp_xyz"].shape[0]
    assert len(edges) >= 70000
    f32, ins = _stage(dev, c, edges, k, c, "f32")
    b16, _ = _stage(dev, c, edges, k, c, "bf16x3")
    ref = _stage_f64(*ins, edges, k, c)
    scale = np.abs(ref).max()
    e32, e16 = np.abs(f32 - ref).max(), np.abs(b16 - ref).max()
    print("C %d E %d: |out|max %.3g; max error vs float64: fp32-MFMA %.3g, "
          "bf16x3 %.3g; bf16x3 vs fp32-MFMA %.3g" % (
              c, len(edges), scale, e32, e16, np.abs(b16 - f32).max()))
    assert e16 <= 1.25 * e32 + 1e-7 * scale
    np.testing.assert_allclose(b16, f32, atol=2e-6 * scale, rtol=0)


def test_bf16x3_edge_stage_edge_cases(dev):
    """Unsorted list (all-atomic flush), foreign / negative dst ids, a ragged
    edge count, empty segments: the same answers as the fp32 kernel."""
    g = gold("graph_small.npz")
    edges = g["ref_edges1"].astype(np.int32)
    k = g["kp_xyz"].shape[0]
    rng = np.random.default_rng(3)
    cases = {}
    e = edges[:len(edges) - 7].copy()            # ragged: not a multiple of 16
    cases["ragged"] = (e, False)
    e = edges.copy()
    e[rng.choice(len(e), 500, replace=False), 1] = k + 5     # foreign ids
    e[rng.choice(len(e), 500, replace=False), 1] = -1
    cases["foreign"] = (e, True)
    cases["shuffled"] = (edges[rng.permutation(len(edges))], True)
    e = edges[edges[:, 1] % 3 != 0]               # a third of the segments empty
    cases["empty_segments"] = (e, False)
    for name, (e, unsorted) in cases.items():
        if len(e) < 70000:
            continue
        f32, _ = _stage(dev, 300, e, k, 5, "f32", unsorted)
        b16, _ = _stage(dev, 300, e, k, 5, "bf16x3", unsorted)
        lowest = np.finfo(np.float32).min
        assert np.array_equal(f32 == lowest, b16 == lowest), name
        m = f32 != lowest
        scale = np.abs(f32[m]).max()
        assert np.abs(f32[m] - b16[m]).max() <= 2e-6 * scale, name


@pytest.mark.parametrize("name,preset", [("car_auto_T3", "car"),
                                         ("car_auto_T3", "car_600k"),
                                         ("ped_cyl_auto_T3", "ped_dense")])
def test_bf16x3_full_size_logits_vs_float64_oracle(dev, name, preset):
    """BASELINE configs 3 and 5 at full size with the split-bf16 edge stage:
    logits / box encodings against the float64 oracle, beside the fp32-MFMA
    path's distance on the same frame."""
    import torch
    from pointgnn_amd import gnn, graph_gen, models
    cfg = configs.get_config(name)
    xyz, inten = synthetic_cloud(seed=0, preset=preset)
    params = weights.init_params(cfg, seed=0, bias_scale=0.05)
    fn = graph_gen.get_graph_generate_fn(cfg["graph_gen_method"])
    coords, kps, edges = fn(T(xyz, dev), **cfg["runtime_graph_gen_kwargs"])
    model = models.get_model(cfg["model_name"])(
        num_classes=cfg["num_classes"], box_encoding_len=7, mode="test",
        **cfg["model_kwargs"]).load_state_dict(params)
    f = T(inten, dev)
    out = {}
    for arith in ("f32", "bf16x3"):
        model.edge_arith = arith
        lg, bx = model.predict(f, coords, kps, edges, False)
        out[arith] = (lg.cpu().numpy(), bx.cpu().numpy())
    assert not np.array_equal(out["f32"][0], out["bf16x3"][0]), \
        "the bf16x3 kernel did not run"
    c_np = [c.cpu().numpy() for c in coords]
    k_np = [k.cpu().numpy() for k in kps]
    e_np = [e.cpu().numpy() for e in edges]
    from conftest import fullsize_oracle
    lg, bx, _ = fullsize_oracle((name, preset), params, cfg, inten, c_np, k_np,
                                e_np)
    d32 = (np.abs(out["f32"][0] - lg).max(), np.abs(out["f32"][1] - bx).max())
    d16 = (np.abs(out["bf16x3"][0] - lg).max(),
           np.abs(out["bf16x3"][1] - bx).max())
    print("%s/%s K %d E1 %d: max|dlogit| / max|dbox| vs float64: fp32-MFMA "
          "%.3g / %.3g, bf16x3 %.3g / %.3g; bf16x3 vs fp32-MFMA %.3g" % (
              name, preset, len(c_np[1]), len(e_np[1]), d32[0], d32[1], d16[0],
              d16[1], np.abs(out["bf16x3"][0] - out["f32"][0]).max()))
    np.testing.assert_allclose(out["bf16x3"][0], lg, atol=FP_TOL, rtol=1e-4)
    np.testing.assert_allclose(out["bf16x3"][1], bx, atol=FP_TOL, rtol=1e-4)
    assert d16[0] <= 1.5 * d32[0] + 2e-7 and d16[1] <= 1.5 * d32[1] + 2e-7
