"""The SECONDARY arithmetics of the per-edge layer on the matrix pipe's 16-bit
formats: 'bf16x3' (csrc/edge_ws_bf16.h: both operands of the 300x300 / 256x256
product split exactly into three bf16 parts, the six products of combined order
<= 2 accumulated in fp32) and 'f16x2' (csrc/edge_ws_f16.h: both operands as two
fp16 values, 22 significand bits, three products; with it the wide last layer
of PointSetPooling's point MLP runs in the same representation,
csrc/pool_ws_f16.h).

Neither is bit-identical to the fp32-MFMA kernel (another summation), so the
bars are: (i) within fp32 rounding noise of the fp32 kernel, (ii) no further
from a float64 evaluation than the fp32 kernel is (x1.25 at the layer, x1.5 on
whole frames) -- at the layer and for whole
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
        host = np.empty(getattr(lib, "pgnn_packed_fc_%s_bytes" % arith)(c, c),
                        np.uint8)
        _lib.check(getattr(lib, "pgnn_pack_fc_%s" % arith)(
            w.ctypes.data, b.ctypes.data, c, c, host.ctypes.data))
        image = T(host, dev)
        head = (_lib.ptr(pd), _lib.ptr(qd), wq, c, _lib.ptr(ed), len(edges), k,
                _lib.ptr(image), c, 0, flag, _lib.ptr(out), wq)
        if arith == "bf16x3":
            _lib.check(lib.pgnn_edge_mlp_scatter_max_bf16x3_fwd(
                *head, None, None, _lib.stream_ptr()), "bf16x3 edge stage")
        else:
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.pgnn_edge_mlp_scatter_max_f16x2_fwd(
                *head, _lib.ptr(status), None, None, _lib.stream_ptr()),
                "f16x2 edge stage")
            assert int(status.item()) == 0, "range flag raised"
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


ARITHS = ["bf16x3", "f16x2"]


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("c", [300, 256])
def test_bf16x3_edge_stage_is_as_close_to_float64_as_the_fp32_kernel(dev, c,
                                                                     arith):
    g = gold("graph_small.npz")
    edges = g["ref_edges1"].astype(np.int32)
    k = g["kp_xyz"].shape[0]
    assert len(edges) >= 70000
    f32, ins = _stage(dev, c, edges, k, c, "f32")
    b16, _ = _stage(dev, c, edges, k, c, arith)
    ref = _stage_f64(*ins, edges, k, c)
    scale = np.abs(ref).max()
    e32, e16 = np.abs(f32 - ref).max(), np.abs(b16 - ref).max()
    print("C %d E %d: |out|max %.3g; max error vs float64: fp32-MFMA %.3g, "
          "%s %.3g; %s vs fp32-MFMA %.3g" % (
              c, len(edges), scale, e32, arith, e16, arith,
              np.abs(b16 - f32).max()))
    assert e16 <= 1.25 * e32 + 1e-7 * scale
    np.testing.assert_allclose(b16, f32, atol=2e-6 * scale, rtol=0)


@pytest.mark.parametrize("arith", ARITHS)
def test_bf16x3_edge_stage_edge_cases(dev, arith):
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
        b16, _ = _stage(dev, 300, e, k, 5, arith, unsorted)
        lowest = np.finfo(np.float32).min
        assert np.array_equal(f32 == lowest, b16 == lowest), name
        m = f32 != lowest
        scale = np.abs(f32[m]).max()
        assert np.abs(f32[m] - b16[m]).max() <= 2e-6 * scale, name


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("name,preset", [("car_auto_T3", "car"),
                                         ("car_auto_T3", "car_600k"),
                                         ("ped_cyl_auto_T3", "ped_dense")])
def test_bf16x3_full_size_logits_vs_float64_oracle(dev, name, preset, arith):
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
    for ar in ("f32", arith):
        model.edge_arith = ar
        lg, bx = model.predict(f, coords, kps, edges, False)
        out["f32" if ar == "f32" else "bf16x3"] = (lg.cpu().numpy(),
                                                   bx.cpu().numpy())
    assert model.edge_range_ok()
    assert not np.array_equal(out["f32"][0], out["bf16x3"][0]), \
        "the %s kernel did not run" % arith
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
          "%.3g / %.3g, %s %.3g / %.3g; %s vs fp32-MFMA %.3g" % (
              name, preset, len(c_np[1]), len(e_np[1]), d32[0], d32[1], arith,
              d16[0], d16[1], arith,
              np.abs(out["bf16x3"][0] - out["f32"][0]).max()))
    np.testing.assert_allclose(out["bf16x3"][0], lg, atol=FP_TOL, rtol=1e-4)
    np.testing.assert_allclose(out["bf16x3"][1], bx, atol=FP_TOL, rtol=1e-4)
    assert d16[0] <= 1.5 * d32[0] + 2e-7 and d16[1] <= 1.5 * d32[1] + 2e-7


def _pool_stage(dev, g, edges, arith, seed=0, unsorted=False, big=None):
    """PointSetPooling's fused stage (gather, point MLP 4-32-64-128-300,
    scatter-max) through the C ABI; returns (out [K, 300], weights, status)."""
    import torch
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    widths = [4, 32, 64, 128, 300]
    layers = []
    for a, b_ in zip(widths[:-1], widths[1:]):
        w = (rng.standard_normal((a, b_)) * np.sqrt(2.0 / a)).astype(np.float32)
        layers.append((w, (0.1 * rng.standard_normal(b_)).astype(np.float32), 0))
    if big is not None:     # blow one hidden activation up (a 32 -> 64 column)
        layers[1][0][:, 5] *= big
    store = gnn.ParamStore({}, device=dev)
    chain = gnn.Chain(store, layers)
    k = int(g["kp_idx"].shape[0])
    feat, xyz = T(g["intensity"], dev), T(g["xyz"], dev)
    kp, ed = T(g["kp_idx"].reshape(-1), dev), T(edges, dev)
    out = torch.empty((k, 304), dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    head = (_lib.ptr(feat), 1, _lib.ptr(xyz), _lib.ptr(kp), _lib.ptr(ed),
            len(edges), k, chain.array, 4)
    tail = (0 if unsorted else 1, _lib.ptr(out), 304, _lib.ptr(_lib.sched_ws()))
    if arith == "f32":
        _lib.check(lib.pgnn_point_set_pooling_fwd(
            *head, *tail, _lib.stream_ptr()), "f32 pooling stage")
    else:
        w, b_ = layers[-1][0], layers[-1][1]
        host = np.empty(lib.pgnn_packed_fc_f16x2_bytes(128, 300), np.uint8)
        _lib.check(lib.pgnn_pack_fc_f16x2_acc(
            w.ctypes.data, b_.ctypes.data, 128, 300, host.ctypes.data))
        image = T(host, dev)
        hidden = None
        if arith == "f16x2":      # ("f16x2_last": the 64 -> 128 layer in fp32)
            w, b_ = layers[-2][0], layers[-2][1]
            host = np.empty(lib.pgnn_packed_fc_f16x2_bytes(64, 128), np.uint8)
            _lib.check(lib.pgnn_pack_fc_f16x2_acc(
                w.ctypes.data, b_.ctypes.data, 64, 128, host.ctypes.data))
            hidden = T(host, dev)
        _lib.check(lib.pgnn_point_set_pooling_f16x2_fwd(
            *head, _lib.ptr(image), _lib.ptr(hidden), *tail, _lib.ptr(status),
            None, None, _lib.stream_ptr()), "f16x2 pooling stage")
    return out.cpu().numpy()[:, :300], layers, int(status.item())


def _pool_f64(g, edges, layers):
    src, dst = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    k = int(g["kp_idx"].shape[0])
    ok = (dst >= 0) & (dst < k)
    xyz, kp = g["xyz"], g["kp_idx"].reshape(-1)
    d = np.where(ok, dst, 0)
    x = np.concatenate([g["intensity"][src], xyz[src] - xyz[kp[d]]],
                       axis=1).astype(np.float64)        # fp32 inputs
    for w, b_, _ in layers:
        x = np.maximum(x @ w.astype(np.float64) + b_.astype(np.float64), 0)
    out = np.full((k, 300), np.finfo(np.float32).min, np.float64)
    np.maximum.at(out, dst[ok], x[ok])
    return out


def test_f16x2_pooling_stage_vs_fp32_kernel_and_float64(dev):
    """The f16x2 form of the pooling stage (last layer in two fp16 parts per
    operand, hidden layers fp32 as before) against the fp32 kernel and a
    float64 evaluation, on the reference's own level-0 edge list."""
    g = gold("graph_small.npz")
    edges = g["ref_edges0"].astype(np.int32)
    assert len(edges) >= 70000
    f32, layers, _ = _pool_stage(dev, g, edges, "f32")
    ref = _pool_f64(g, edges, layers)
    lowest = np.finfo(np.float32).min
    m = f32 != lowest
    scale = np.abs(ref[m]).max()
    e32 = np.abs(f32 - ref)[m].max()
    for arith in ("f16x2_last", "f16x2"):   # last layer only / 64->128 as well
        f16, _, status = _pool_stage(dev, g, edges, arith)
        assert status == 0
        assert not np.array_equal(f32, f16), "the f16x2 kernel did not run"
        assert np.array_equal(f32 == lowest, f16 == lowest)
        e16 = np.abs(f16 - ref)[m].max()
        print("pooling E %d: |out|max %.3g; max error vs float64: fp32-MFMA "
              "%.3g, %s %.3g; %s vs fp32-MFMA %.3g" % (
                  len(edges), scale, e32, arith, e16, arith,
                  np.abs(f16 - f32)[m].max()))
        assert e16 <= 1.5 * e32 + 1e-7 * scale
        np.testing.assert_allclose(f16[m], f32[m], atol=2e-6 * scale, rtol=0)


def test_f16x2_pooling_stage_edge_cases(dev):
    """Ragged edge count, foreign / negative set ids, a shuffled list, empty
    sets: the answers of the fp32 kernel; a hidden activation beyond fp16's
    range raises the flag; below the size threshold the entry declines."""
    import torch
    from pointgnn_amd import _lib
    lib = _lib.load()
    g = gold("graph_small.npz")
    edges = g["ref_edges0"].astype(np.int32)
    k = int(g["kp_idx"].shape[0])
    rng = np.random.default_rng(4)
    lowest = np.finfo(np.float32).min
    cases = {"ragged": (edges[:len(edges) - 5].copy(), False)}
    e = edges.copy()
    e[rng.choice(len(e), 400, replace=False), 1] = k + 3
    e[rng.choice(len(e), 400, replace=False), 1] = -1
    cases["foreign"] = (e, True)
    cases["shuffled"] = (edges[rng.permutation(len(edges))], True)
    cases["empty_sets"] = (edges[edges[:, 1] % 7 != 0], False)
    for name, (e, unsorted) in cases.items():
        f32, _, _ = _pool_stage(dev, g, e, "f32", unsorted=unsorted)
        f16, _, st = _pool_stage(dev, g, e, "f16x2", unsorted=unsorted)
        assert st == 0, name
        assert np.array_equal(f32 == lowest, f16 == lowest), name
        m = f32 != lowest
        assert np.abs(f32[m] - f16[m]).max() <= 2e-6 * np.abs(f32[m]).max(), name
    _, _, st = _pool_stage(dev, g, edges, "f16x2", big=3e5)
    assert st == 1
    # few edges: the entry declines (the caller takes the fp32 entry) unless
    # the size test is lifted
    few = edges[:3000]
    _lib.set_tunable("b16_force", 0)
    with pytest.raises(_lib.PointGnnHipError, match="code -3"):
        _pool_stage(dev, g, few, "f16x2")
    _lib.set_tunable("b16_force", 1)
    try:
        f32, _, _ = _pool_stage(dev, g, few, "f32")
        f16, _, _ = _pool_stage(dev, g, few, "f16x2")
    finally:
        _lib.set_tunable("b16_force", 0)
    m = f32 != lowest
    assert np.array_equal(m, f16 != lowest)
    assert np.abs(f32[m] - f16[m]).max() <= 2e-6 * np.abs(f32[m]).max()


def test_f16x2_flags_activations_out_of_range(dev):
    """fp16 ends at 65504: the f16x2 kernel clamps there and raises the range
    flag when a gathered activation reached 32768; model.edge_range_ok() reads
    and clears it."""
    import torch
    from pointgnn_amd import _lib, gnn
    lib = _lib.load()
    g = gold("graph_small.npz")
    edges = g["ref_edges1"].astype(np.int32)
    k = g["kp_xyz"].shape[0]
    c = 300
    wq = gnn.padded_width(c)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((c, c)) / np.sqrt(c)).astype(np.float32)
    b = np.zeros(c, np.float32)
    host = np.empty(lib.pgnn_packed_fc_f16x2_bytes(c, c), np.uint8)
    _lib.check(lib.pgnn_pack_fc_f16x2(w.ctypes.data, b.ctypes.data, c, c,
                                      host.ctypes.data))
    image = T(host, dev)
    for big, want in ((100.0, 0), (40000.0, 1)):
        p = np.zeros((k, wq), np.float32)
        p[:, :c] = rng.standard_normal((k, c))
        p[7, 5] = big
        q = np.zeros((k, wq), np.float32)
        out = torch.empty((k, wq), dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        pd, qd, ed = T(p, dev), T(q, dev), T(edges, dev)   # (kept alive)
        _lib.check(lib.pgnn_edge_mlp_scatter_max_f16x2_fwd(
            _lib.ptr(pd), _lib.ptr(qd), wq, c, _lib.ptr(ed), len(edges), k,
            _lib.ptr(image), c, 0, 1, _lib.ptr(out), wq, _lib.ptr(status),
            None, None, _lib.stream_ptr()), "f16x2 edge stage")
        assert int(status.item()) == want, (big, int(status.item()))
        assert torch.isfinite(out[:, :c]).all()
    # a weight outside fp16's range is refused when the image is built
    w[3, 4] = 70000.0
    rc = lib.pgnn_pack_fc_f16x2(w.ctypes.data, b.ctypes.data, c, c,
                                host.ctypes.data)
    assert rc == _lib.E_UNSUPPORTED


def test_engine_reruns_a_flagged_f16x2_batch_in_fp32(dev):
    """A batch in which the 'f16x2' range guard trips (the batch
    entry points raised and threw every output away): the engine runs the
    batch again with the fp32 edge stage and returns those results -- the
    fp32 engine's, bit for bit -- instead."""
    import torch
    from pointgnn_amd import configs, weights
    from pointgnn_amd.engine import InferenceEngine
    from pointgnn_amd.synthetic import synthetic_cloud
    cfg = configs.car_auto_config(1)
    params = weights.init_params(cfg, seed=4, bias_scale=0.05)
    name = "layer1/combined_features/fully_connected_1/biases"
    params[name] = params[name] + np.float32(40000.0)   # h ~ 4e4 > 16384
    frames = []
    for s in range(4):
        xyz, inten = synthetic_cloud(seed=s, preset="small")
        frames.append((T(xyz, dev), T(inten, dev)))
    ref = InferenceEngine(cfg, params, device=dev, edge_arith="f32")
    want = ref.run_frames_on_streams(frames, 2)
    eng = InferenceEngine(cfg, params, device=dev, edge_arith="f16x2")
    got = eng.run_frames_on_streams(frames, 2)
    assert eng.f16x2_batch_reruns == 1 and eng.model.edge_arith == "f16x2"
    assert len(eng.frame_shapes) == len(frames)
    pip = eng.run_frames_pipelined(frames, compute_streams=2, deferred=True,
                                   graph_streams=2)
    torch.cuda.synchronize()
    assert eng.f16x2_batch_reruns == 2
    for (l0, b0), (l1, b1), (l2, b2) in zip(want, got, pip):
        assert torch.isfinite(l0).all()
        assert torch.equal(l0, l1) and torch.equal(b0, b1)
        assert torch.equal(l0, l2) and torch.equal(b0, b2)
    # in range: no rerun
    ok = InferenceEngine(cfg, weights.init_params(cfg, seed=4, bias_scale=0.05),
                         device=dev, edge_arith="f16x2")
    ok.run_frames_on_streams(frames, 2)
    assert getattr(ok, "f16x2_batch_reruns", 0) == 0
