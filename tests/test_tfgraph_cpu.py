"""Pins the oracles to the REFERENCE'S OWN TensorFlow graphs.

`tests/golden/tfgraph_*.npz` hold what `checkpoints/*/model-*.meta` (the
MetaGraphDef train.py:178-405 built and tf.train.Saver serialized) evaluates to
under oracle/tf_meta_interp.py -- logits, box encodings, the re-weighted
cross-tower losses, the tower-mean gradient of every variable, the learning
rate and the streaming tf.metrics values (tests/golden/make_golden_tfgraph.py).
Here:
  * the interpreter itself is unit-tested on hand-built protobufs;
  * oracle/gnn_oracle.predict must reproduce the graph's logits/boxes;
  * oracle/train_oracle (float64 autograd) must reproduce losses + gradients;
  * oracle/metrics_oracle must reproduce the tf.metrics values;
  * with /root/reference present the graphs are re-evaluated live.
No GPU."""
import glob
import os
import struct

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from oracle import gnn_oracle as gn
from oracle import metrics_oracle as mo
from oracle import tf_meta_interp as ti
from oracle import train_oracle as to
from golden.make_golden_tfgraph import (CHECKPOINTS, fixture_weights,
                                        graph_inputs, sample_positions,
                                        tower_batch)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = sorted(CHECKPOINTS)


def gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


# ------------------------------------------------------- protobuf + kernels
def _vi(x):
    out = b""
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _ld(field, payload):
    return _vi(field << 3 | 2) + _vi(len(payload)) + payload


def _attr(key, value):
    return _ld(5, _ld(1, key.encode()) + _ld(2, value))


def _node(name, op, inputs=(), attrs=()):
    b = _ld(1, name.encode()) + _ld(2, op.encode())
    for i in inputs:
        b += _ld(3, i.encode())
    for k, v in attrs:
        b += _attr(k, v)
    return ti.Node(memoryview(b))


def _const(name, arr):
    arr = np.asarray(arr)
    code = {np.dtype(np.float32): 1, np.dtype(np.int32): 3,
            np.dtype(np.bool_): 10}[arr.dtype]
    shape = b"".join(_ld(2, _vi(1 << 3) + _vi(d)) for d in arr.shape)
    tensor = _vi(1 << 3) + _vi(code) + _ld(2, shape) + _ld(4, arr.tobytes())
    return _node(name, "Const", attrs=[("value", _ld(8, tensor)),
                                       ("dtype", _vi(6 << 3) + _vi(code))])


def _i(v):
    return _vi(3 << 3) + _vi(v & (2 ** 64 - 1))


def test_interpreter_wire_format_and_kernels():
    x = np.arange(24, dtype=np.float32).reshape(4, 6)
    f32 = _vi(6 << 3) + _vi(1)
    nodes = [
        _node("x", "Placeholder", attrs=[("dtype", f32)]),
        _const("b", np.array([1, 0], np.int32)),
        _const("e", np.array([3, 4], np.int32)),
        _const("s", np.array([1, 2], np.int32)),
        # x[1:3, ::2] with end_mask on axis 1 -> begin [1,0] end [3,*] stride [1,2]
        _node("ss", "StridedSlice", ["x", "b", "e", "s"],
              [("end_mask", _i(2)), ("begin_mask", _i(0)),
               ("shrink_axis_mask", _i(0))]),
        # x[2] via shrink_axis_mask
        _const("b2", np.array([2], np.int32)),
        _const("e2", np.array([3], np.int32)),
        _const("s2", np.array([1], np.int32)),
        _node("row", "StridedSlice", ["x", "b2", "e2", "s2"],
              [("shrink_axis_mask", _i(1))]),
        _const("ids", np.array([2, 0, 2, -1], np.int32)),
        _const("num", np.array(4, np.int32)),
        _node("smax", "UnsortedSegmentMax", ["x", "ids", "num"]),
        _node("ssum", "UnsortedSegmentSum", ["x", "ids", "num"]),
        _const("pred", np.array(True)),
        _node("sw", "Switch", ["x", "pred"]),
        _node("neg", "Neg", ["sw:0"]),          # dead branch
        _node("pos", "Identity", ["sw:1"]),
        _node("m", "Merge", ["neg", "pos"]),
        _node("ctl", "Identity", ["x", "^m"]),
    ]
    g = ti.Graph(nodes)
    ss, row, smax, ssum, m, ctl = g.run(
        ["ss", "row", "smax", "ssum", "m", "ctl"], {"x": x})
    assert np.array_equal(ss, x[1:3, ::2])
    assert np.array_equal(row, x[2])
    low = np.finfo(np.float32).min
    assert np.array_equal(smax[0], x[1]) and np.all(smax[1] == low)
    assert np.array_equal(smax[2], np.maximum(x[0], x[2]))
    assert np.all(smax[3] == low)               # id -1 dropped
    assert np.array_equal(ssum[2], x[0] + x[2]) and np.all(ssum[3] == 0)
    assert np.array_equal(m, x) and np.array_equal(ctl, x)
    with pytest.raises(KeyError):
        g.run("ss")                             # placeholder not fed


def test_interpreter_variables_and_stateful_ops():
    f32 = _vi(6 << 3) + _vi(1)
    shape = _ld(7, _ld(2, _vi(1 << 3) + _vi(2)))
    nodes = [
        _node("v", "VariableV2", attrs=[("dtype", f32), ("shape", shape)]),
        _const("lr", np.array(0.5, np.float32)),
        _const("g", np.array([2.0, -4.0], np.float32)),
        _node("apply", "ApplyGradientDescent", ["v", "lr", "g"]),
        _node("after", "Identity", ["v", "^apply"]),
    ]
    g = ti.Graph(nodes)
    with pytest.raises(KeyError):
        g.run("v")
    g.set_variables({"v": np.array([1.0, 1.0], np.float32)})
    g.run(["apply"])
    assert np.array_equal(g.variables["v"], [0.0, 3.0])


# ----------------------------------------------------- oracle vs graph golden
@pytest.mark.parametrize("name", ALL)
def test_gnn_oracle_matches_reference_tf_graph(name):
    """oracle/gnn_oracle.predict == the reference's serialized graph (tower 0's
    t_logits / t_pred_box / t_probs), float32."""
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, kind = fixture_weights(name, cfg)
    assert kind == str(t["weights_kind"])
    kw = graph_inputs("graph_tiny.npz")
    lg, bx = gn.predict(w, cfg, kw["features"], kw["coords"], kw["keypoints"],
                        kw["edges"], dtype=np.float32)
    assert lg.shape == t["logits"].shape and bx.shape == t["box_encodings"].shape
    # same NumPy kernels in the same order: differences are BLAS blocking only
    np.testing.assert_allclose(lg, t["logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(bx, t["box_encodings"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(gn.softmax(lg), t["probs"], atol=2e-6, rtol=0)
    lg64, bx64 = gn.predict(w, cfg, kw["features"], kw["coords"],
                            kw["keypoints"], kw["edges"], dtype=np.float64)
    assert np.abs(lg64 - t["logits"]).max() < 2e-4
    assert np.abs(bx64 - t["box_encodings"]).max() < 2e-4


def _oracle_batches(cfg, n):
    out = []
    for t in range(n):
        kw = tower_batch(cfg, t)
        out.append((kw["features"], kw["coords"], kw["keypoints"], kw["edges"],
                    kw["labels"], kw["gt_boxes"], kw["valid"]))
    return out


@pytest.mark.parametrize("name", ["car_auto_T0", "car_auto_T3",
                                  "ped_cyl_auto_T3"])
def test_train_oracle_matches_reference_tf_graph(name):
    """Losses (after unify_copies) and the tower-mean gradient of every
    variable, reference graph (float32) vs oracle/train_oracle (float64)."""
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    loss, g_data, g_reg = to.step_gradients(
        w, cfg, _oracle_batches(cfg, int(t["num_towers"])))
    for k in ("cls_loss", "loc_loss", "reg_loss"):
        assert abs(loss[k] - float(t[k])) < 2e-5 * max(1.0, abs(loss[k])), k
    assert abs(loss["cls_loss"] + loss["loc_loss"] + loss["reg_loss"] -
               float(t["total_loss"])) < 1e-4
    worst = 0.0
    for v in g_data:
        ref = (g_data[v] + g_reg[v]).reshape(-1)
        got = t["grad/" + v].astype(np.float64)
        pos = sample_positions(v, ref.size)
        # float32 graph vs float64 autograd: ReLU-kink / arg-max flips move
        # single entries (see tests/test_gpu_train.py); the bar is Frobenius
        fro = np.linalg.norm(got - ref[pos]) / (np.linalg.norm(ref[pos]) + 1e-12)
        nrm = abs(np.linalg.norm(ref) - float(t["gradnorm/" + v])) / \
            (np.linalg.norm(ref) + 1e-12)
        worst = max(worst, fro, nrm)
        assert fro < 5e-3 and nrm < 5e-3, (v, fro, nrm)
    print(name, "worst relative gradient error %.3g" % worst)


@pytest.mark.parametrize("name", ["car_auto_T0", "ped_cyl_auto_T3"])
def test_metrics_oracle_matches_reference_tf_graph(name):
    """tf.metrics.recall / precision / auc(PR, careful_interpolation) / mean as
    the reference's graph evaluates them over two streaming steps
    (train.py:301-373: labels, predictions and probs of tower 0)."""
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    nt = int(t["num_towers"])
    m = mo.StreamingMetricsOracle(cfg["num_classes"])
    for step, order in ((1, list(range(nt))), (2, list(range(nt))[::-1])):
        kw = tower_batch(cfg, order[0])          # what tower 0 is fed
        lg, _ = gn.predict(w, cfg, kw["features"], kw["coords"],
                           kw["keypoints"], kw["edges"], dtype=np.float32)
        m.update(gn.softmax(lg), kw["labels"])
        for c in range(cfg["num_classes"]):
            assert abs(m.recall(c) - float(t["metric%d/recall_%d" % (step, c)])) < 1e-6
            assert abs(m.precision(c) -
                       float(t["metric%d/precision_%d" % (step, c)])) < 1e-6
            assert abs(m.pr_auc(c) - float(t["metric%d/mAP_%d" % (step, c)])) < 2e-5, \
                (step, c, m.pr_auc(c), float(t["metric%d/mAP_%d" % (step, c)]))
    # tf.metrics.mean of the cross-tower losses: both steps see the same towers
    for k in ("cls", "loc", "reg", "total"):
        assert abs(float(t["metric2/mean_%s_loss" % k]) -
                   float(t["%s_loss" % k])) < 1e-5 * max(1, float(t["%s_loss" % k]))


def test_sgd_update_of_reference_tf_graph():
    """ApplyGradientDescent on the tower-mean gradient at the decayed rate
    (train.py:375-405): var' = var - lr * grad, lr = 0.125 * 0.1^floor(step/400k)."""
    for name in ("car_auto_T0", "car_auto_T3"):
        t = gold("tfgraph_%s.npz" % name)
        cfg = configs.get_config(name)
        w, _ = fixture_weights(name, cfg)
        tc = configs.get_train_config(CHECKPOINTS[name])
        from pointgnn_amd.train import learning_rate
        lr = learning_rate(tc, int(t["global_step"]))
        assert abs(lr - float(t["learning_rate"])) < 1e-9
        v0 = [k for k in t if k.startswith("updated/")][0][len("updated/"):]
        g = t["grad/" + v0].reshape(w[v0].shape)   # small variable: stored whole
        want = np.asarray(w[v0], np.float32) - np.float32(t["learning_rate"]) * g
        np.testing.assert_allclose(t["updated/" + v0], want, rtol=0, atol=1e-7)


# ------------------------------------------------------------ live reference
@pytest.mark.parametrize("name", ["car_auto_T0", "car_auto_T1"])
def test_reference_graph_live(name):
    """With /root/reference present: re-evaluate the shipped .meta and compare
    with the committed fixture AND with the oracle, bit for bit (same NumPy
    kernels in the same order)."""
    metas = glob.glob("/root/reference/checkpoints/%s/model-*.meta"
                      % CHECKPOINTS[name])
    if not metas:
        pytest.skip("/root/reference not present (GPU box)")
    from oracle import tf_graph_ref as tg
    ref = tg.ReferenceGraph(metas[0])
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    # the graph's variables are exactly the checkpoint's (names and shapes)
    assert {v: tuple(s) for v, s in ref.variable_shapes().items()} == \
        {v: tuple(np.shape(w[v])) for v in ref.variable_names}
    ref.set_weights(w)
    kw = graph_inputs("graph_tiny.npz")
    lg, bx, pr = ref.predict(**kw)
    t = gold("tfgraph_%s.npz" % name)
    np.testing.assert_allclose(lg, t["logits"], atol=2e-5, rtol=0)
    lo, bo = gn.predict(w, cfg, kw["features"], kw["coords"], kw["keypoints"],
                        kw["edges"], dtype=np.float32)
    assert np.array_equal(lg, lo) and np.array_equal(bx, bo)
    # per-layer: every UnsortedSegmentMax output of tower 0 equals the oracle's
    tw = ref.towers[0]
    seg = [n for n in ref.g.order
           if ref.g.nodes[n].op == "UnsortedSegmentMax"
           and not n.startswith("gradients") and "_1/" not in n
           and "_2/" not in n and "_3/" not in n]
    assert len(seg) == len(cfg["model_kwargs"]["layer_configs"]) - 1
    fd = ref.feed(0, is_training=False, **kw)
    vals = ref.g.run(seg, fd)
    _, _, feats = gn.predict(w, cfg, kw["features"], kw["coords"],
                             kw["keypoints"], kw["edges"], dtype=np.float32,
                             return_features=True)
    assert vals[0].shape == (kw["coords"][1].shape[0], 300)
    assert tw["logits"] in ref.g.nodes and len(feats) == len(seg) + 1
