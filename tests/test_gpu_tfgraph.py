"""The HIP path against the REFERENCE'S OWN TensorFlow graphs
(tests/golden/tfgraph_*.npz = `checkpoints/*/model-*.meta` evaluated by
oracle/tf_meta_interp.py; see tests/golden/make_golden_tfgraph.py).

No oracle restatement sits between the kernels and the reference here: logits,
box encodings, losses, gradients and metrics produced through the C-ABI are
compared with what the reference's serialized graph computes on the same
inputs.  Tolerances: forward 2e-4 absolute (north star: 1e-3); losses 1e-4
relative; gradients per variable by Frobenius error over the stored entries
(float32 ReLU-kink / arg-max flips move single entries, see
tests/test_gpu_train.py)."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from golden.make_golden_tfgraph import (CHECKPOINTS, fixture_weights,
                                        graph_inputs, sample_positions,
                                        tower_batch)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = sorted(CHECKPOINTS)
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


@pytest.mark.parametrize("name", ALL)
def test_predict_matches_reference_tf_graph(dev, name, edge_arith):
    """t_logits / t_pred_box / t_probs of tower 0 (train.py:227-230), on the
    fp32-MFMA edge stage and on the split-bf16 one (same bar; the latter's
    distance to the reference graph at most 1.5x the former's)."""
    from pointgnn_amd import models
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    model = models.get_model(cfg["model_name"])(
        num_classes=cfg["num_classes"], box_encoding_len=7, mode="test",
        **cfg["model_kwargs"])
    model.load_state_dict(w)
    kw = graph_inputs("graph_tiny.npz")
    model.edge_arith = edge_arith
    logits, boxes = model.predict(kw["features"], kw["coords"],
                                  kw["keypoints"], kw["edges"],
                                  is_training=False)
    print("%s [%s] max|dlogit| %.3g max|dbox| %.3g vs the reference TF graph" % (
        name, edge_arith, np.abs(logits - t["logits"]).max(),
        np.abs(boxes - t["box_encodings"]).max()))
    if edge_arith != "f32":
        model.edge_arith = "f32"
        l32, b32 = model.predict(kw["features"], kw["coords"], kw["keypoints"],
                                 kw["edges"], is_training=False)
        # 'bf16x3' changes the edge stage of the GNN layers only; 'f16x2'
        # also the wide layers of the pooling stage (every config has one)
        has_16 = edge_arith == "f16x2" or any(
            l["type"] == "scatter_max_graph_auto_center_net"
            for l in cfg["model_kwargs"]["layer_configs"])
        assert np.array_equal(l32, logits) != has_16, \
            "the 16-bit kernel did not run" if has_16 else "T0 differs"
        for got, ref32, ref in ((logits, l32, t["logits"]),
                                (boxes, b32, t["box_encodings"])):
            assert np.abs(got - ref).max() <= \
                1.5 * np.abs(ref32 - ref).max() + 2e-7
    np.testing.assert_allclose(logits, t["logits"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(boxes, t["box_encodings"], atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(model.postprocess(logits), t["probs"],
                               atol=1e-4, rtol=0)


def _batches(cfg, n):
    out = []
    for i in range(n):
        kw = tower_batch(cfg, i)
        out.append((kw["features"], kw["coords"], kw["keypoints"], kw["edges"],
                    kw["labels"], kw["gt_boxes"], kw["valid"]))
    return out


@pytest.mark.parametrize("name", ALL)
def test_train_step_matches_reference_tf_graph(dev, name):
    """One process, the towers' frames merged by batch_data
    (train.py:135-171), against the graph's tower-mean gradient and
    re-weighted cross-tower losses (train.py:264-299, 397-404): the
    unify_copies weights make the tower mean equal the global per-vertex
    mean, which is what the Trainer computes."""
    from pointgnn_amd import train
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    tc = configs.get_train_config(CHECKPOINTS[name])
    tr = train.Trainer(cfg, train_config=tc, params=w, device=dev)
    tr.global_step = int(t["global_step"])
    merged = train.batch_data(_batches(cfg, int(t["num_towers"])))
    out = tr.train_step(merged, apply=False)
    for k in ("cls_loss", "loc_loss", "reg_loss"):
        assert abs(out[k] - float(t[k])) < 1e-4 * max(1.0, abs(float(t[k]))), \
            (k, out[k], float(t[k]))
    assert abs(out["learning_rate"] - float(t["learning_rate"])) < 1e-9
    got = tr.grad_dict()
    scale = float(cfg["model_kwargs"]["regularizer_kwargs"]["scale"])
    worst = 0.0
    for v in tr.offsets:
        g = np.asarray(got[v], np.float64)
        if v.endswith("/weights"):     # reg_loss = scale * sum|W| (models.py:307)
            g = g + scale * np.sign(np.asarray(w[v], np.float64))
        ref = t["grad/" + v].astype(np.float64)
        pos = sample_positions(v, g.size)
        fro = np.linalg.norm(g.reshape(-1)[pos] - ref) / \
            (np.linalg.norm(ref) + 1e-12)
        nrm = abs(np.linalg.norm(g) - float(t["gradnorm/" + v])) / \
            (float(t["gradnorm/" + v]) + 1e-12)
        worst = max(worst, fro, nrm)
        # (observed worst over the six graphs: 1.04e-3, car_fixed_T3 -- a few
        # ReLU masks of the float32 forward differ from the float64 evaluation of
        # the graph; 2.9e-6 for car_auto_T0, which has no GNN layer)
        assert fro < 2e-3 and nrm < 2e-3, (v, fro, nrm)
    print(name, "worst relative gradient error vs the reference TF graph "
          "%.3g" % worst)


@pytest.mark.parametrize("name", ["car_auto_T0", "car_auto_T3"])
def test_sgd_update_matches_reference_tf_graph(dev, name):
    """The `GradientDescent` train op (train.py:404): one applied step moves
    the first variable exactly where the reference graph moves it."""
    from pointgnn_amd import train
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    tc = configs.get_train_config(CHECKPOINTS[name])
    tr = train.Trainer(cfg, train_config=tc, params=w, device=dev)
    tr.global_step = int(t["global_step"])
    tr.train_step(train.batch_data(_batches(cfg, int(t["num_towers"]))))
    v0 = [k for k in t if k.startswith("updated/")][0][len("updated/"):]
    new = tr.state_dict()[v0]
    moved = np.abs(t["updated/" + v0] - np.asarray(w[v0])).max()
    assert moved > 0
    np.testing.assert_allclose(new, t["updated/" + v0], rtol=0,
                               atol=2e-3 * moved + 1e-8)


@pytest.mark.parametrize("name", ["car_auto_T0", "ped_cyl_auto_T3"])
def test_streaming_metrics_match_reference_tf_graph(dev, name):
    """tf.metrics.recall / precision / auc of tower 0 over two steps
    (train.py:301-373) from the device's own probabilities."""
    from pointgnn_amd import models
    from pointgnn_amd.metrics import StreamingMetrics
    t = gold("tfgraph_%s.npz" % name)
    cfg = configs.get_config(name)
    w, _ = fixture_weights(name, cfg)
    model = models.get_model(cfg["model_name"])(
        num_classes=cfg["num_classes"], box_encoding_len=7, mode="train",
        **cfg["model_kwargs"])
    model.load_state_dict(w)
    nt = int(t["num_towers"])
    m = StreamingMetrics(cfg["num_classes"])
    for step, first in ((1, 0), (2, nt - 1)):
        kw = tower_batch(cfg, first)             # what tower 0 is fed
        logits, _ = model.predict(kw["features"], kw["coords"],
                                  kw["keypoints"], kw["edges"],
                                  is_training=True)
        r = m.update(model.postprocess(logits), kw["labels"])
        for c in range(cfg["num_classes"]):
            for key in ("recall_%d", "precision_%d", "mAP_%d"):
                ref = float(t["metric%d/%s" % (step, key % c)])
                # a probability within float32 rounding of a threshold or an
                # arg-max tie may fall on the other side: 1 count in ~1e3
                assert abs(r[key % c] - ref) < 5e-3, (step, key % c, r[key % c], ref)
