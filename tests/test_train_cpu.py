"""Host-side pieces of the training step that need no GPU."""
import os

import numpy as np

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from pointgnn_amd.train import batch_data, learning_rate


def _frame(n, k, e0, e1, seed):
    rng = np.random.default_rng(seed)
    coords = [rng.standard_normal((n, 3)).astype(np.float32),
              rng.standard_normal((k, 3)).astype(np.float32)]
    coords.append(coords[1])
    kps = [rng.integers(0, n, (k, 1)).astype(np.int32),
           np.arange(k, dtype=np.int32).reshape(-1, 1)]
    edges = [np.stack([rng.integers(0, n, e0), np.sort(rng.integers(0, k, e0))], 1).astype(np.int32),
             np.stack([rng.integers(0, k, e1), np.sort(rng.integers(0, k, e1))], 1).astype(np.int32)]
    return (rng.random((n, 1)).astype(np.float32), coords, kps, edges,
            rng.integers(0, 4, (k, 1)).astype(np.int32),
            rng.standard_normal((k, 1, 7)).astype(np.float32),
            rng.random((k, 1, 1)).astype(np.float32))


def test_batch_data_offsets_like_reference():
    """train.py:135-171: point indices shift by the running point count of the
    level, centre indices by the running centre count."""
    f0, f1 = _frame(50, 7, 30, 40, 0), _frame(31, 5, 11, 13, 1)
    v, coords, kps, edges, labels, boxes, valid = batch_data([f0, f1])
    assert v.shape == (81, 1) and coords[0].shape == (81, 3)
    assert coords[1].shape == (12, 3) and labels.shape == (12, 1)
    assert np.array_equal(kps[0][:7], f0[2][0])
    assert np.array_equal(kps[0][7:], f1[2][0] + 50)        # level-0 points
    assert np.array_equal(kps[1][7:], f1[2][1] + 7)         # level-1 points
    assert np.array_equal(edges[0][30:, 0], f1[3][0][:, 0] + 50)
    assert np.array_equal(edges[0][30:, 1], f1[3][0][:, 1] + 7)
    assert np.array_equal(edges[1][40:, 0], f1[3][1][:, 0] + 7)
    assert np.array_equal(edges[1][40:, 1], f1[3][1][:, 1] + 7)
    assert np.all(np.diff(edges[1][:, 1]) >= 0)             # still grouped by dst
    assert boxes.shape == (12, 1, 7) and valid.shape == (12, 1, 1)


def test_learning_rate_schedule():
    tc = {'initial_lr': 0.125, 'decay_step': 400000, 'decay_factor': 0.1}
    assert learning_rate(tc, 0) == 0.125
    assert learning_rate(tc, 399999) == 0.125
    assert abs(learning_rate(tc, 400000) - 0.0125) < 1e-12
    assert abs(learning_rate(tc, 1399999) - 0.125e-3) < 1e-12


def test_mask_matched_oracle_replays_its_own_decisions():
    """oracle/train_oracle.Decisions: replaying the decisions a forward
    recorded reproduces its loss and gradients exactly; replaying a float32
    forward's decisions in float64 brings the two gradients to float32 noise
    (the mechanism behind tests/test_gpu_train.py::
    test_full_gradient_matches_mask_matched_oracle)."""
    import torch
    from oracle import train_oracle as to
    from pointgnn_amd import configs, weights
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                  "golden", "graph_tiny.npz")))
    k = g["kp_xyz"].shape[0]
    cfg = configs.get_config("car_auto_T3")
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 4, (k, 1)).astype(np.int32)
    labels[rng.random((k, 1)) < 0.5] = 0
    boxes = (rng.standard_normal((k, 1, 7)) * 1.5).astype(np.float32)
    valid = (labels > 0).astype(np.float32).reshape(k, 1, 1)
    batch = (g["intensity"], [g["xyz"], g["kp_xyz"], g["kp_xyz"]],
             [g["kp_idx"], np.arange(k, dtype=np.int32).reshape(-1, 1)],
             [g["ref_edges0"], g["ref_edges1"]], labels, boxes, valid)
    params = weights.init_params(cfg, seed=5, bias_scale=0.1)
    l0, g0, _ = to.step_gradients(params, cfg, [batch])
    rec = to.Decisions()
    l1, g1, _ = to.step_gradients(params, cfg, [batch], decisions=[rec])
    assert l0 == l1 and sum(rec.flips()) == 0
    rep = to.Decisions(rec.taken)
    l2, g2, _ = to.step_gradients(params, cfg, [batch], decisions=[rep])
    assert l0 == l2 and rep.pos == len(rec.taken)
    for n in g0:
        assert np.array_equal(g0[n], g1[n])
        np.testing.assert_allclose(g2[n], g0[n], rtol=0,
                                   atol=1e-12 * (np.abs(g0[n]).max() + 1e-30))
    rec32 = to.Decisions()
    _, g32, _ = to.step_gradients(params, cfg, [batch], dtype=torch.float32,
                                  decisions=[rec32])
    rep64 = to.Decisions(rec32.taken)
    _, g64, _ = to.step_gradients(params, cfg, [batch], decisions=[rep64])
    for n in g0:
        scale = np.abs(g64[n]).max() + 1e-30
        assert np.abs(g32[n] - g64[n]).max() <= 1e-5 * scale, n
