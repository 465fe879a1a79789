"""HIP streaming metrics (pgnn_metrics_update / pgnn_metrics_compute through
the C-ABI) against the oracle's restatement of tf.metrics.* (train.py:301-368).

Bars: the int64 counters bit-exact; recall / precision / PR-AUC within 2e-6
(float32 arithmetic, reduction order differs from NumPy's pairwise sum)."""
import numpy as np
import pytest
import torch

import pointgnn_amd  # noqa: F401
from oracle import metrics_oracle as mo

pytestmark = pytest.mark.gpu


def _check(m, o, tol=2e-6):
    got, ref = m.result(), o.result()
    for k, v in ref.items():
        assert got[k] == pytest.approx(v, abs=tol), k


@pytest.mark.parametrize("nc,rows,grid", [(4, 3000, None), (6, 11000, None),
                                           (4, 2500, 199), (2, 70, 4)])
def test_streamed_updates_match_oracle(nc, rows, grid):
    from pointgnn_amd.metrics import StreamingMetrics
    m = StreamingMetrics(nc)
    o = mo.StreamingMetricsOracle(nc)
    expect = np.zeros(m.state.numel(), np.int64)
    for step in range(4):
        p, l = mo.synthetic_step(100 * nc + step, rows + 37 * step, nc, grid=grid)
        if step == 2:  # device tensors with a padded row stride are accepted
            wide = torch.zeros((p.shape[0], 8), device='cuda')
            wide[:, :nc] = torch.from_numpy(p).cuda()
            m.update(wide[:, :nc], torch.from_numpy(l).cuda())
        else:
            m.update(p, l)
        o.update(p, l)
        expect += mo.state_counts(p, l, nc)
        assert np.array_equal(m.state.cpu().numpy(), expect)
        _check(m, o)
    m.reset()
    assert int(m.state.abs().sum()) == 0
    assert m.result()['mAP_0'] == 0.0


def test_threshold_hits_ties_and_extremes():
    """Probabilities exactly on thresholds (strict >), exact 0 and 1, argmax
    ties (first index wins), a class that never occurs, an empty batch."""
    from pointgnn_amd.metrics import StreamingMetrics
    nc = 3
    thr = mo.thresholds(200)
    vals = np.concatenate([thr[1:-1], np.nextafter(thr[1:-1], np.float32(2)),
                           np.float32([0.0, 1.0, 0.5, 0.5])]).astype(np.float32)
    probs = np.stack([vals, vals, 1 - vals], axis=1).astype(np.float32)
    labels = (np.arange(vals.shape[0]) % 2).astype(np.int32)  # class 2 absent
    m, o = StreamingMetrics(nc), mo.StreamingMetricsOracle(nc)
    m.update(probs[:0], labels[:0])
    assert int(m.state.abs().sum()) == 0
    m.update(probs, labels)
    o.update(probs, labels)
    assert np.array_equal(m.state.cpu().numpy(),
                          mo.state_counts(probs, labels, nc))
    _check(m, o)
    assert m.result()['recall_2'] == 0.0


def test_loss_means_and_reference_keys():
    from pointgnn_amd.metrics import StreamingMetrics
    m = StreamingMetrics(4)
    rng = np.random.default_rng(0)
    tot = {k: [] for k in ('cls_loss', 'loc_loss', 'reg_loss', 'total_loss')}
    cw = []
    for step in range(3):
        p, l = mo.synthetic_step(step, 500, 4)
        d = {'cls_loss': float(rng.random()), 'loc_loss': float(rng.random()),
             'reg_loss': float(rng.random()),
             'classwise_loc_loss': [rng.random(7).astype(np.float32)
                                    for _ in range(4)]}
        r = m.update(p, l, d)
        for k in ('cls_loss', 'loc_loss', 'reg_loss'):
            tot[k].append(d[k])
        tot['total_loss'].append(d['cls_loss'] + d['loc_loss'] + d['reg_loss'])
        cw.append(np.stack(d['classwise_loc_loss']))
    for k, v in tot.items():
        assert r[k] == pytest.approx(np.mean(v), rel=1e-6)
    cw = np.stack(cw)  # [steps, nc, 7]
    for c in range(4):
        assert r['loc_loss_cls_%d' % c] == pytest.approx(cw[:, c].mean(), rel=1e-6)
        for b in range(7):
            assert r['loc_loss_cls_%d_box_%d' % (c, b)] == \
                pytest.approx(cw[:, c, b].mean(), rel=1e-6)
    keys = set(r)
    for c in range(4):
        assert {'recall_%d' % c, 'precision_%d' % c, 'mAP_%d' % c} <= keys
    assert 'Class_3: recall=' in m.format(r)


def test_metrics_follow_a_training_step():
    """probs of the trained model + the labels of the batch, the way
    train.py fetches them together with the loss."""
    from pointgnn_amd import configs, graph_gen, train
    from pointgnn_amd.metrics import StreamingMetrics
    from pointgnn_amd.synthetic import synthetic_cloud
    cfg = configs.get_config("car_auto_T1")
    dev = torch.device('cuda')
    tr = train.Trainer(cfg, seed=0, device=dev)
    xyz, inten = synthetic_cloud(seed=0, preset='tiny')
    fn = graph_gen.get_graph_generate_fn(cfg['graph_gen_method'])
    coords, kps, edges = fn(torch.from_numpy(xyz).to(dev),
                            **cfg['runtime_graph_gen_kwargs'])
    k = int(coords[1].shape[0])
    rng = np.random.default_rng(1)
    labels = rng.integers(0, 4, (k, 1)).astype(np.int32)
    logits, _ = tr.forward(torch.from_numpy(inten).to(dev), coords, kps, edges)
    probs = torch.softmax(logits[:, :4], dim=1)
    m, o = StreamingMetrics(4), mo.StreamingMetricsOracle(4)
    m.update(probs, labels)
    o.update(probs.cpu().numpy(), labels)
    _check(m, o)
