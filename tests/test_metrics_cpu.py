"""The metrics oracle against independent implementations (scikit-learn)."""
import numpy as np
import pytest

from oracle import metrics_oracle as mo


def test_thresholds_are_tensorflows():
    t = mo.thresholds(200)
    assert t.dtype == np.float32 and t.shape == (200,)
    assert t[0] < 0 and t[-1] > 1 and np.all(np.diff(t) > 0)
    assert t[1] == np.float32(1.0 / 199) and t[198] == np.float32(198.0 / 199)


def test_precision_recall_match_sklearn():
    from sklearn.metrics import precision_score, recall_score
    o = mo.StreamingMetricsOracle(4)
    ps, ls = [], []
    for s in range(3):  # three streamed steps == one concatenated set
        p, l = mo.synthetic_step(s, 700 + 100 * s, 4)
        o.update(p, l)
        ps.append(p)
        ls.append(l.reshape(-1))
    p, l = np.concatenate(ps), np.concatenate(ls)
    pred = p.argmax(1)
    for c in range(4):
        assert o.recall(c) == pytest.approx(
            recall_score(l == c, pred == c, zero_division=0), abs=1e-6)
        assert o.precision(c) == pytest.approx(
            precision_score(l == c, pred == c, zero_division=0), abs=1e-6)


def test_pr_auc_near_sklearn_on_the_threshold_grid():
    """With scores that sit strictly between thresholds the 200 operating
    points contain every distinct one, and the interpolated area must be close
    to the exact step-wise area scikit-learn integrates."""
    from sklearn.metrics import precision_recall_curve, auc
    o = mo.StreamingMetricsOracle(3)
    p, l = mo.synthetic_step(11, 5000, 3)
    # snap scores to cell midpoints of the threshold grid
    q = ((np.floor(p * 199) + 0.5) / 199).astype(np.float32)
    o.update(q, l)
    for c in range(3):
        prec, rec, _ = precision_recall_curve(l.reshape(-1) == c, q[:, c])
        ref = auc(rec, prec)
        assert abs(o.pr_auc(c) - ref) < 0.02, (c, o.pr_auc(c), ref)


def test_degenerate_cases():
    o = mo.StreamingMetricsOracle(2)
    assert o.result() == {'recall_0': 0.0, 'precision_0': 0.0, 'mAP_0': 0.0,
                          'recall_1': 0.0, 'precision_1': 0.0, 'mAP_1': 0.0}
    # a perfect classifier has PR area 1 for both classes
    probs = np.array([[0.9, 0.1]] * 5 + [[0.2, 0.8]] * 3, np.float32)
    labels = np.array([0] * 5 + [1] * 3)
    o.update(probs, labels)
    r = o.result()
    assert r['recall_0'] == 1.0 and r['precision_1'] == 1.0
    assert r['mAP_0'] == pytest.approx(1.0, abs=1e-5)
    assert r['mAP_1'] == pytest.approx(1.0, abs=1e-5)
