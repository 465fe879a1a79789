"""CPU restatement of the streaming metrics train.py:301-368 / eval.py:176-245
register.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).

PARITY UNPINNED at the TensorFlow boundary: the arithmetic lives in
tensorflow-gpu==1.15.0 (README.md:20-23, absent here, no network) --
`tf.metrics.recall`, `tf.metrics.precision`, `tf.metrics.mean` and
`tf.metrics.auc(num_thresholds=200, curve='PR',
summation_method='careful_interpolation')` of
tensorflow/python/ops/metrics_impl.py.  This file restates that published
algorithm the slow way TensorFlow does it (a [T, K] broadcast compare per
update, float32 counters); tests/test_metrics_cpu.py pins the parts that have
an independent implementation here (precision / recall against scikit-learn,
the PR area against scikit-learn's on score-quantised data).
"""
import numpy as np

K_EPSILON = 1e-7


def thresholds(num_thresholds=200):
    """metrics_impl.py auc(): Python doubles -> float32 constant."""
    inner = [(i + 1) * 1.0 / (num_thresholds - 1)
             for i in range(num_thresholds - 2)]
    return np.array([0.0 - K_EPSILON] + inner + [1.0 + K_EPSILON],
                    dtype=np.float32)


def _div_no_nan(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    out = np.zeros(np.broadcast(a, b).shape, np.float32)
    np.divide(a, b, out=out, where=(b != 0))
    return out


class StreamingMetricsOracle(object):
    def __init__(self, num_classes, num_thresholds=200):
        self.nc = num_classes
        self.T = num_thresholds
        self.thr = thresholds(num_thresholds)
        z = lambda *s: np.zeros(s, np.float32)
        self.tp, self.fp, self.fn = z(self.nc), z(self.nc), z(self.nc)
        self.atp, self.afp = z(self.nc, self.T), z(self.nc, self.T)
        self.afn, self.atn = z(self.nc, self.T), z(self.nc, self.T)

    def update(self, probs, labels):
        probs = np.asarray(probs, np.float32)
        labels = np.asarray(labels).reshape(-1)
        pred = np.argmax(probs, axis=1) if probs.shape[0] else \
            np.zeros((0,), np.int64)
        for c in range(self.nc):
            is_c, said_c = labels == c, pred == c
            self.tp[c] += np.float32(np.sum(is_c & said_c))
            self.fp[c] += np.float32(np.sum(~is_c & said_c))
            self.fn[c] += np.float32(np.sum(is_c & ~said_c))
            # _confusion_matrix_at_thresholds: pred_is_pos = predictions > thr
            pos = probs[None, :, c] > self.thr[:, None]
            self.atp[c] += np.sum(pos & is_c[None], axis=1).astype(np.float32)
            self.afp[c] += np.sum(pos & ~is_c[None], axis=1).astype(np.float32)
            self.afn[c] += np.sum(~pos & is_c[None], axis=1).astype(np.float32)
            self.atn[c] += np.sum(~pos & ~is_c[None], axis=1).astype(np.float32)

    def recall(self, c):
        d = self.tp[c] + self.fn[c]
        return float(self.tp[c] / d) if d > 0 else 0.0

    def precision(self, c):
        d = self.tp[c] + self.fp[c]
        return float(self.tp[c] / d) if d > 0 else 0.0

    def pr_auc(self, c):
        """interpolate_pr_auc (Davis & Goadrich 2006)."""
        n = self.T
        tp, fp, fn = self.atp[c], self.afp[c], self.afn[c]
        dtp = tp[:n - 1] - tp[1:]
        p = tp + fp
        slope = _div_no_nan(dtp, np.maximum(p[:n - 1] - p[1:], 0))
        intercept = tp[1:] - slope * p[1:]
        both = (p[:n - 1] > 0) & (p[1:] > 0)
        ratio = np.where(both, _div_no_nan(p[:n - 1], np.maximum(p[1:], 0)),
                         np.ones_like(p[1:]))
        inc = _div_no_nan(slope * (dtp + intercept * np.log(ratio)),
                          np.maximum(tp[1:] + fn[1:], 0))
        return float(np.sum(inc, dtype=np.float32))

    def result(self):
        out = {}
        for c in range(self.nc):
            out['recall_%d' % c] = self.recall(c)
            out['precision_%d' % c] = self.precision(c)
            out['mAP_%d' % c] = self.pr_auc(c)
        return out


def state_counts(probs, labels, num_classes, num_thresholds=200):
    """The int64 counters pgnn_metrics_update accumulates (layout stated in
    include/pointgnn_hip.h / csrc/metrics.hip): per class [tp, fp, fn,
    bins of the label==c rows (T+1), bins of the other rows (T+1)], bin =
    number of thresholds below the probability."""
    probs = np.asarray(probs, np.float32)
    labels = np.asarray(labels).reshape(-1)
    thr = thresholds(num_thresholds)
    pred = np.argmax(probs, axis=1) if probs.shape[0] else \
        np.zeros((0,), np.int64)
    block = 3 + 2 * (num_thresholds + 1)
    out = np.zeros((num_classes, block), np.int64)
    for c in range(num_classes):
        is_c, said_c = labels == c, pred == c
        out[c, 0] = np.sum(is_c & said_c)
        out[c, 1] = np.sum(~is_c & said_c)
        out[c, 2] = np.sum(is_c & ~said_c)
        bins = np.sum(probs[:, c][:, None] > thr[None, :], axis=1)
        out[c, 3:4 + num_thresholds] = np.bincount(
            bins[is_c], minlength=num_thresholds + 1)
        out[c, 4 + num_thresholds:] = np.bincount(
            bins[~is_c], minlength=num_thresholds + 1)
    return out.reshape(-1)


def synthetic_step(seed, n_rows, num_classes, sharp=3.0, grid=None):
    """Softmax outputs correlated with the labels; `grid` snaps the
    probabilities to multiples of 1/grid (ties and threshold hits)."""
    rng = np.random.default_rng(seed)
    labels = rng.choice(num_classes, size=n_rows,
                        p=[0.7] + [0.3 / (num_classes - 1)] * (num_classes - 1))
    logits = rng.standard_normal((n_rows, num_classes)).astype(np.float32)
    logits[np.arange(n_rows), labels] += np.float32(sharp) * \
        rng.random(n_rows).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    probs = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    if grid:
        probs = (np.round(probs * grid) / grid).astype(np.float32)
    return probs, labels.astype(np.int32).reshape(-1, 1)
