"""Streaming metrics of the training / evaluation loops (train.py:301-368,
eval.py:176-245), the step right after `model.loss`.

The reference registers one `tf.metrics.*` op per key and fetches all of
them with every `sess.run`; `StreamingMetrics.update` is that fetch: it folds
one step in and returns the running values under the reference's keys
(`recall_%d`, `precision_%d`, `mAP_%d`, `cls_loss`, `loc_loss`, `reg_loss`,
`total_loss`, `loc_loss_cls_%d`, `loc_loss_cls_%d_box_%d`).  `reset` is the
per-epoch `local_variables_initializer` (train.py:518-521, eval.py:302).

Class counters live on the device (csrc/metrics.hip) as int64, so N ranks can
add their states with one integer all-reduce (`sync`); the reference only
looks at tower 0 (train.py:299-301), which is what an un-synced rank 0 gives.
"""
import numpy as np
import torch

from . import _lib


def allreduce_state(state, group=None):
    """Add the int64 class counters of all ranks (RCCL on GPUs, gloo in the
    CPU tests); a no-op for one process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and \
            dist.get_world_size(group) > 1:
        dist.all_reduce(state, group=group)
    return state


class StreamingMetrics(object):
    def __init__(self, num_classes, box_encoding_len=7, num_thresholds=200,
                 device=None):
        self.lib = _lib.load()
        self.num_classes = int(num_classes)
        self.box_len = int(box_encoding_len)
        self.num_thresholds = int(num_thresholds)
        self.device = torch.device(device if device is not None else 'cuda')
        n_bytes = int(self.lib.pgnn_metrics_state_bytes(self.num_classes,
                                                        self.num_thresholds))
        if n_bytes <= 0:
            raise ValueError("num_classes / num_thresholds out of range")
        self.state = torch.zeros(n_bytes // 8, dtype=torch.int64,
                                 device=self.device)
        self._values = torch.empty((self.num_classes, 3), dtype=torch.float32,
                                   device=self.device)
        self.reset()

    def reset(self):
        """tf.variables_initializer(tf.local_variables())."""
        self.state.zero_()
        # tf.metrics.mean keeps float32 (total, count) pairs
        self._mean = {}

    def _mean_update(self, key, value):
        value = np.asarray(value, dtype=np.float32)
        total, count = self._mean.get(key, (np.float32(0), np.float32(0)))
        total = np.float32(total + value.sum(dtype=np.float32))
        count = np.float32(count + np.float32(value.size))
        self._mean[key] = (total, count)
        return float(total / count) if count != 0 else 0.0

    def update(self, probs, class_labels, loss_dict=None):
        """probs [K, nc] (model.postprocess), class_labels [K,1] or [K] int32,
        loss_dict: what model.loss / Trainer.train_step returned (optional).
        Returns the running values, like fetching metrics_update_ops."""
        p = torch.as_tensor(probs).to(self.device, torch.float32)
        lab = torch.as_tensor(class_labels).to(self.device, torch.int32)
        lab = lab.reshape(-1).contiguous()
        if p.dim() != 2 or p.shape[1] != self.num_classes:
            raise ValueError("probs must be [K, %d]" % self.num_classes)
        if p.stride(1) != 1:
            p = p.contiguous()
        if int(lab.shape[0]) != int(p.shape[0]):
            raise ValueError("labels do not match probs")
        _lib.check(self.lib.pgnn_metrics_update(
            _lib.ptr(p), p.stride(0) if p.shape[0] else self.num_classes,
            _lib.ptr(lab), int(p.shape[0]), self.num_classes,
            self.num_thresholds, _lib.ptr(self.state), _lib.stream_ptr()),
            "pgnn_metrics_update")
        if loss_dict is not None:
            self._loss_update(loss_dict)
        return self.result()

    def _loss_update(self, d):
        cls, loc, reg = (float(d['cls_loss']), float(d['loc_loss']),
                         float(d['reg_loss']))
        self._mean_update('cls_loss', cls)
        self._mean_update('loc_loss', loc)
        self._mean_update('reg_loss', reg)
        total = np.float32(np.float32(np.float32(cls) + np.float32(loc)) +
                           np.float32(reg))  # train.py:265
        self._mean_update('total_loss', total)
        if 'classwise_loc_loss' in d:
            for c, row in enumerate(d['classwise_loc_loss']):
                row = np.asarray(torch.as_tensor(row).detach().cpu()
                                 if isinstance(row, torch.Tensor) else row,
                                 dtype=np.float32).reshape(-1)
                # tf.metrics.mean of a [7] vector: mean over its elements too
                self._mean_update('loc_loss_cls_%d' % c, row)
                for b in range(row.shape[0]):
                    self._mean_update('loc_loss_cls_%d_box_%d' % (c, b), row[b])

    def sync(self, group=None):
        """Add the class counters of all ranks (one int64 all-reduce)."""
        allreduce_state(self.state, group)

    def result(self):
        _lib.check(self.lib.pgnn_metrics_compute(
            _lib.ptr(self.state), self.num_classes, self.num_thresholds,
            _lib.ptr(self._values), _lib.stream_ptr()), "pgnn_metrics_compute")
        vals = self._values.cpu().numpy()
        out = {}
        for c in range(self.num_classes):
            out['recall_%d' % c] = float(vals[c, 0])
            out['precision_%d' % c] = float(vals[c, 1])
            out['mAP_%d' % c] = float(vals[c, 2])
        for key, (total, count) in self._mean.items():
            out[key] = float(total / count) if count != 0 else 0.0
        return out

    def format(self, results):
        """The per-class lines train.py:598-616 prints."""
        lines = []
        for c in range(self.num_classes):
            lines.append('Class_%d: recall=%f, prec=%f, mAP=%f, loc=%f' % (
                c, results['recall_%d' % c], results['precision_%d' % c],
                results['mAP_%d' % c], results.get('loc_loss_cls_%d' % c, 0.0)))
            if 'loc_loss_cls_%d_box_0' % c in results:
                lines.append(
                    "         x=%.4f y=%.4f z=%.4f l=%.4f h=%.4f w=%.4f y=%.4f"
                    % tuple(results['loc_loss_cls_%d_box_%d' % (c, b)]
                            for b in range(7)))
        return "\n".join(lines)
