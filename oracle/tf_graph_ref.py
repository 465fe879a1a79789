"""ORACLE (test infrastructure only) -- the reference's own training graph as a
callable.

Only `tests/` and `tests/golden/make_golden_tfgraph.py` may import this module;
the product path never does.

`ReferenceGraph(meta_path)` wraps `oracle/tf_meta_interp.Graph` around one of
the MetaGraphDefs the reference ships (`checkpoints/*/model-*.meta`, written by
`train.py:496,634-636`) and locates -- structurally, from the wiring
`train.py:178-405` leaves behind, not from hard-coded node numbers --
  * the placeholders of every tower, in the creation order of
    `train.py:183-222` (features, coords x(L+1), edges xL, keypoint indices xL,
    class labels, encoded boxes, valid boxes, is_training);
  * `t_logits`, `t_pred_box`, `t_probs` of every tower (`train.py:227-231`,
    i.e. `models/models.py:79-168` -> `models/gnn.py:133-373`);
  * the per-tower and cross-tower losses after the `unify_copies` re-weighting
    (`train.py:232-299`, `models/models.py:170-311`);
  * each tower's `tf.gradients` result per variable and their mean
    (`train.py:397-404`, `util/tf_util.py:3-43`);
  * the `GradientDescent` train op with its `ExponentialDecay` learning rate
    (`train.py:375-405`) and the `tf.metrics` update ops (`train.py:301-373`).
A tower whose placeholders are not fed is simply never evaluated (fetching
tower 0's logits touches tower 0 only), so the same graph serves as the
inference reference for `run.py`'s `model.predict` (`run.py:135-141`: the same
`predict` code with `mode='test'`, which changes nothing for the shipped
configs -- no dropout, no batch norm).
"""
import re

import numpy as np

from . import tf_meta_interp as ti


def _is_grad(name):
    return name.startswith("gradients")


class ReferenceGraph(object):
    def __init__(self, meta_path):
        self.g = g = ti.load_meta(meta_path)
        self.consumers = {}
        for name in g.order:
            for s in g.nodes[name].inputs:
                d, _ = ti._split_input(s)
                self.consumers.setdefault(d, []).append(name)
        # ---- towers
        xent = [n for n in g.order
                if g.nodes[n].op == "SparseSoftmaxCrossEntropyWithLogits"
                and not _is_grad(n)]
        self.num_towers = len(xent)
        ph = [n for n in g.order if g.nodes[n].op == "Placeholder"
              and g.nodes[n].device]
        per = len(ph) // self.num_towers
        assert per * self.num_towers == len(ph) and (per - 6) % 3 == 0
        self.num_levels = L = (per - 6) // 3
        self.towers = []
        boxes = [n for n in g.order
                 if re.match(r"^[^/]+/predictor/concat$", n)]
        softmax = [n for n in g.order if g.nodes[n].op == "Softmax"
                   and not _is_grad(n)]
        assert len(boxes) == len(softmax) == self.num_towers
        for t in range(self.num_towers):
            p = ph[t * per:(t + 1) * per]
            tw = dict(
                features=p[0], coords=p[1:2 + L], edges=p[2 + L:2 + 2 * L],
                keypoints=p[2 + 2 * L:2 + 3 * L], labels=p[2 + 3 * L],
                gt_boxes=p[3 + 3 * L], valid=p[4 + 3 * L],
                is_training=p[5 + 3 * L],
                logits=ti._split_input(g.nodes[xent[t]].inputs[0])[0],
                box_encodings=boxes[t], probs=softmax[t])
            assert g.nodes[softmax[t]].inputs[0] == g.nodes[xent[t]].inputs[0]
            self.towers.append(tw)
        # ---- cross-tower losses: the value fed to each tf.metrics.mean
        self.cross = {}
        for key in ("cls", "loc", "reg", "total"):
            self.cross[key] = self._scope_value_input("mean_%s_loss" % key)
        # per-tower (re-weighted) losses are the Pack inputs of those means
        for key in ("cls", "loc", "reg", "total"):
            mean = g.nodes[self.cross[key]]
            assert mean.op == "Mean", mean
            pack = g.nodes[ti._split_input(mean.inputs[0])[0]]
            assert pack.op == "Pack" and len(pack.inputs) == self.num_towers
            for t, s in enumerate(pack.inputs):
                self.towers[t][key + "_loss"] = s
        # ---- gradients
        self.apply = {}
        self.avg_grad = {}
        self.tower_grad = {}
        for n in g.order:
            nd = g.nodes[n]
            if nd.op != "ApplyGradientDescent":
                continue
            var = nd.inputs[0]
            self.apply[var] = n
            self.lr = nd.inputs[1]
            self.avg_grad[var] = nd.inputs[2]
            mean = g.nodes[ti._split_input(nd.inputs[2])[0]]
            cat = g.nodes[ti._split_input(mean.inputs[0])[0]]
            assert mean.op == "Mean" and cat.op == "ConcatV2"
            per_tower = []
            for s in cat.inputs[:-1]:
                ex = g.nodes[ti._split_input(s)[0]]
                assert ex.op == "ExpandDims"
                per_tower.append(ex.inputs[0])
            assert len(per_tower) == self.num_towers
            self.tower_grad[var] = per_tower
        self.train_op = g.collections["train_op"][0]
        # tf.metrics.* update ops (train.py:301-373), keyed by the reference's
        # metric names ('recall_0', 'mAP_2', 'loc_loss_cls_1_box_3', ...)
        scopes = []
        for v in g.collections.get("metric_variables", []):
            sc = v.split("/")[0]
            if sc not in scopes:
                scopes.append(sc)
        self.metric_update = {}
        self.metric_value = {}
        for sc in scopes:
            # tf.metrics.{mean,recall,precision}: '<scope>/value|update_op';
            # tf.metrics.auc(careful_interpolation): interpolate_pr_auc[_1]
            for val, upd in (("/value", "/update_op"),
                             ("/interpolate_pr_auc", "/interpolate_pr_auc_1")):
                if sc + upd in g.nodes:
                    self.metric_value[sc] = sc + val
                    self.metric_update[sc] = sc + upd
            assert sc in self.metric_update, sc
        self.variable_names = g.trainable_variables()
        assert set(self.variable_names) == set(self.apply)

    def _scope_value_input(self, scope):
        """The one non-constant tensor entering name scope `scope/` from
        outside (the `values` argument of tf.metrics.mean)."""
        g = self.g
        ext = set()
        for n in g.order:
            if not n.startswith(scope + "/"):
                continue
            for s in g.nodes[n].inputs:
                d, k = ti._split_input(s)
                if k is None or d.startswith(scope + "/"):
                    continue
                if g.nodes[d].op in ("Const", "VariableV2"):
                    continue
                ext.add(d)
        assert len(ext) == 1, (scope, ext)
        return ext.pop()

    # ------------------------------------------------------------ variables
    def variable_shapes(self):
        return {v: self.g.variable_shape(v) for v in self.variable_names}

    def set_weights(self, params, global_step=0):
        """`params`: {TF variable name: ndarray} (the checkpoint dict)."""
        missing = [v for v in self.variable_names if v not in params]
        if missing:
            raise KeyError("no value for %s" % missing[:3])
        self.g.set_variables({v: params[v] for v in self.variable_names})
        self.g.set_variables({"Variable": np.int32(global_step)})

    def reset_metrics(self):
        """tf.local_variables_initializer(): metric accumulators to zero."""
        for v in self.g.collections.get("local_variables", []):
            nd = self.g.nodes[v]
            self.g.variables[v] = np.zeros(nd.attr("shape")[1],
                                           dtype=nd.dtype("dtype"))

    # ----------------------------------------------------------------- feeds
    def feed(self, tower, features, coords, keypoints, edges, labels=None,
             gt_boxes=None, valid=None, is_training=True):
        tw = self.towers[tower]
        fd = {tw["features"]: np.asarray(features, np.float32)}
        for p, c in zip(tw["coords"], coords):
            fd[p] = np.asarray(c, np.float32)
        for p, e in zip(tw["edges"], edges):
            fd[p] = np.asarray(e, np.int32)
        for p, k in zip(tw["keypoints"], keypoints):
            fd[p] = np.asarray(k, np.int32).reshape(-1, 1)
        fd[tw["is_training"]] = np.bool_(is_training)
        if labels is not None:
            fd[tw["labels"]] = np.asarray(labels, np.int32).reshape(-1, 1)
            fd[tw["gt_boxes"]] = np.asarray(gt_boxes, np.float32)
            fd[tw["valid"]] = np.asarray(valid, np.float32)
        return fd

    # ------------------------------------------------------------ evaluation
    def predict(self, features, coords, keypoints, edges, tower=0,
                extra_fetches=()):
        """(logits [K,nc], box_encodings [K,nc,7], probs) of one tower."""
        tw = self.towers[tower]
        fd = self.feed(tower, features, coords, keypoints, edges,
                       is_training=False)
        out = self.g.run([tw["logits"], tw["box_encodings"], tw["probs"]] +
                         list(extra_fetches), fd)
        return out if extra_fetches else tuple(out)

    def losses_and_gradients(self, tower_inputs, metrics=False):
        """One `sess.run` of everything `train.py:546-577` fetches except the
        variable update.  `tower_inputs`: one dict of `feed` kwargs per tower.
        -> dict(cls_loss, loc_loss, reg_loss, total_loss [cross-tower means],
                tower_losses [per tower dicts], grads {var: mean gradient},
                tower_grads {var: [per tower]}, learning_rate[, metrics]).
        `metrics=True` also runs every tf.metrics update op once (call
        `reset_metrics()` first, like train.py:523)."""
        assert len(tower_inputs) == self.num_towers
        fd = {}
        for t, kw in enumerate(tower_inputs):
            fd.update(self.feed(t, **kw))
        names = [self.cross[k] for k in ("cls", "loc", "reg", "total")]
        for tw in self.towers:
            names += [tw[k + "_loss"] for k in ("cls", "loc", "reg", "total")]
        names += [self.avg_grad[v] for v in self.variable_names]
        for v in self.variable_names:
            names += self.tower_grad[v]
        names.append(self.lr)
        mkeys = sorted(self.metric_update) if metrics else []
        names += [self.metric_update[k] for k in mkeys]
        vals = self.g.run(names, fd)
        it = iter(vals)
        out = {k + "_loss": next(it) for k in ("cls", "loc", "reg", "total")}
        out["tower_losses"] = [
            {k + "_loss": next(it) for k in ("cls", "loc", "reg", "total")}
            for _ in self.towers]
        out["grads"] = {v: next(it) for v in self.variable_names}
        out["tower_grads"] = {v: [next(it) for _ in self.towers]
                              for v in self.variable_names}
        out["learning_rate"] = next(it)
        if metrics:   # running values after this step's update (train.py:568)
            out["metrics"] = {k: next(it) for k in mkeys}
        return out

    def train_step(self, tower_inputs):
        """Run the `GradientDescent` train op once (`train.py:546-577` with
        only 'train_op' fetched): variables and global_step are updated in
        place.  Returns the new {var: value}."""
        fd = {}
        for t, kw in enumerate(tower_inputs):
            fd.update(self.feed(t, **kw))
        self.g.run([self.train_op], fd)
        return {v: self.g.variables[v].copy() for v in self.variable_names}
