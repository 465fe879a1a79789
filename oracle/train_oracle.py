"""ORACLE (test infrastructure only) -- CPU restatement of the reference's
training step: forward (oracle/gnn_oracle.py semantics), loss
(`models/models.py:170-311`), tower weighting (`train.py:264-288`) and
gradients, in float64 on torch-CPU autograd.

PARITY UNPINNED at the TensorFlow boundary (TF 1.15 not installable; the
reference ships no gradient fixtures).  Documented TF semantics restated:
  * tf.nn.sparse_softmax_cross_entropy_with_logits: logsumexp(z) - z[label];
  * tf.losses.huber_loss(delta=1, weights=valid, reduction=NONE):
    0.5*min(|e|,1)^2 + (|e| - min(|e|,1)), times valid;
  * gradient of tf.math.unsorted_segment_max: split equally among the rows that
    equal the segment max (torch's scatter_reduce 'amax' has the same rule);
  * slim.l1_regularizer(scale)(W) = scale * sum|W| on FC weights only.

MASK-MATCHED EVALUATION (round 4).  The loss is piecewise smooth: which side
of zero a pre-activation falls on (ReLU) and which edge row wins a segment
maximum are DECISIONS; a float32 and a float64 forward can take a handful of
them differently, and then single gradient entries legitimately differ.  To
separate that from a wrong formula, `forward` / `step_gradients` take a
`Decisions` object: in record mode it notes every decision the float64 forward
takes; in replay mode it applies decisions supplied from outside (the
device's: y > 0 of its saved activations, row == segment max of its saved
rows) -- ReLU becomes a multiplication by the given 0/1 mask and segment-max
the mean of the given winner rows (TF's equal split among ties) -- so the
float64 gradient is that of the SAME smooth piece the device differentiated.
Decision order: the order `forward` reaches them (per stage: [auto-offset
ReLUs], per-edge MLP ReLUs, the segment max, update/output MLP ReLUs; then
cls head, then each loc head).
Only tests/ may import this module.
"""
import numpy as np
import torch


class Decisions(object):
    """Recorder / replayer of the forward's non-smooth choices (see header).
    `supplied`: list of boolean arrays in decision order, or None to record.
    After a forward: `.taken` = the list of masks used, `.own` = the masks the
    float64 values themselves would have chosen (for counting flips)."""

    def __init__(self, supplied=None):
        self.supplied = list(supplied) if supplied is not None else None
        self.pos = 0
        self.taken, self.own = [], []

    def _next(self, own):
        own = np.asarray(own)
        self.own.append(own)
        if self.supplied is None:
            self.taken.append(own)
            return None
        m = np.asarray(self.supplied[self.pos]).astype(bool)
        assert m.shape == own.shape, "decision %d: mask %s, site %s" % (
            self.pos, m.shape, own.shape)
        self.pos += 1
        self.taken.append(m)
        return m

    def relu(self, x):
        m = self._next((x.detach() > 0).numpy())
        if m is None:
            return torch.relu(x)
        return x * torch.as_tensor(m, dtype=x.dtype)

    def segment_max(self, data, seg, num):
        plain = _segment_max(data, seg, num)
        m = self._next((data.detach() == plain.detach()[seg]).numpy())
        if m is None:
            return plain
        w = torch.as_tensor(m, dtype=data.dtype)
        idx = seg.reshape(-1, 1).expand(-1, data.shape[1])
        cnt = torch.zeros((num, data.shape[1]), dtype=data.dtype)
        cnt = cnt.scatter_add(0, idx, w)
        tot = torch.zeros((num, data.shape[1]), dtype=data.dtype)
        tot = tot.scatter_add(0, idx, data * w)
        return tot / torch.clamp(cnt, min=1.0)

    def flips(self):
        """Per decision site: number of entries where the float64 forward's
        own choice differs from the one applied."""
        return [int((a != b).sum()) for a, b in zip(self.own, self.taken)]


def _mlp(x, params, scope, n_layers, is_logits, dec=None):
    for i in range(n_layers):
        name = scope + '/fully_connected' + ('' if i == 0 else '_%d' % i)
        x = x @ params[name + '/weights'] + params[name + '/biases']
        if not (is_logits and i == n_layers - 1):
            x = torch.relu(x) if dec is None else dec.relu(x)
    return x


def _segment_max(data, seg, num):
    idx = seg.reshape(-1, 1).expand(-1, data.shape[1])
    out = torch.full((num, data.shape[1]), float('-inf'), dtype=data.dtype)
    return out.scatter_reduce(0, idx, data, 'amax', include_self=True)


def forward(params, config, input_v, coords, kps, edges, dtype=torch.float64,
            decisions=None):
    """params: {name: torch float64 tensor (requires_grad)}."""
    dec = decisions
    smax = _segment_max if dec is None else dec.segment_max
    t64 = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    i64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int64)
    feats = t64(input_v)
    coords = [t64(c) for c in coords]
    for lc in config['model_kwargs']['layer_configs'][:-1]:
        lvl, scope, kw = lc['graph_level'], lc['scope'], lc['kwargs']
        e = i64(edges[lvl])
        src, dst = e[:, 0], e[:, 1]
        if lc['type'] == 'scatter_max_point_set_pooling':
            kp = i64(kps[lvl]).reshape(-1)
            x = coords[lvl]
            f = torch.cat([feats[src], x[src] - x[kp[dst]]], dim=1)
            f = _mlp(f, params, scope + '/extract_vertex_features',
                     len(kw['point_MLP_depth_list']), False, dec)
            agg = smax(f, dst, kp.shape[0])
            feats = _mlp(agg, params, scope + '/combined_features',
                         len(kw['output_MLP_depth_list']), False, dec)
        else:
            x = coords[lvl]
            h = feats
            s_h, s_x = h[src], x[src]
            if kw['auto_offset']:
                x = x + _mlp(h, params, scope,
                             len(kw['auto_offset_MLP_depth_list']), True, dec)
            ef = torch.cat([s_h, s_x - x[dst]], dim=1)
            ef = _mlp(ef, params, scope + '/extract_vertex_features',
                      len(kw['edge_MLP_depth_list']), False, dec)
            agg = smax(ef, dst, h.shape[0])
            feats = _mlp(agg, params, scope + '/combined_features',
                         len(kw['update_MLP_depth_list']), True, dec) + h
    ps = config['model_kwargs']['layer_configs'][-1]['scope'] + '/predictor'
    logits = _mlp(feats, params, ps + '/cls', 2, True, dec)
    boxes = [_mlp(feats, params, ps + '/loc/cls_%d' % j, 3, True,
                  dec)[:, None, :]
             for j in range(config['num_classes'])]
    return logits, torch.cat(boxes, dim=1)


def loss_terms(config, logits, pred_box, labels, gt_box, valid,
               dtype=torch.float64):
    """Returns (sum CE, sum loc, n, n_valid) as torch scalars."""
    lab = torch.as_tensor(np.asarray(labels), dtype=torch.int64).reshape(-1)
    gt = torch.as_tensor(np.asarray(gt_box), dtype=dtype).reshape(
        lab.shape[0], -1)
    va = torch.as_tensor(np.asarray(valid), dtype=dtype).reshape(-1)
    ce = torch.logsumexp(logits, dim=1) - logits[torch.arange(len(lab)), lab]
    kind = config['loss'].get('cls_loss_type', 'softmax')
    kw = config['loss'].get('cls_loss_kwargs', {}) or {}
    if kind == 'focal_softmax':
        # models/loss.py:31-48: (1 - softmax(z)[label])^gamma * CE
        py = torch.softmax(logits, dim=1)[torch.arange(len(lab)), lab]
        ce = (1.0 - py) ** kw.get('gamma', 2) * ce
    elif kind == 'focal_sigmoid':
        # models/loss.py:5-29: per (vertex, class) sigmoid CE, modulated by
        # (1 - p_t)^gamma and weighted alpha / 1 - alpha; models.py:229 takes
        # the mean over vertices AND classes
        t = torch.nn.functional.one_hot(lab, logits.shape[1]).to(dtype)
        p = torch.sigmoid(logits)
        xent = torch.clamp(logits, min=0) - logits * t + \
            torch.log1p(torch.exp(-logits.abs()))
        pt = t * p + (1 - t) * (1 - p)
        alpha = kw.get('alpha', 0.5)
        aw = t * alpha + (1 - t) * (1 - alpha)
        ce = ((1 - pt) ** kw.get('gamma', 2) * aw * xent).mean(dim=1)
    elif kind == 'top_k_softmax':
        # models.py:222-228: mean of the k largest per-vertex CE values.  The
        # callers divide the returned sum by n (and the towers re-weight by
        # n_tower / n_total, train.py:268-284), so the k selected values are
        # returned scaled by n / k.  Stable descending sort = tf.math.top_k's
        # order (equal values: lower index first).
        k = int(kw['k'])
        order = torch.sort(ce.detach(), descending=True, stable=True)[1][:k]
        pick = torch.zeros_like(ce)
        pick[order] = 1.0
        ce = ce * pick * (float(len(lab)) / k)
    elif kind != 'softmax':
        raise NotImplementedError(kind)
    pb = pred_box[torch.arange(len(lab)), lab]
    err = pb - gt
    ae = err.abs()
    quad = torch.clamp(ae, max=1.0)
    hub = (0.5 * quad * quad + (ae - quad)) * va[:, None]
    lkw = config['loss'].get('loc_loss_kwargs', {}) or {}
    if 'classwise_loc_loss_weight' in lkw:   # models.py:240-246 (train mode)
        cw = torch.as_tensor(np.asarray(lkw['classwise_loc_loss_weight']),
                             dtype=dtype)
        hub = hub * cw[lab][:, None]
    loc = hub.mean(dim=1)
    if config['loss'].get('loc_loss_type', 'huber_loss') == 'top_k_huber_loss':
        # models.py:266-291: the k largest per-vertex means; num_valid_endpoint
        # counts the valid vertices among them
        k = int(lkw['k'])
        order = torch.sort(loc.detach(), descending=True, stable=True)[1][:k]
        pick = torch.zeros_like(loc)
        pick[order] = 1.0
        return (ce.sum(), (loc * pick).sum(), float(len(lab)),
                float((va * pick).sum()))
    return ce.sum(), loc.sum(), float(len(lab)), float(va.sum())


def step_gradients(np_params, config, rank_batches, dtype=torch.float64,
                   decisions=None):
    """Global loss and gradients for a list of per-rank batches, following
    train.py:264-297 + util/tf_util.py:3-43: every tower's cls/loc loss is
    re-weighted by its share of (valid) endpoints and the tower gradients are
    averaged -- which equals the gradient of the global per-vertex means.
    Returns (loss dict, {name: ndarray grad of cls+loc}, {name: grad of reg}).
    decisions: one `Decisions` per batch (see the module header) or None."""
    params = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)
              for k, v in np_params.items()}
    cls_w = config['loss']['cls_loss_weight']
    loc_w = config['loss']['loc_loss_weight']
    terms = []
    for bi, (input_v, coords, kps, edges, labels, boxes,
             valid) in enumerate(rank_batches):
        logits, pred = forward(params, config, input_v, coords, kps, edges,
                               dtype, None if decisions is None
                               else decisions[bi])
        terms.append(loss_terms(config, logits, pred, labels, boxes, valid,
                                dtype))
    n_tot = sum(t[2] for t in terms)
    nv_tot = sum(t[3] for t in terms)
    cls = cls_w * sum(t[0] for t in terms) / n_tot
    loc = loc_w * sum(t[1] for t in terms) / nv_tot if nv_tot > 0 else \
        torch.zeros((), dtype=dtype)
    scale = config['model_kwargs']['regularizer_kwargs']['scale']
    reg = scale * sum(p.abs().sum() for k, p in params.items()
                      if k.endswith('/weights'))
    names = list(params)
    g_data = torch.autograd.grad(cls + loc, [params[n] for n in names],
                                 allow_unused=True, retain_graph=True)
    g_reg = torch.autograd.grad(reg, [params[n] for n in names],
                                allow_unused=True)
    z = lambda g, n: (np.zeros(params[n].shape) if g is None
                      else g.detach().numpy())
    return ({'cls_loss': float(cls.detach()), 'loc_loss': float(loc.detach()),
             'reg_loss': float(reg.detach()), 'num_endpoint': n_tot,
             'num_valid_endpoint': nv_tot},
            {n: z(g, n) for n, g in zip(names, g_data)},
            {n: z(g, n) for n, g in zip(names, g_reg)})
