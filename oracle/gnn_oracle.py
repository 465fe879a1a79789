"""ORACLE (test infrastructure only) -- NumPy restatement of the reference's GNN
arithmetic (`models/gnn.py`, `models/models.py:predict/postprocess`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product path never does.

PARITY UNPINNED at the TensorFlow boundary: the arithmetic lives in the
third-party `tensorflow-gpu==1.15.0` (README.md:20-23), which is neither under
/root/reference nor installable here, and the reference ships no test or
golden vector for this path.  This file therefore restates the
*documented* TF 1.15 op semantics at the reference's call sites:
  * `slim.fully_connected(x, n, activation_fn, normalizer_fn=None)` =
    act(x @ W + b), W [in,out]  (gnn.py:63-103);
  * `tf.gather` (gnn.py:256-262, 338-339, 348), `tf.concat` (:266, :350);
  * `tf.math.unsorted_segment_max` (gnn.py:107): per-segment channel-wise
    max; an empty segment yields float32 lowest();
  * `tf.nn.softmax` (models.py:167).
What *is* pinned: the trained `car_auto_T0` / `car_auto_T1` weights (read
without TF by point-gnn_amd/tf_bundle.py) flow through this restatement to
committed golden logits (tests/golden), and every function runs in float32
(`dtype=np.float32`, the reference's precision) or float64 (shadow used to
bound rounding).
"""
import numpy as np

FLOAT32_LOWEST = np.finfo(np.float32).min


# ------------------------------------------------------------------ MLPs
BATCH_NORM_EPSILON = 0.001   # slim.batch_norm's default


def fully_connected(x, w, b, relu):
    """slim.fully_connected (gnn.py:93-103).  `b` an array: normalizer 'NONE',
    act(x W + b).  `b` a tuple (moving_mean, moving_variance, beta or None,
    gamma or None): the batch-norm normalizers of gnn.py:17-23 at inference
    (slim creates no biases then) -- tf.nn.batch_normalization's documented
    (y - mean) * rsqrt(var + eps) [* gamma] [+ beta], NOT folded, so that the
    product's folded weights are checked against the unfolded expression.
    UNPINNED: no shipped config or checkpoint uses a normalizer."""
    if isinstance(b, tuple):
        mean, var, beta, gamma = b
        y = (x @ w - mean) / np.sqrt(var + x.dtype.type(BATCH_NORM_EPSILON))
        if gamma is not None:
            y = y * gamma
        if beta is not None:
            y = y + beta
    else:
        y = x @ w + b
    if relu:
        y = np.maximum(y, 0)
    return y


def multi_layer_neural_network(x, layers, is_logits):
    """gnn.py:86-104.  `layers` = [(W, b), ...]; ReLU on every layer except the
    last one when is_logits."""
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        x = fully_connected(x, w, b, relu=not (is_logits and i == n - 1))
    return x


def scatter_max(data, seg_ids, num_segments):
    """tf.math.unsorted_segment_max (gnn.py:106-109)."""
    out = np.full((num_segments, data.shape[1]),
                  FLOAT32_LOWEST if data.dtype == np.float32
                  else np.finfo(data.dtype).min, dtype=data.dtype)
    if data.shape[0] == 0:
        return out
    seg_ids = np.asarray(seg_ids).astype(np.int64)
    order = np.argsort(seg_ids, kind='stable')
    s = seg_ids[order]
    starts = np.flatnonzero(np.concatenate([[True], s[1:] != s[:-1]]))
    red = np.maximum.reduceat(data[order], starts, axis=0)
    out[s[starts]] = red
    return out


# ------------------------------------------------------------------ layers
CHUNK_ROWS = 1 << 18


def _edge_mlp_scatter_max(gather, layers, dst, num_segments, dtype):
    """max over segments of MLP(gather(rows)) evaluated in row chunks, so a
    full-size frame ([E,300] float64 activations) fits in host memory.  max is
    exact and order-free: chunking changes nothing in the result."""
    n = dst.shape[0]
    if n <= CHUNK_ROWS:
        return scatter_max(multi_layer_neural_network(
            gather(slice(0, n)), layers, is_logits=False), dst, num_segments)
    out = None
    for lo in range(0, n, CHUNK_ROWS):
        sl = slice(lo, min(n, lo + CHUNK_ROWS))
        part = scatter_max(multi_layer_neural_network(
            gather(sl), layers, is_logits=False), dst[sl], num_segments)
        out = part if out is None else np.maximum(out, part)
    return out


def _layers(params, scope, dtype):
    """Collect (W, b) of `scope/fully_connected[_i]` in creation order."""
    out = []
    i = 0
    while True:
        name = scope + '/fully_connected' + ('' if i == 0 else '_%d' % i)
        if name + '/weights' not in params:
            break
        if name + '/biases' in params:
            b = params[name + '/biases'].astype(dtype)
        else:   # a batch-norm normalizer (see fully_connected)
            bn = name + '/BatchNorm/'
            b = (params[bn + 'moving_mean'].astype(dtype),
                 params[bn + 'moving_variance'].astype(dtype),
                 params[bn + 'beta'].astype(dtype)
                 if bn + 'beta' in params else None,
                 params[bn + 'gamma'].astype(dtype)
                 if bn + 'gamma' in params else None)
        out.append((params[name + '/weights'].astype(dtype), b))
        i += 1
    return out


def point_set_pooling(params, scope, point_features, point_coordinates,
                      keypoint_indices, set_indices, dtype=np.float32):
    """PointSetPooling.apply_regular, gnn.py:222-283."""
    src = set_indices[:, 0].astype(np.int64)
    dst = set_indices[:, 1].astype(np.int64)
    pf_all = point_features.astype(dtype)
    pc_all = point_coordinates.astype(dtype)
    kp_all = keypoint_indices.reshape(-1).astype(np.int64)

    def gather(sl):
        pf = pf_all[src[sl]]
        pc = pc_all[src[sl]]
        kc = pc_all[kp_all[dst[sl]]]
        return np.concatenate([pf, pc - kc], axis=-1)
    agg = _edge_mlp_scatter_max(
        gather, _layers(params, scope + '/extract_vertex_features', dtype),
        dst, keypoint_indices.shape[0], dtype)
    return multi_layer_neural_network(
        agg, _layers(params, scope + '/combined_features', dtype),
        is_logits=False)


def graphnet_auto_center(params, scope, vertex_features, vertex_coordinates,
                         edges, auto_offset=True, dtype=np.float32,
                         return_intermediates=False):
    """GraphNetAutoCenter.apply_regular, gnn.py:298-373."""
    h = vertex_features.astype(dtype)
    x = vertex_coordinates.astype(dtype)
    src = edges[:, 0].astype(np.int64)
    dst = edges[:, 1].astype(np.int64)
    x_src = x          # source coordinates are gathered BEFORE the offset
    if auto_offset:    # (gnn.py:338-346)
        offset = multi_layer_neural_network(h, _layers(params, scope, dtype),
                                            is_logits=True)
        x = x + offset

    def gather(sl):
        return np.concatenate([h[src[sl]], x_src[src[sl]] - x[dst[sl]]],
                              axis=-1)
    agg = _edge_mlp_scatter_max(
        gather, _layers(params, scope + '/extract_vertex_features', dtype),
        dst, h.shape[0], dtype)
    upd = multi_layer_neural_network(
        agg, _layers(params, scope + '/combined_features', dtype),
        is_logits=True)
    out = upd + h
    if return_intermediates:
        return out, dict(offset_coords=x, aggregated=agg)
    return out


def class_aware_predictor(params, scope, features, num_classes,
                          dtype=np.float32):
    """ClassAwarePredictor.apply_regular, gnn.py:133-163 with the
    `classaware_predictor` MLP shapes of models.py:60-64."""
    f = features.astype(dtype)
    logits = multi_layer_neural_network(
        f, _layers(params, scope + '/predictor/cls', dtype), is_logits=True)
    boxes = []
    for c in range(num_classes):
        b = multi_layer_neural_network(
            f, _layers(params, scope + '/predictor/loc/cls_%d' % c, dtype),
            is_logits=True)
        boxes.append(b[:, None, :])
    return logits, np.concatenate(boxes, axis=1)


def predict(params, config, initial_vertex_features, vertex_coord_list,
            keypoint_indices_list, edges_list, dtype=np.float32,
            return_features=False):
    """MultiLayerFastLocalGraphModelV2.predict, models.py:79-163."""
    layer_configs = config['model_kwargs']['layer_configs']
    feats = initial_vertex_features.astype(dtype)
    feats_list = [feats]
    for lc in layer_configs[:-1]:
        lvl = lc['graph_level']
        if lc['type'] == 'scatter_max_point_set_pooling':
            feats = point_set_pooling(
                params, lc['scope'], feats, vertex_coord_list[lvl],
                keypoint_indices_list[lvl], edges_list[lvl], dtype)
        elif lc['type'] == 'scatter_max_graph_auto_center_net':
            feats = graphnet_auto_center(
                params, lc['scope'], feats, vertex_coord_list[lvl],
                edges_list[lvl], auto_offset=lc['kwargs']['auto_offset'],
                dtype=dtype)
        else:
            raise NotImplementedError(lc['type'])
        feats_list.append(feats)
    pc = layer_configs[-1]
    assert pc['type'] == 'classaware_predictor'
    logits, boxes = class_aware_predictor(params, pc['scope'], feats,
                                          config['num_classes'], dtype)
    if return_features:
        return logits, boxes, feats_list
    return logits, boxes


def softmax(logits):
    """models.py:165-168."""
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)
