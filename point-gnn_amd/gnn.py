"""GNN operators -- the reference's `models/gnn.py` surface (same class names,
constructor/`apply_regular` signatures and kwargs) executing on the fused HIP
kernels of csrc/gnn.hip through the ctypes C-ABI.

TF1 builds a graph once and finds variables through `tf.variable_scope`; this
implementation is eager, so the two implicit TF mechanisms are explicit here:

  * `parameters(store)`  -- context manager binding a `ParamStore` (a state
    dict keyed by the reference's TF variable names);
  * `variable_scope(name)` -- context manager mirroring tf.variable_scope.

    with gnn.parameters(store), gnn.variable_scope('layer2'):
        h = gnn.GraphNetAutoCenter().apply_regular(h, xyz, None, edges, **kw)

Only what every shipped config uses has a device path: activation 'ReLU',
normalization 'NONE', scatter-max aggregation (other registry keys of
gnn.py:17-32 raise NotImplementedError).  Tensors are torch CUDA float32 /
int32; activations are kept zero-padded to a multiple of 16 columns
(`padded_width`) between operators -- `.features(t, width)` strips the pad.
"""
import contextlib
import ctypes
import threading
from functools import partial

import numpy as np
import torch

from . import _lib
from .weights import mlp_names

__all__ = ["PointSetPooling", "GraphNetAutoCenter", "ClassAwarePredictor",
           "fuse_vertex_stages",
           "multi_layer_neural_network_fn", "multi_layer_fc_fn",
           "graph_scatter_max_fn", "ParamStore", "parameters",
           "variable_scope", "padded_width"]


def padded_width(n):
    return (int(n) + 15) // 16 * 16


# --------------------------------------------------------------------------
# parameter store / scopes
# --------------------------------------------------------------------------
class ParamStore(object):
    """Holds the model's variables (NumPy, reference TF names) and the
    device-side packed images derived from them (built lazily, cached)."""

    def __init__(self, params, device=None):
        self.params = {k: np.asarray(v) for k, v in params.items()}
        self.device = device
        self._cache = {}
        self._edge_arith = 'f32'

    @property
    def edge_arith(self):
        """Arithmetic of GraphNetAutoCenter's per-edge product: EDGE_ARITHS."""
        return self._edge_arith

    @edge_arith.setter
    def edge_arith(self, value):
        if value not in EDGE_ARITHS:
            raise ValueError("edge_arith must be one of %r, not %r"
                             % (EDGE_ARITHS, value))
        self._edge_arith = value

    def range_status(self):
        """Device int32 the 'f16x2' edge kernel raises when an activation left
        fp16's comfortable range (>= 32768); zeroed when created / read."""
        if getattr(self, '_range_status', None) is None:
            self._range_status = torch.zeros(1, dtype=torch.int32,
                                             device=self._dev())
        return self._range_status

    def edge_range_ok(self):
        """Reads (and clears) the flag: one small device-to-host read.  True
        when every 'f16x2' edge stage since the last call stayed in range."""
        st = getattr(self, '_range_status', None)
        if st is None:
            return True
        ok = int(st.item()) == 0
        if not ok:
            st.zero_()
        return ok

    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise _lib.PointGnnHipError(
                    "pointgnn_amd needs a GPU: there is no CPU fallback")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def has(self, name):
        return name + '/weights' in self.params

    def fc(self, name):
        """(W [in,out], b [out]) of one slim.fully_connected.  A layer built
        with a batch-norm normalizer (gnn.py:17-23: 'fused_BN_center', 'BN',
        'BN_center' -- slim then creates no biases but
        `BatchNorm/{beta,moving_mean,moving_variance}`, gamma only with
        scale=True) is returned FOLDED: at inference slim.batch_norm is the
        per-column affine map (y - mean) / sqrt(var + 0.001) [* gamma] [+ beta]
        of y = x W, i.e. the layer x W' + b' with W' = W s, b' = beta - mean s
        (s in float64, rounded once)."""
        w = self.params[name + '/weights']
        if name + '/biases' in self.params:
            return (w.astype(np.float32),
                    self.params[name + '/biases'].astype(np.float32))
        bn = name + '/BatchNorm/'
        if bn + 'moving_mean' not in self.params:
            raise KeyError("%s: neither biases nor BatchNorm statistics"
                           % name)
        mean = self.params[bn + 'moving_mean'].astype(np.float64)
        var = self.params[bn + 'moving_variance'].astype(np.float64)
        s = 1.0 / np.sqrt(var + BATCH_NORM_EPSILON)
        if bn + 'gamma' in self.params:
            s = s * self.params[bn + 'gamma'].astype(np.float64)
        beta = self.params[bn + 'beta'].astype(np.float64) \
            if bn + 'beta' in self.params else 0.0
        return ((w.astype(np.float64) * s).astype(np.float32),
                (beta - mean * s).astype(np.float32))

    def mlp(self, scope, n_layers):
        return [self.fc(n) for n in mlp_names(scope, n_layers)]

    def pack(self, w, b):
        """[k_in, n_out] weights + bias -> device tensor in MFMA fragment order
        (pgnn_pack_fc)."""
        lib = _lib.load()
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        k_in, n_out = w.shape
        host = np.empty(lib.pgnn_packed_fc_floats(k_in, n_out), np.float32)
        _lib.check(lib.pgnn_pack_fc(w.ctypes.data, b.ctypes.data, k_in, n_out,
                                    host.ctypes.data), "pgnn_pack_fc")
        return torch.from_numpy(host).to(self._dev())

    def cached(self, key, builder):
        if key not in self._cache:
            self._cache[key] = builder()
        return self._cache[key]

    def invalidate(self):
        self._cache.clear()


class Chain(object):
    """A packed MLP chain: keeps the device buffers alive next to the ctypes
    array of `pgnn_fc_layer` descriptors that points at them."""

    def __init__(self, store, layers):
        """layers: [(W [k,n], b [n], relu_from)]"""
        self.buffers = []
        self.array = (_lib.FcLayer * len(layers))()
        self.n = len(layers)
        for i, (w, b, relu_from) in enumerate(layers):
            buf = store.pack(w, b)
            self.buffers.append(buf)
            self.array[i].packed = buf.data_ptr()
            self.array[i].k_in = int(w.shape[0])
            self.array[i].n_out = int(w.shape[1])
            self.array[i].relu_from = int(relu_from)
        self.k_in = int(layers[0][0].shape[0])
        self.n_out = int(layers[-1][0].shape[1])


class _State(threading.local):
    """The bound ParamStore and the variable-scope stack, PER THREAD: two
    Python threads driving operators (the frame pipeline's builder thread, a
    serving front end) each see their own `parameters()` / `variable_scope()`
    nesting, like tf.variable_scope's thread-local stack."""

    def __init__(self):
        self.store = None
        self.scope = []
        self.fuse = False      # fuse_vertex_stages() active
        self.pending = None    # an operator's deferred per-vertex tail
        self.taken = None      # ... taken over by its consumer, not launched yet


_state = _State()


@contextlib.contextmanager
def parameters(store):
    prev = _state.store
    _state.store = store
    try:
        yield store
    finally:
        _state.store = prev


@contextlib.contextmanager
def variable_scope(name):
    _state.scope.append(name)
    try:
        yield
    finally:
        _state.scope.pop()


def _scope(*suffix):
    return '/'.join(list(_state.scope) + list(suffix))


# --------------------------------------------------------------------------
# fusing the per-vertex stages across operator boundaries
# --------------------------------------------------------------------------
class _Pending(object):
    """The tail of an operator that has not been launched yet:
    y = chain(x[:, :nx]) (+ residual).  `y` is allocated (and already returned
    to the caller); the operator that receives it as its input launches the
    tail TOGETHER with its own per-vertex head
    (pgnn_vertex_update_pre_edge_fwd / pgnn_mlp2_fwd), anything else launches
    it on its own (`_flush_pending`)."""
    __slots__ = ('chain', 'x', 'nx', 'residual', 'count', 'y')


@contextlib.contextmanager
def fuse_vertex_stages(enabled=True):
    """Inside this context PointSetPooling / GraphNetAutoCenter do not launch
    their last per-vertex MLP (output MLP; update MLP + residual) themselves:
    it runs in the same launch as the per-vertex head of the operator that
    consumes the result (the next GraphNetAutoCenter's offset MLP / Q / P, the
    predictor's heads) -- half the K-row launches of a frame, and h stays in
    LDS between the two.  Results are bit-identical.  The returned tensor is
    written by that later launch: a caller that hands it to anything but the
    next operator must leave the context first (it flushes on exit).
    models.MultiLayerFastLocalGraphModelV2.predict wraps its layer loop in
    it."""
    prev = _state.fuse
    _flush_pending()
    _state.fuse = bool(enabled)
    try:
        yield
    finally:
        _flush_pending()
        _state.fuse = prev


def _flush_pending():
    """Launches, on its own, whatever tail has not run yet: the parked one and
    one an operator took (`_take_pending`) but did not get to launch -- it
    raised in between; its output tensor is already in the caller's hands."""
    for slot in ('taken', 'pending'):
        p = getattr(_state, slot)
        setattr(_state, slot, None)
        if p is not None:
            mlp_forward(p.chain, p.x, p.nx, residual=p.residual, count=p.count,
                        out=p.y)


def _take_pending(t):
    """The deferred tail whose output IS `t` (else None, after launching
    whatever was pending on its own).  The taker calls `_tail_launched()` once
    the tail has run (inside its own launch or separately)."""
    p = _state.pending
    if p is not None and p.y is t and _state.taken is None:
        _state.pending = None
        _state.taken = p
        return p
    _flush_pending()
    return None


def _tail_launched():
    _state.taken = None


def _finish_rows(chain, x, nx, residual=None, count=None):
    """An operator's last per-vertex MLP: launched now, or deferred to the
    consumer inside fuse_vertex_stages()."""
    if not _state.fuse:
        return mlp_forward(chain, x, nx, residual=residual, count=count)
    _flush_pending()
    if count is None:
        count = _lib.count_of(x)
    p = _Pending()
    p.chain, p.x, p.nx, p.residual, p.count = chain, _as_f32(x), int(nx), \
        residual, count
    p.y = torch.empty((int(x.shape[0]), padded_width(chain.n_out)),
                      dtype=torch.float32, device=x.device)
    if count is not None:
        _lib.tag_count(p.y, count)
    _state.pending = p
    return p.y


def _store():
    if _state.store is None:
        raise RuntimeError("no ParamStore bound: wrap the call in "
                           "`with pointgnn_amd.gnn.parameters(store):`")
    return _state.store


# slim.batch_norm's default epsilon (the reference never overrides it)
BATCH_NORM_EPSILON = 0.001
# normalization_fn_dict keys (gnn.py:17-23) these INFERENCE operators accept:
# the batch-norm kinds run on their moving statistics (is_training=False), a
# per-column affine map that ParamStore.fc folds into the layer.  'IN'
# (instance_normalization, gnn.py:9-15) takes its moments over all rows of the
# tensor at run time -- a reduction over every edge between two layers of a
# fused stage -- and has no device path; training with batch statistics has
# none either (train.Trainer refuses such configs).
FOLDED_NORMALIZATIONS = ('fused_BN_center', 'BN', 'BN_center')


def _check_kinds(activation_type, normalization_type):
    if normalization_type not in ('NONE', None) + FOLDED_NORMALIZATIONS:
        raise NotImplementedError(
            "normalization %r: 'NONE' (what every shipped config uses) and "
            "the batch-norm kinds at inference have a device path"
            % (normalization_type,))
    if activation_type != 'ReLU':
        raise NotImplementedError(
            "activation %r: only 'ReLU' has a device path" % (activation_type,))


def _relu_chain(store, scope, widths, is_logits):
    """Chain for multi_layer_neural_network_fn (gnn.py:86-104)."""
    def build():
        fcs = store.mlp(scope, len(widths))
        layers = []
        for i, (w, b) in enumerate(fcs):
            if w.shape[1] != widths[i]:
                raise ValueError("%s: layer %d has width %d, config says %d"
                                 % (scope, i, w.shape[1], widths[i]))
            linear = is_logits and i == len(fcs) - 1
            layers.append((w, b, w.shape[1] if linear else 0))
        return Chain(store, layers)
    return store.cached(('mlp', scope, tuple(widths), bool(is_logits)), build)


def _as_f32(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


def _as_i32(t):
    if t.dtype != torch.int32 or not t.is_contiguous():
        t = t.to(torch.int32).contiguous()
    return t


def mlp_forward(chain, x, nx, x2=None, nx2=0, residual=None, count=None,
                out=None):
    """y = chain(concat(x[:, :nx], x2[:, :nx2])) (+ residual); returns a
    [rows, padded_width(n_out)] tensor (pad columns are zero).  `count` (a
    _lib.DeviceCount; default: the one `x` is tagged with): capacity form --
    x has capacity rows, the count is on the device, the result is tagged."""
    lib = _lib.load()
    if count is None:
        count = _lib.count_of(x)
    x = _as_f32(x)
    rows = int(x.shape[0])
    out_w = padded_width(chain.n_out)
    y = out if out is not None else \
        torch.empty((rows, out_w), dtype=torch.float32, device=x.device)
    if x2 is not None:
        x2 = _as_f32(x2)
    if residual is not None:
        residual = _as_f32(residual)
        assert residual.shape[1] >= out_w
    args = (_lib.ptr(x), x.stride(0), int(nx), _lib.ptr(x2),
            x2.stride(0) if x2 is not None else 0, int(nx2), rows, chain.array,
            chain.n, _lib.ptr(residual),
            residual.stride(0) if residual is not None else 0, _lib.ptr(y),
            y.stride(0))
    if count is None:
        _lib.check(lib.pgnn_mlp_fwd(*args, _lib.stream_ptr()), "pgnn_mlp_fwd")
        return y
    _lib.check(lib.pgnn_mlp_fwd_dyn(*args, count.arg(), _lib.stream_ptr()),
               "pgnn_mlp_fwd_dyn")
    return _lib.tag_count(y, count)


# --------------------------------------------------------------------------
# function-style API of the reference (eager equivalents)
# --------------------------------------------------------------------------
def multi_layer_neural_network_fn(features, Ks=(64, 32, 64), is_logits=False,
                                  normalization_type="fused_BN_center",
                                  activation_type='ReLU'):
    """gnn.py:86-104.  Variables are `<scope>/fully_connected[_i]` of the
    current variable scope.  Returns [rows, padded_width(Ks[-1])]."""
    _check_kinds(activation_type, normalization_type)
    assert features.dim() == 2
    _flush_pending()
    store = _store()
    chain = _relu_chain(store, _scope(), list(Ks), is_logits)
    return mlp_forward(chain, features, chain.k_in)


def multi_layer_fc_fn(sv, mask=None, Ks=(64, 32, 64), num_classes=4,
                      is_logits=False, num_layer=4,
                      normalization_type="fused_BN_center",
                      activation_type='ReLU'):
    """gnn.py:34-84: Ks hidden layers then a `num_classes`-wide layer (linear
    when is_logits)."""
    assert sv.dim() == 2
    assert len(Ks) == num_layer - 1
    out = multi_layer_neural_network_fn(
        sv, Ks=list(Ks) + [num_classes], is_logits=is_logits,
        normalization_type=normalization_type,
        activation_type=activation_type)
    if mask is not None:
        out = out * mask
    return out


def graph_scatter_max_fn(point_features, point_centers, num_centers,
                         ids_sorted=False):
    """gnn.py:106-109 = tf.math.unsorted_segment_max.  Standalone kernel
    (csrc/scatter_max.hip); the layers below use the fused epilogue instead."""
    _flush_pending()   # (a deferred tail may be what writes point_features)
    lib = _lib.load()
    data = _as_f32(point_features)
    ids = _as_i32(point_centers.reshape(-1))
    n_rows, n_cols = int(data.shape[0]), int(data.shape[1])
    out = torch.empty((int(num_centers), n_cols), dtype=torch.float32,
                      device=data.device)
    _lib.check(lib.pgnn_scatter_max_f32(
        _lib.ptr(data), data.stride(0) if n_rows else n_cols, _lib.ptr(ids),
        n_rows, n_cols, int(num_centers), _lib.ptr(out), n_cols,
        1 if ids_sorted else 0, _lib.stream_ptr()), "pgnn_scatter_max_f32")
    return out


def _scatter_sum(point_features, point_centers, num_centers, mean):
    _flush_pending()   # (a deferred tail may be what writes point_features)
    lib = _lib.load()
    data = _as_f32(point_features)
    ids = _as_i32(point_centers.reshape(-1))
    n_rows, n_cols = int(data.shape[0]), int(data.shape[1])
    out = torch.empty((int(num_centers), n_cols), dtype=torch.float32,
                      device=data.device)
    counts = torch.empty((int(num_centers),), dtype=torch.int32,
                         device=data.device) if mean else None
    _lib.check(lib.pgnn_scatter_sum_f32(
        _lib.ptr(data), data.stride(0) if n_rows else n_cols, _lib.ptr(ids),
        n_rows, n_cols, int(num_centers), _lib.ptr(out), n_cols,
        1 if mean else 0, _lib.ptr(counts), _lib.stream_ptr()),
        "pgnn_scatter_sum_f32")
    return out


def graph_scatter_sum_fn(point_features, point_centers, num_centers):
    """gnn.py:111-114 = tf.math.unsorted_segment_sum (standalone op; the fused
    layers implement scatter-max only, like every shipped config)."""
    return _scatter_sum(point_features, point_centers, num_centers, False)


def graph_scatter_mean_fn(point_features, point_centers, num_centers):
    """gnn.py:116-119 = tf.math.unsorted_segment_mean."""
    return _scatter_sum(point_features, point_centers, num_centers, True)


def _edges_sorted_flag(edges):
    """Edges produced by pointgnn_amd.graph_gen are grouped by ascending
    destination; foreign edge lists are checked once (cheap device reduction,
    one sync) so that unsorted input takes the all-atomic path."""
    cached = getattr(edges, "_pgnn_sorted", None)
    if cached is not None:
        return cached
    if edges.shape[0] < 2:
        return 1
    d = edges[:, 1]
    flag = 1 if bool((d[1:] >= d[:-1]).all().item()) else 0
    mark_sorted(edges, flag)
    return flag


def mark_sorted(edges, flag=1):
    """Tag an edge tensor as grouped by ascending destination (skips the
    check in the layers)."""
    try:
        edges._pgnn_sorted = int(flag)
    except Exception:
        pass
    return edges


def _both_counts(cnt_a, n_a, cnt_b, n_b, device):
    """A *_dyn entry takes both of its sizes from the device: the one the
    host does know becomes a one-element device constant."""
    def const(n):
        t = torch.full((1,), int(n), dtype=torch.int32, device=device)
        return _lib.DeviceCount(t, int(n))
    return (cnt_a if cnt_a is not None else const(n_a),
            cnt_b if cnt_b is not None else const(n_b))


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
class PointSetPooling(object):
    """gnn.py:211-283."""

    def __init__(self, point_feature_fn=multi_layer_neural_network_fn,
                 aggregation_fn=graph_scatter_max_fn,
                 output_fn=multi_layer_neural_network_fn):
        if aggregation_fn is not graph_scatter_max_fn:
            raise NotImplementedError("only scatter-max aggregation is fused")
        self._point_feature_fn = point_feature_fn
        self._aggregation_fn = aggregation_fn
        self._output_fn = output_fn

    def apply_regular(self, point_features, point_coordinates,
                      keypoint_indices, set_indices,
                      point_MLP_depth_list=None,
                      point_MLP_normalization_type='fused_BN_center',
                      point_MLP_activation_type='ReLU',
                      output_MLP_depth_list=None,
                      output_MLP_normalization_type='fused_BN_center',
                      output_MLP_activation_type='ReLU'):
        """point_features [N,M], point_coordinates [N,3], keypoint_indices
        [K,1], set_indices [S,2] (point, set) -> [K, padded_width(out)]."""
        _check_kinds(point_MLP_activation_type, point_MLP_normalization_type)
        _check_kinds(output_MLP_activation_type, output_MLP_normalization_type)
        _flush_pending()
        lib = _lib.load()
        store = _store()
        with variable_scope('extract_vertex_features'):
            point_chain = _relu_chain(store, _scope(),
                                      list(point_MLP_depth_list), False)
        feats = _as_f32(point_features)
        xyz = _as_f32(point_coordinates)
        # capacity form (graph_gen's deferred_counts): K and the edge count
        # are device-side, the tensors' leading dimensions are capacities
        cnt_k = _lib.count_of(keypoint_indices)
        cnt_e = _lib.count_of(set_indices)
        kp = _as_i32(keypoint_indices.reshape(-1))
        edges = _as_i32(set_indices)
        n_feat = int(feats.shape[1])
        if n_feat != point_chain.k_in - 3 and \
                n_feat == padded_width(point_chain.k_in - 3):
            # (the zero-padded rows of a previous operator: a pooling level
            # above the first)
            n_feat = point_chain.k_in - 3
        assert point_chain.k_in == n_feat + 3, \
            "point MLP expects %d inputs" % point_chain.k_in
        k = int(kp.shape[0])
        if n_feat > 13:
            # wider than the fused kernel's gather (the previous level's
            # features; no shipped config): edge rows, row MLP, scatter-max
            if cnt_k is not None or cnt_e is not None:
                raise NotImplementedError(
                    "PointSetPooling of wide features in capacity form")
            n_e = int(edges.shape[0])
            rows_in = torch.empty((n_e, padded_width(n_feat + 3)),
                                  dtype=torch.float32, device=xyz.device)
            _lib.check(lib.pgnn_pool_features_wide_fwd(
                _lib.ptr(feats), feats.stride(0), n_feat, _lib.ptr(xyz),
                _lib.ptr(kp), _lib.ptr(edges), n_e, _lib.ptr(rows_in),
                rows_in.stride(0), _lib.stream_ptr()),
                "pgnn_pool_features_wide_fwd")
            rows = mlp_forward(point_chain, rows_in, point_chain.k_in)
            agg = graph_scatter_max_fn(
                rows, edges[:, 1], k,
                ids_sorted=_edges_sorted_flag(set_indices) == 1)
            with variable_scope('combined_features'):
                out_chain = _relu_chain(store, _scope(),
                                        list(output_MLP_depth_list), False)
            return _finish_rows(out_chain, agg, point_chain.n_out)
        if int(feats.shape[1]) != n_feat or feats.stride(0) != n_feat:
            # the fused kernel gathers feat[s * n_feat + i]: a narrow level
            # above the first hands over zero-padded [N, 16] rows
            feats = feats[:, :n_feat].contiguous()
        agg = torch.empty((k, padded_width(point_chain.n_out)),
                          dtype=torch.float32, device=xyz.device)
        args = (_lib.ptr(feats), n_feat, _lib.ptr(xyz), _lib.ptr(kp),
                _lib.ptr(edges), int(edges.shape[0]), k, point_chain.array,
                point_chain.n, _edges_sorted_flag(set_indices), _lib.ptr(agg),
                agg.stride(0), _lib.ptr(_lib.sched_ws(xyz.device)))
        if cnt_k is not None or cnt_e is not None:
            cnt_k, cnt_e = _both_counts(cnt_k, k, cnt_e, int(edges.shape[0]),
                                        xyz.device)
        done = False
        if store.edge_arith == 'f16x2' and point_chain.n == 4:
            # the model's 16-bit arithmetic covers this stage's wide layer too
            # (csrc/pool_ws_f16.h); other shapes / few edges: the fp32 kernel
            with variable_scope('extract_vertex_features'):
                pscope = _scope()

            def build(li=-1):
                w, b = store.mlp(pscope, point_chain.n)[li]
                w = np.ascontiguousarray(w, dtype=np.float32)
                b = np.ascontiguousarray(b, dtype=np.float32)
                host = np.empty(lib.pgnn_packed_fc_f16x2_bytes(*w.shape),
                                np.uint8)
                _lib.check(lib.pgnn_pack_fc_f16x2_acc(
                    w.ctypes.data, b.ctypes.data, w.shape[0], w.shape[1],
                    host.ctypes.data), "pgnn_pack_fc_f16x2_acc")
                return torch.from_numpy(host).to(store._dev())
            try:
                image = store.cached(
                    ('pool_f16x2', pscope, tuple(point_MLP_depth_list)), build)
                hidden = store.cached(
                    ('pool_f16x2_h', pscope, tuple(point_MLP_depth_list)),
                    lambda: build(-2))
            except _lib.PointGnnHipError as err:
                if "code %d" % _lib.E_UNSUPPORTED not in str(err):
                    raise
                image = None     # a weight outside fp16's range
            if image is not None:
                rc = lib.pgnn_point_set_pooling_f16x2_fwd(
                    *(args[:9] + (_lib.ptr(image), _lib.ptr(hidden)) +
                      args[9:] +
                      (_lib.ptr(store.range_status()),
                       cnt_e.arg() if cnt_e is not None else None,
                       cnt_k.arg() if cnt_k is not None else None,
                       _lib.stream_ptr())))
                if rc != _lib.E_UNSUPPORTED:
                    _lib.check(rc, "pgnn_point_set_pooling_f16x2_fwd")
                    done = True
        ws_bytes = ctypes.c_size_t(0)
        if not done:
            # chains whose last layer does not fit one CU's LDS (ped_cyl's
            # 4-32-64-128-256-512) run as two launches through a workspace of
            # hidden rows (csrc/pool_split.h); everything else asks for none
            _lib.check(lib.pgnn_point_set_pooling_workspace_bytes(
                point_chain.array, point_chain.n, n_feat, int(edges.shape[0]),
                int(cnt_e.hint) if cnt_e is not None else 0,
                ctypes.byref(ws_bytes)),
                "pgnn_point_set_pooling_workspace_bytes")
        if done:
            pass
        elif ws_bytes.value:
            # (a stream-ordered allocation of the caching allocator: the next
            # frame on this stream gets the same block back)
            work = torch.empty(ws_bytes.value // 4, dtype=torch.float32,
                               device=xyz.device)
            _lib.check(lib.pgnn_point_set_pooling_fwd_ws(
                *args, cnt_e.arg() if cnt_e is not None else None,
                cnt_k.arg() if cnt_k is not None else None, _lib.ptr(work),
                ws_bytes.value, _lib.stream_ptr()),
                "pgnn_point_set_pooling_fwd_ws")
        elif cnt_k is None and cnt_e is None:
            _lib.check(lib.pgnn_point_set_pooling_fwd(
                *args, _lib.stream_ptr()), "pgnn_point_set_pooling_fwd")
        else:
            _lib.check(lib.pgnn_point_set_pooling_fwd_dyn(
                *args, cnt_e.arg(), cnt_k.arg(), _lib.stream_ptr()),
                "pgnn_point_set_pooling_fwd_dyn")
        with variable_scope('combined_features'):
            out_chain = _relu_chain(store, _scope(),
                                    list(output_MLP_depth_list), False)
        return _finish_rows(out_chain, agg, point_chain.n_out, count=cnt_k)


# when set to a list, every GraphNetAutoCenter call appends its (P, Q) per-vertex
# tensors: bench.py times the edge kernel on a frame's real inputs with it
EDGE_INPUT_TAP = None

# Arithmetic of the per-edge 300x300 / 256x256 product (gnn.py:355-365), an
# attribute of the MODEL (`model.edge_arith`, `InferenceEngine(...,
# edge_arith=)`; it travels with the model's ParamStore, so two models in one
# process -- or frames of two engines in flight on different streams -- do not
# share a switch):
#   'f32'    -- fp32 MFMA (default; the parity reference, bit-identical between
#               its kernel variants);
#   'bf16x3' -- SECONDARY: both operands split exactly into three bf16 parts,
#               six bf16 MFMAs per block accumulated in fp32
#               (csrc/edge_ws_bf16.h, pgnn_edge_mlp_scatter_max_bf16x3_fwd):
#               agrees with 'f32' to fp32 rounding noise, ~1.7x faster.  Falls
#               back to 'f32' where the kernel does not apply (few edges,
#               other layer shapes).
#   'f16x2'  -- SECONDARY: both operands as TWO fp16 values (22 significand
#               bits, round to nearest), three fp16 MFMAs per block
#               (csrc/edge_ws_f16.h, pgnn_edge_mlp_scatter_max_f16x2_fwd):
#               not exact -- the stage's distance to float64 grows by a few per
#               cent -- ~3x faster than 'f32'.  Activations are clamped at
#               65504; the kernel flags frames in which one could reach 32768 and
#               `model.edge_range_ok()` (read by the engine with a frame's
#               results) reports it: rerun such a frame in 'f32'.  Also covers
#               the wide last layer of PointSetPooling's point MLP
#               (csrc/pool_ws_f16.h, pgnn_point_set_pooling_f16x2_fwd; car's
#               4-32-64-128-300 chain), same representation, same guard.
EDGE_ARITHS = ('f32', 'bf16x3', 'f16x2')


class GraphNetAutoCenter(object):
    """gnn.py:285-373."""

    def __init__(self, edge_feature_fn=multi_layer_neural_network_fn,
                 aggregation_fn=graph_scatter_max_fn,
                 update_fn=multi_layer_neural_network_fn,
                 auto_offset_fn=multi_layer_neural_network_fn):
        if aggregation_fn is not graph_scatter_max_fn:
            raise NotImplementedError("only scatter-max aggregation is fused")
        self._edge_feature_fn = edge_feature_fn
        self._aggregation_fn = aggregation_fn
        self._update_fn = update_fn
        self._auto_offset_fn = auto_offset_fn

    def apply_regular(self, input_vertex_features, input_vertex_coordinates,
                      NOT_USED, edges, edge_MLP_depth_list=None,
                      edge_MLP_normalization_type='fused_BN_center',
                      edge_MLP_activation_type='ReLU',
                      update_MLP_depth_list=None,
                      update_MLP_normalization_type='fused_BN_center',
                      update_MLP_activation_type='ReLU', auto_offset=False,
                      auto_offset_MLP_depth_list=None,
                      auto_offset_MLP_normalization_type='fused_BN_center',
                      auto_offset_MLP_feature_activation_type='ReLU'):
        """h [K, >=C] (zero-padded), x [K,3], edges [E,2] (src, dst) ->
        [K, padded_width(C)]."""
        _check_kinds(edge_MLP_activation_type, edge_MLP_normalization_type)
        _check_kinds(update_MLP_activation_type, update_MLP_normalization_type)
        # the producer's deferred tail (fuse_vertex_stages): launched together
        # with this operator's per-vertex head below
        pend = _take_pending(input_vertex_features)
        lib = _lib.load()
        store = _store()
        scope = _scope()
        st = _lib.stream_ptr()
        cnt_k = _lib.count_of(input_vertex_features)
        if cnt_k is None:
            cnt_k = _lib.count_of(input_vertex_coordinates)
        cnt_e = _lib.count_of(edges)
        h = _as_f32(input_vertex_features)
        x = _as_f32(input_vertex_coordinates)
        e = _as_i32(edges)
        k = int(h.shape[0])
        if cnt_k is not None or cnt_e is not None:   # capacity form
            cnt_k, cnt_e = _both_counts(cnt_k, k, cnt_e, int(e.shape[0]),
                                        h.device)

        edge_widths = list(edge_MLP_depth_list)
        edge_scope = scope + '/extract_vertex_features'

        def build_edge():
            fcs = store.mlp(edge_scope, len(edge_widths))
            w1, b1 = fcs[0]
            c = w1.shape[0] - 3
            # vertex-side image of the first edge layer: P = [h, x] @ W1 + b1
            p_chain = Chain(store, [(w1, b1, w1.shape[1])])
            wq = padded_width(w1.shape[1])
            wx = np.zeros((3, wq), np.float32)
            wx[:, :w1.shape[1]] = w1[c:c + 3]
            wx_dev = torch.from_numpy(wx).to(store._dev())
            if len(fcs) > 1:
                rest = Chain(store, [(w, b, 0) for (w, b) in fcs[1:]])
            else:
                rest = None
            return c, p_chain, wx_dev, rest
        c, p_chain, wx_dev, rest = store.cached(
            ('edge', edge_scope, tuple(edge_widths)), build_edge)
        if rest is None:
            raise NotImplementedError("edge MLP needs at least two layers")
        assert h.shape[1] >= c, "vertex features narrower than edge MLP input"

        # per vertex, one launch (gnn.py:341-356): [optional] coordinate offset
        # delta = MLP(h) (auto-registration), Q = (x + delta) @ W1[C:],
        # P = [h, x] @ W1 + b1, and the lowest() fill of the aggregation buffer
        off_chain = None
        if auto_offset:
            _check_kinds(auto_offset_MLP_feature_activation_type,
                         auto_offset_MLP_normalization_type)
            off_chain = _relu_chain(store, scope,
                                    list(auto_offset_MLP_depth_list), True)
            assert off_chain.k_in == c, "offset MLP input must be the features"
        wq = int(wx_dev.shape[1])
        q = torch.empty((k, wq), dtype=torch.float32, device=h.device)
        p = torch.empty((k, wq), dtype=torch.float32, device=h.device)
        agg = torch.empty((k, padded_width(rest.n_out)), dtype=torch.float32,
                          device=h.device)
        pre_args = (_lib.ptr(h), h.stride(0), c, _lib.ptr(x),
                    off_chain.array if off_chain is not None else None,
                    off_chain.n if off_chain is not None else 0, p_chain.array,
                    _lib.ptr(wx_dev), k, _lib.ptr(p), _lib.ptr(q), wq,
                    _lib.ptr(agg), agg.stride(0))
        if pend is not None:
            res = pend.residual
            front = (_lib.ptr(pend.x), pend.x.stride(0), pend.nx,
                     pend.chain.array, pend.chain.n, _lib.ptr(res),
                     res.stride(0) if res is not None else 0, _lib.ptr(h),
                     h.stride(0))
            if cnt_k is None:
                rc = lib.pgnn_vertex_update_pre_edge_fwd(
                    *(front + pre_args[2:]), st)
            else:
                rc = lib.pgnn_vertex_update_pre_edge_fwd_dyn(
                    *(front + pre_args[2:]), cnt_k.arg(), st)
            if rc == _lib.E_UNSUPPORTED:
                # too many rows / too wide for the one-launch form: the
                # producer's tail on its own, then the plain head below
                _flush_pending()
                pend = None
            else:
                _lib.check(rc, "pgnn_vertex_update_pre_edge_fwd")
                _tail_launched()
        if pend is not None:
            pass
        elif cnt_k is None:
            _lib.check(lib.pgnn_vertex_pre_edge_fwd(*pre_args, st),
                       "pgnn_vertex_pre_edge_fwd")
        else:
            _lib.check(lib.pgnn_vertex_pre_edge_fwd_dyn(
                *pre_args, cnt_k.arg(), st), "pgnn_vertex_pre_edge_fwd_dyn")
        if EDGE_INPUT_TAP is not None:   # measurement hook (bench.py)
            EDGE_INPUT_TAP.append((p, q))
        # per-edge: ReLU(P[src] - Q[dst]) -> remaining edge layers -> max
        edge_args = (_lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e),
                     int(e.shape[0]), k, rest.array, rest.n,
                     _edges_sorted_flag(edges) | 2, _lib.ptr(agg),
                     agg.stride(0), _lib.ptr(_lib.sched_ws(h.device)))
        done = False
        if store.edge_arith != 'f32' and rest.n == 1:
            done = self._edge_split(lib, store, edge_scope, edge_widths, p, q,
                                    wq, rest, e, k, edges, agg, cnt_e, cnt_k,
                                    st)
        if done:
            pass
        elif cnt_k is None:
            _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd(*edge_args, st),
                       "pgnn_edge_mlp_scatter_max_fwd")
        else:
            _lib.check(lib.pgnn_edge_mlp_scatter_max_fwd_dyn(
                *edge_args, cnt_e.arg(), cnt_k.arg(), st),
                "pgnn_edge_mlp_scatter_max_fwd_dyn")
        # update + residual, gnn.py:367-372
        upd_chain = _relu_chain(store, scope + '/combined_features',
                                list(update_MLP_depth_list), True)
        assert upd_chain.n_out == c, "update MLP must preserve the width"
        if h.shape[1] < padded_width(c):
            hp = torch.zeros((k, padded_width(c)), dtype=torch.float32,
                             device=h.device)
            hp[:, :h.shape[1]] = h
            h = hp
        return _finish_rows(upd_chain, agg, rest.n_out, residual=h,
                            count=cnt_k)


    @staticmethod
    def _edge_split(lib, store, edge_scope, edge_widths, p, q, wq, rest, e, k,
                    edges, agg, cnt_e, cnt_k, st):
        """The edge stage on the matrix pipe's 16-bit formats ('bf16x3' /
        'f16x2'); False (nothing done) where the kernel does not apply."""
        arith = store.edge_arith

        def build():
            w, b = store.mlp(edge_scope, len(edge_widths))[1]
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            nbytes = getattr(lib, "pgnn_packed_fc_%s_bytes" % arith)(*w.shape)
            host = np.empty(nbytes, np.uint8)
            name = "pgnn_pack_fc_%s" % arith
            _lib.check(getattr(lib, name)(
                w.ctypes.data, b.ctypes.data, w.shape[0], w.shape[1],
                host.ctypes.data), name)
            return torch.from_numpy(host).to(store._dev())
        image = store.cached(('edge_' + arith, edge_scope, tuple(edge_widths)),
                             build)
        head = (_lib.ptr(p), _lib.ptr(q), wq, int(rest.k_in), _lib.ptr(e),
                int(e.shape[0]), k, _lib.ptr(image), int(rest.n_out),
                int(rest.array[0].relu_from), _edges_sorted_flag(edges) | 2,
                _lib.ptr(agg), agg.stride(0))
        tail = (cnt_e.arg() if cnt_e is not None else None,
                cnt_k.arg() if cnt_k is not None else None, st)
        if arith == 'bf16x3':
            name = "pgnn_edge_mlp_scatter_max_bf16x3_fwd"
            rc = lib.pgnn_edge_mlp_scatter_max_bf16x3_fwd(*(head + tail))
        else:
            name = "pgnn_edge_mlp_scatter_max_f16x2_fwd"
            rc = lib.pgnn_edge_mlp_scatter_max_f16x2_fwd(
                *(head + (_lib.ptr(store.range_status()),) + tail))
        if rc == _lib.E_UNSUPPORTED:
            return False
        _lib.check(rc, name)
        return True


class ClassAwarePredictor(object):
    """gnn.py:121-163.  cls_fn / loc_fn are `partial(multi_layer_fc_fn,
    Ks=..., num_layer=...)` exactly as in models.py:60-69; all heads run as one
    fused block-diagonal MLP chain."""

    def __init__(self, cls_fn, loc_fn):
        self._cls_fn = cls_fn
        self._loc_fn = loc_fn

    @staticmethod
    def _ks(fn):
        kw = getattr(fn, 'keywords', None) or {}
        if 'Ks' not in kw:
            raise NotImplementedError(
                "predictor heads must be partial(multi_layer_fc_fn, Ks=...)")
        return tuple(kw['Ks'])

    def apply_regular(self, features, num_classes, box_encoding_len,
                      normalization_type='fused_BN_center',
                      activation_type='ReLU'):
        """features [K, >=C] -> (logits [K, num_classes], box_encodings
        [K, num_classes, box_encoding_len]).  With one fused head group (every
        shipped config) both results are strided VIEWS of the chain's output
        rows -- box_encodings is [K, nc, 8][:, :, :bl], not contiguous, and
        shares storage with logits: call .contiguous() before handing a data
        pointer on or reshaping with .view()."""
        _check_kinds(activation_type, normalization_type)
        pend = _take_pending(features)
        store = _store()
        scope = _scope('predictor')
        cls_ks, loc_ks = self._ks(self._cls_fn), self._ks(self._loc_fn)
        if len(cls_ks) != 1 or len(loc_ks) != 2 or cls_ks[0] != loc_ks[0] \
                or loc_ks[0] != loc_ks[1]:
            raise NotImplementedError("head shapes other than (h,) / (h,h)")
        hw = cls_ks[0]
        f = _as_f32(features)
        nc, bl = int(num_classes), int(box_encoding_len)
        assert nc <= 16 and bl <= 8

        def build():
            cls = store.mlp(scope + '/cls', 2)
            locs = [store.mlp(scope + '/loc/cls_%d' % j, 3) for j in range(nc)]
            c = cls[0][0].shape[0]
            per_group = max(1, (320 // hw) - 1)   # loc heads beside cls
            groups = []   # (has_cls, [loc ids])
            ids = list(range(nc))
            groups.append((True, ids[:per_group]))
            ids = ids[per_group:]
            per_rest = max(1, 320 // hw)
            while ids:
                groups.append((False, ids[:per_rest]))
                ids = ids[per_rest:]
            chains = []
            for has_cls, lids in groups:
                nh = len(lids) + (1 if has_cls else 0)
                base = 16 if has_cls else 0       # logits block in L2/L3 out
                w1 = np.zeros((c, hw * nh), np.float32)
                b1 = np.zeros(hw * nh, np.float32)
                w2 = np.zeros((hw * nh, base + hw * len(lids)), np.float32)
                b2 = np.zeros(base + hw * len(lids), np.float32)
                w3 = np.zeros((base + hw * len(lids), base + 8 * len(lids)),
                              np.float32)
                b3 = np.zeros(base + 8 * len(lids), np.float32)
                slot = 0
                if has_cls:
                    w1[:, :hw], b1[:hw] = cls[0]
                    w2[:hw, :nc], b2[:nc] = cls[1]
                    w3[np.arange(nc), np.arange(nc)] = 1.0  # pass logits on
                    slot = 1
                for i, j in enumerate(lids):
                    s = slot + i
                    w1[:, hw * s:hw * (s + 1)], b1[hw * s:hw * (s + 1)] = \
                        locs[j][0]
                    w2[hw * s:hw * (s + 1), base + hw * i:base + hw * (i + 1)], \
                        b2[base + hw * i:base + hw * (i + 1)] = locs[j][1]
                    w3[base + hw * i:base + hw * (i + 1),
                       base + 8 * i:base + 8 * i + bl], \
                        b3[base + 8 * i:base + 8 * i + bl] = locs[j][2]
                chain = Chain(store, [(w1, b1, 0), (w2, b2, base),
                                      (w3, b3, w3.shape[1])])
                chains.append((has_cls, lids, base, chain))
            return c, chains
        c, chains = store.cached(('heads', scope, nc, bl, hw), build)
        cnt = _lib.count_of(features)   # capacity form: rows behind the
        logits = None                   # count are undefined in the outputs
        # one group of heads (nc <= 4 at 64 hidden units): the box encodings
        # are a strided VIEW of the chain's output rows, [K, nc, 8][:, :, :bl]
        # -- no copy kernel on the frame's path; several groups are gathered
        # into one tensor
        boxes = None if len(chains) == 1 else \
            torch.empty((f.shape[0], nc, bl), dtype=torch.float32,
                        device=f.device)
        for has_cls, lids, base, chain in chains:
            y = None
            if pend is not None:
                # the producer's update MLP + residual and this group of heads
                # in one launch (pgnn_mlp2_fwd); it writes f as well
                y = self._fused_heads(pend, f, c, chain, cnt)
                if y is None:
                    _flush_pending()
                else:
                    _tail_launched()
                pend = None
            if y is None:
                y = mlp_forward(chain, f, c, count=cnt)
            if has_cls:
                logits = y[:, :nc]
            blk = y[:, base:base + 8 * len(lids)].unflatten(1, (len(lids), 8))
            if boxes is None:
                boxes = blk[:, :, :bl]
            else:
                boxes[:, lids[0]:lids[0] + len(lids), :] = blk[:, :, :bl]
        return _lib.tag_count(logits, cnt), _lib.tag_count(boxes, cnt)


    @staticmethod
    def _fused_heads(pend, f, c, chain, cnt):
        lib = _lib.load()
        res = pend.residual
        rows = int(f.shape[0])
        y = torch.empty((rows, padded_width(chain.n_out)),
                        dtype=torch.float32, device=f.device)
        args = (_lib.ptr(pend.x), pend.x.stride(0), pend.nx, pend.chain.array,
                pend.chain.n, _lib.ptr(res),
                res.stride(0) if res is not None else 0, _lib.ptr(f),
                f.stride(0), int(c), chain.array, chain.n, _lib.ptr(y),
                y.stride(0), rows)
        if cnt is None:
            rc = lib.pgnn_mlp2_fwd(*args, _lib.stream_ptr())
        else:
            rc = lib.pgnn_mlp2_fwd_dyn(*args, cnt.arg(), _lib.stream_ptr())
        if rc == _lib.E_UNSUPPORTED:
            return None
        _lib.check(rc, "pgnn_mlp2_fwd")
        return y if cnt is None else _lib.tag_count(y, cnt)


class ClassAwareSeparatedPredictor(object):
    """gnn.py:165-209 -- registered by the reference (models.py:70), used by no
    shipped config: the class head sees all features, the box head of class j
    only the j-th of `num_classes` equal column groups.  Composed from the
    injected `cls_fn` / `loc_fn` exactly as the reference does."""

    def __init__(self, cls_fn, loc_fn):
        self._cls_fn = cls_fn
        self._loc_fn = loc_fn

    def apply_regular(self, features, num_classes, box_encoding_len,
                      normalization_type='fused_BN_center',
                      activation_type='ReLU'):
        _check_kinds(activation_type, normalization_type)
        _flush_pending()
        f = _as_f32(features)
        nc, bl = int(num_classes), int(box_encoding_len)
        kw = dict(is_logits=True, normalization_type=normalization_type,
                  activation_type=activation_type)
        boxes = []
        with variable_scope('predictor'):
            with variable_scope('cls'):
                y = self._cls_fn(f, num_classes=nc, **kw)
                logits = y[:, :nc]
            # tf.split(features, num_classes, axis=-1): the true width is what
            # the stored first-layer weights of the class head expect
            width = int(_store().mlp(_scope('cls'), 1)[0][0].shape[0])
            assert width % nc == 0, "features do not split into num_classes"
            step = width // nc
            with variable_scope('loc'):
                for j in range(nc):
                    with variable_scope('cls_%d' % j):
                        b = self._loc_fn(f[:, j * step:(j + 1) * step],
                                         num_classes=bl, **kw)
                        boxes.append(b[:, :bl].unsqueeze(1))
        return logits, torch.cat(boxes, dim=1)


def features(t, width):
    """Strip the zero padding: [rows, padded] -> [rows, width] view."""
    return t[:, :width]
