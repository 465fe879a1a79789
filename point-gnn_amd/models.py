"""Model assembly -- the reference's `models/models.py` surface
(`get_model('multi_layer_fast_local_graph_model_v2')`, `.predict`,
`.postprocess`) on the device operators of `pointgnn_amd.gnn`.

TF1 calls `predict` once with placeholders and runs it later with
`sess.run(fetches, feed_dict)` (run.py:137-140, 260); here `predict` is eager
and is called per frame with real arrays -- NumPy (uploaded, result returned
as NumPy) or torch CUDA tensors (device-resident).  Weights come from
`load_state_dict()` (a dict keyed by the reference's TF variable names, e.g.
`tf_bundle.load_checkpoint(dir)`), the eager stand-in for
`tf.train.Saver.restore` (run.py:199-201).
"""
from functools import partial

import numpy as np
import torch

from . import gnn
from .weights import init_params

__all__ = ["MultiLayerFastLocalGraphModelV2", "get_model", "cls_loss_kind",
           "loss_top_k", "top_k_selection"]


def cls_loss_kind(cls_loss_type, cls_loss_kwargs=None):
    """models.py:210-228 -> (cls_kind of pgnn_loss_fwd_bwd_ex, alpha, gamma)
    with the defaults of models/loss.py:5,31."""
    kw = cls_loss_kwargs or {}
    if cls_loss_type == 'softmax':
        return 0, 0.0, 0.0
    if cls_loss_type == 'focal_softmax':
        return 1, 0.0, float(kw.get('gamma', 2))
    if cls_loss_type == 'focal_sigmoid':
        return 2, float(kw.get('alpha', 0.5)), float(kw.get('gamma', 2))
    if cls_loss_type == 'top_k_softmax':     # softmax CE of the k worst vertices
        return 0, 0.0, 0.0
    raise ValueError("cls_loss_type %r (models.py:210-211 knows softmax, "
                     "top_k_softmax, focal_sigmoid, focal_softmax)"
                     % (cls_loss_type,))


def loss_top_k(cls_loss_type, cls_loss_kwargs, loc_loss_type, loc_loss_kwargs):
    """(k of 'top_k_softmax' or 0, k of 'top_k_huber_loss' or 0):
    models.py:222-228, 266-291."""
    if loc_loss_type not in ('huber_loss', 'top_k_huber_loss'):
        raise ValueError("loc_loss_type %r (models.py:237,266 knows "
                         "huber_loss, top_k_huber_loss)" % (loc_loss_type,))
    k_cls = int((cls_loss_kwargs or {})['k']) \
        if cls_loss_type == 'top_k_softmax' else 0
    k_loc = int((loc_loss_kwargs or {})['k']) \
        if loc_loss_type == 'top_k_huber_loss' else 0
    if (cls_loss_type == 'top_k_softmax' and k_cls < 1) or \
            (loc_loss_type == 'top_k_huber_loss' and k_loc < 1):
        raise ValueError("top-k losses need k >= 1")
    return k_cls, k_loc


def top_k_selection(lg, lab, pb, bl, gt, va, kind, alpha, gamma, cw, k_cls,
                    k_loc):
    """The vertices tf.math.top_k picks (models.py:227, 281) as float masks
    [K] on the device: (sel_cls or None, sel_loc or None).  lg [K, >= nc]
    logits (row stride lg.stride(0)), lab int32 [K], pb [K, nc, bl], gt [K,
    bl], va [K], cw the class-wise loc weights or None -- contiguous CUDA
    tensors.  Per-vertex losses from pgnn_loss_fwd_bwd_sel, membership from
    pgnn_topk_mask_f32; nothing is read back."""
    import ctypes
    from . import _lib
    if not (k_cls or k_loc):
        return None, None
    lib = _lib.load()
    dev = lg.device
    k, nc = int(lab.shape[0]), int(pb.shape[1])
    if max(k_cls, k_loc) > k:
        raise ValueError("top-k loss: k = %d exceeds the batch's %d vertices "
                         "(tf.math.top_k fails the same way)"
                         % (max(k_cls, k_loc), k))
    point_c = torch.empty(k, dtype=torch.float32, device=dev)
    point_l = torch.empty(k, dtype=torch.float32, device=dev)
    sums = torch.empty(4, dtype=torch.float64, device=dev)
    st = _lib.stream_ptr()
    _lib.check(lib.pgnn_loss_fwd_bwd_sel(
        _lib.ptr(lg), lg.stride(0), _lib.ptr(lab), _lib.ptr(pb), bl,
        _lib.ptr(gt), _lib.ptr(va), k, nc, ctypes.c_float(0.0),
        ctypes.c_float(0.0), None, 0.0, 0.0, kind, ctypes.c_float(alpha),
        ctypes.c_float(gamma), _lib.ptr(cw), None, None, ctypes.c_float(1.0),
        _lib.ptr(point_c), _lib.ptr(point_l), _lib.ptr(sums), None, None, st),
        "pgnn_loss_fwd_bwd_sel")
    ws_bytes = int(lib.pgnn_topk_mask_workspace_bytes(k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out = []
    for values, kk in ((point_c, k_cls), (point_l, k_loc)):
        if not kk:
            out.append(None)
            continue
        mask = torch.empty(k, dtype=torch.float32, device=dev)
        _lib.check(lib.pgnn_topk_mask_f32(
            _lib.ptr(values), k, kk, _lib.ptr(mask), _lib.ptr(ws), ws_bytes,
            st), "pgnn_topk_mask_f32")
        out.append(mask)
    return out[0], out[1]


class MultiLayerFastLocalGraphModelV2(object):
    """models.py:22-168."""

    def __init__(self, num_classes, box_encoding_len, regularizer_type=None,
                 regularizer_kwargs=None, layer_configs=None, mode=None):
        self.num_classes = num_classes
        self.box_encoding_len = box_encoding_len
        if regularizer_type is None:
            assert regularizer_kwargs is None, 'No regularizer no kwargs'
        elif regularizer_type not in ('l1', 'l2', 'l1_l2'):
            raise KeyError(regularizer_type)
        self._regularizer_type = regularizer_type
        self._regularizer_kwargs = regularizer_kwargs
        self._layer_configs = layer_configs
        self._default_layers_type = {
            'scatter_max_point_set_pooling': gnn.PointSetPooling(
                point_feature_fn=gnn.multi_layer_neural_network_fn,
                aggregation_fn=gnn.graph_scatter_max_fn,
                output_fn=gnn.multi_layer_neural_network_fn),
            'scatter_max_graph_auto_center_net': gnn.GraphNetAutoCenter(
                edge_feature_fn=gnn.multi_layer_neural_network_fn,
                aggregation_fn=gnn.graph_scatter_max_fn,
                update_fn=gnn.multi_layer_neural_network_fn,
                auto_offset_fn=gnn.multi_layer_neural_network_fn),
            'classaware_predictor': gnn.ClassAwarePredictor(
                cls_fn=partial(gnn.multi_layer_fc_fn, Ks=(64,), num_layer=2),
                loc_fn=partial(gnn.multi_layer_fc_fn, Ks=(64, 64,),
                               num_layer=3)),
            'classaware_predictor_128': gnn.ClassAwarePredictor(
                cls_fn=partial(gnn.multi_layer_fc_fn, Ks=(128,), num_layer=2),
                loc_fn=partial(gnn.multi_layer_fc_fn, Ks=(128, 128),
                               num_layer=3)),
            'classaware_separated_predictor': gnn.ClassAwareSeparatedPredictor(
                cls_fn=partial(gnn.multi_layer_fc_fn, Ks=(64,), num_layer=2),
                loc_fn=partial(gnn.multi_layer_fc_fn, Ks=(64, 64,),
                               num_layer=3)),
        }
        assert mode in ['train', 'eval', 'test'], 'Unsupported mode'
        self._mode = mode
        self._store = None
        self._edge_arith = 'f32'

    @property
    def edge_arith(self):
        """Arithmetic of the per-edge product of the GraphNetAutoCenter layers
        (gnn.EDGE_ARITHS): 'f32' (default, the parity reference) or the
        secondary 'bf16x3'.  Not part of the reference's surface."""
        return self._edge_arith

    @edge_arith.setter
    def edge_arith(self, value):
        if value not in gnn.EDGE_ARITHS:
            raise ValueError("edge_arith must be one of %r, not %r"
                             % (gnn.EDGE_ARITHS, value))
        self._edge_arith = value
        if self._store is not None:
            self._store.edge_arith = value

    def edge_range_ok(self):
        """edge_arith 'f16x2' only: False when an activation of a frame since
        the last call reached 32768 (fp16 ends at 65504; the kernel clamps) --
        such a frame's results are unreliable, rerun it with edge_arith 'f32'.
        One small device-to-host read; always True for the other arithmetics."""
        return self._store is None or self._store.edge_range_ok()

    # ---- weights ----------------------------------------------------------
    def load_state_dict(self, params, device=None):
        """params: {TF variable name: ndarray} (extra keys such as the
        global-step `Variable` are ignored)."""
        self._store = gnn.ParamStore(params, device)
        self._store.edge_arith = self._edge_arith
        return self

    def init_weights(self, config, seed=0, **kw):
        """Seeded Xavier-uniform weights of the shapes `config` implies."""
        return self.load_state_dict(init_params(config, seed=seed, **kw))

    def state_dict(self):
        return dict(self._store.params)

    # ---- forward ----------------------------------------------------------
    def predict(self, t_initial_vertex_features, t_vertex_coord_list,
                t_keypoint_indices_list, t_edges_list, is_training):
        """models.py:79-163.  Returns (logits [K, num_classes], box_encodings
        [K, num_classes, box_encoding_len])."""
        if self._store is None:
            raise RuntimeError("call load_state_dict()/init_weights() first")
        if is_training:
            # slim.batch_norm(is_training=True) normalizes with the batch's own
            # statistics (models.py:112 arg_scope): only the moving-statistics
            # form is folded into the layers here
            for cfg in self._layer_configs:
                for key, val in (cfg.get('kwargs') or {}).items():
                    if key.endswith('normalization_type') and \
                            val in gnn.FOLDED_NORMALIZATIONS:
                        raise NotImplementedError(
                            "%s: %s = %r with is_training=True (batch "
                            "statistics) has no device path" % (
                                cfg['scope'], key, val))
        was_np = not isinstance(t_initial_vertex_features, torch.Tensor)
        if was_np:
            dev = self._store._dev()

            def up(a, dt):
                return torch.from_numpy(np.ascontiguousarray(
                    np.asarray(a), dtype=dt)).to(dev)
            t_initial_vertex_features = up(t_initial_vertex_features,
                                           np.float32)
            t_vertex_coord_list = [up(c, np.float32)
                                   for c in t_vertex_coord_list]
            t_keypoint_indices_list = [
                None if k is None else up(k, np.int32)
                for k in t_keypoint_indices_list]
            t_edges_list = [up(e, np.int32) for e in t_edges_list]
        # set `model.keep_features = True` to keep every layer's output
        # (models.py:113-147's `tfeatures` after each layer) in
        # `model.feature_list` -- per-layer parity checks
        keep = getattr(self, 'keep_features', False)
        self.feature_list = []
        # `model.fuse_vertex_stages = False`: every operator launches its own
        # per-vertex stages (the unfused reference form the tests compare with)
        fuse = getattr(self, 'fuse_vertex_stages', True)
        with gnn.parameters(self._store), gnn.fuse_vertex_stages(fuse):
            feats = t_initial_vertex_features
            *body, head = self._layer_configs
            for cfg in body:
                level = cfg['graph_level']
                op = self._default_layers_type[cfg['type']]
                with gnn.variable_scope(cfg['scope']):
                    feats = op.apply_regular(
                        feats, t_vertex_coord_list[level],
                        t_keypoint_indices_list[level], t_edges_list[level],
                        **cfg['kwargs'])
                if keep:
                    self.feature_list.append(feats)
            if head['type'] not in ('classaware_predictor',
                                    'classaware_predictor_128',
                                    'classaware_separated_predictor'):
                raise AssertionError("last layer must be a predictor, got %r"
                                     % (head['type'],))
            with gnn.variable_scope(head['scope']):
                logits, box_encodings = \
                    self._default_layers_type[head['type']].apply_regular(
                        feats, num_classes=self.num_classes,
                        box_encoding_len=self.box_encoding_len,
                        **head['kwargs'])
        self.last_features = feats
        if was_np:
            return logits.cpu().numpy(), box_encodings.cpu().numpy()
        return logits, box_encodings

    def postprocess(self, logits):
        """models.py:165-168."""
        if isinstance(logits, torch.Tensor):
            return torch.softmax(logits, dim=-1)
        z = logits - logits.max(axis=-1, keepdims=True)
        e = np.exp(z)
        return e / e.sum(axis=-1, keepdims=True)

    def loss(self, logits, labels, pred_box, gt_box, valid_box,
             cls_loss_type='focal_sigmoid', cls_loss_kwargs={},
             loc_loss_type='huber_loss', loc_loss_kwargs={},
             loc_loss_weight=1.0, cls_loss_weight=1.0):
        """models.py:170-311: cls 'softmax' (every shipped config),
        'top_k_softmax', 'focal_softmax', 'focal_sigmoid'; loc 'huber_loss' /
        'top_k_huber_loss' with the optional 'classwise_loc_loss_weight'
        (train mode).  Same keys as the reference's
        loss_dict; values are Python floats (classwise_loc_loss: list of
        [box_len] tensors).  Device tensors in; the per-vertex arithmetic runs
        in pgnn_loss_fwd_bwd.  Gradients: see pointgnn_amd.train.Trainer."""
        import ctypes
        from . import _lib
        if isinstance(loc_loss_weight, dict):
            loc_loss_weight = loc_loss_weight[self._mode]
        if isinstance(cls_loss_weight, dict):
            cls_loss_weight = cls_loss_weight[self._mode]
        if isinstance(cls_loss_type, dict):      # models.py:203-208
            cls_loss_type = cls_loss_type[self._mode]
            cls_loss_kwargs = cls_loss_kwargs[self._mode]
        if isinstance(loc_loss_type, dict):
            loc_loss_type = loc_loss_type[self._mode]
            loc_loss_kwargs = loc_loss_kwargs[self._mode]
        kind, alpha, gamma = cls_loss_kind(cls_loss_type, cls_loss_kwargs)
        k_cls, k_loc = loss_top_k(cls_loss_type, cls_loss_kwargs,
                                  loc_loss_type, loc_loss_kwargs)
        lib = _lib.load()
        dev = logits.device
        k, nc = int(logits.shape[0]), int(logits.shape[1])
        bl = int(pred_box.shape[-1])
        lg = logits.to(torch.float32).contiguous()
        lab = labels.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        pb = pred_box.to(torch.float32).contiguous()
        gt = gt_box.to(device=dev, dtype=torch.float32).reshape(k, bl).contiguous()
        va = valid_box.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        sums = torch.zeros(4, dtype=torch.float64, device=dev)
        # models.py:240-246: the per-class weights apply in 'train' mode only
        cw = None
        if 'classwise_loc_loss_weight' in (loc_loss_kwargs or {}) and \
                self._mode == 'train':
            cw = torch.tensor(loc_loss_kwargs['classwise_loc_loss_weight'],
                              dtype=torch.float32, device=dev)
        sel_c = sel_l = None
        if k_cls or k_loc:
            sel_c, sel_l = top_k_selection(lg, lab, pb, bl, gt, va, kind,
                                           alpha, gamma, cw, k_cls, k_loc)
            _lib.check(lib.pgnn_loss_fwd_bwd_sel(
                _lib.ptr(lg), lg.stride(0), _lib.ptr(lab), _lib.ptr(pb), bl,
                _lib.ptr(gt), _lib.ptr(va), k, nc, ctypes.c_float(0.0),
                ctypes.c_float(0.0), None, 0.0, 0.0, kind,
                ctypes.c_float(alpha), ctypes.c_float(gamma), _lib.ptr(cw),
                _lib.ptr(sel_c), _lib.ptr(sel_l),
                ctypes.c_float(k / k_cls if k_cls else 1.0), None, None,
                _lib.ptr(sums), None, None, _lib.stream_ptr()),
                "pgnn_loss_fwd_bwd_sel")
        elif kind == 0 and cw is None:
            _lib.check(lib.pgnn_loss_fwd_bwd(
                _lib.ptr(lg), lg.stride(0), _lib.ptr(lab), _lib.ptr(pb), bl,
                _lib.ptr(gt), _lib.ptr(va), k, nc, ctypes.c_float(0.0),
                ctypes.c_float(0.0), _lib.ptr(sums), None, None,
                _lib.stream_ptr()), "pgnn_loss_fwd_bwd")
        else:
            _lib.check(lib.pgnn_loss_fwd_bwd_ex(
                _lib.ptr(lg), lg.stride(0), _lib.ptr(lab), _lib.ptr(pb), bl,
                _lib.ptr(gt), _lib.ptr(va), k, nc, ctypes.c_float(0.0),
                ctypes.c_float(0.0), None, 0.0, 0.0, kind,
                ctypes.c_float(alpha), ctypes.c_float(gamma), _lib.ptr(cw),
                _lib.ptr(sums), None, None, _lib.stream_ptr()),
                "pgnn_loss_fwd_bwd_ex")
        s_ce, s_loc, n, nv = [float(v) for v in sums.cpu()]
        reg = 0.0
        if self._regularizer_type == 'l1' and self._store is not None:
            scale = float(self._regularizer_kwargs['scale'])
            reg = scale * float(sum(np.abs(v).sum() for name, v in
                                    self._store.params.items()
                                    if name.endswith('/weights')))
        # per-class sums of the per-vertex Huber vectors (diagnostic only)
        sel = pb[torch.arange(k, device=dev), lab.long()]
        err = sel - gt
        ae = err.abs()
        quad = torch.clamp(ae, max=1.0)
        all_loc = loc_loss_weight * (0.5 * quad * quad + (ae - quad)) * va[:, None]
        if cw is not None:
            all_loc = all_loc * cw[lab.long()][:, None]
        if sel_l is not None:        # models.py:286-299: the selected rows only
            all_loc = all_loc * sel_l[:, None]
        classwise = [all_loc[lab == c].sum(dim=0) for c in range(self.num_classes)]
        return {
            'cls_loss': cls_loss_weight * s_ce / max(n, 1.0),
            'loc_loss': loc_loss_weight * s_loc / nv if nv > 0 else 0.0,
            'reg_loss': reg, 'num_endpoint': int(n), 'num_valid_endpoint': nv,
            'classwise_loc_loss': classwise,
        }


def get_model(model_name):
    """models.py:313-319."""
    model_map = {
        'multi_layer_fast_local_graph_model_v2':
            MultiLayerFastLocalGraphModelV2,
    }
    return model_map[model_name]
