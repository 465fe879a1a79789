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

__all__ = ["MultiLayerFastLocalGraphModelV2", "get_model"]


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

    # ---- weights ----------------------------------------------------------
    def load_state_dict(self, params, device=None):
        """params: {TF variable name: ndarray} (extra keys such as the
        global-step `Variable` are ignored)."""
        self._store = gnn.ParamStore(params, device)
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
        with gnn.parameters(self._store):
            tfeatures = t_initial_vertex_features
            for idx in range(len(self._layer_configs) - 1):
                layer_config = self._layer_configs[idx]
                graph_level = layer_config['graph_level']
                with gnn.variable_scope(layer_config['scope']):
                    flgn = self._default_layers_type[layer_config['type']]
                    tfeatures = flgn.apply_regular(
                        tfeatures,
                        t_vertex_coord_list[graph_level],
                        t_keypoint_indices_list[graph_level],
                        t_edges_list[graph_level],
                        **layer_config['kwargs'])
            predictor_config = self._layer_configs[-1]
            assert (predictor_config['type'] == 'classaware_predictor' or
                    predictor_config['type'] == 'classaware_predictor_128' or
                    predictor_config['type'] ==
                    'classaware_separated_predictor')
            predictor = self._default_layers_type[predictor_config['type']]
            with gnn.variable_scope(predictor_config['scope']):
                logits, box_encodings = predictor.apply_regular(
                    tfeatures, num_classes=self.num_classes,
                    box_encoding_len=self.box_encoding_len,
                    **predictor_config['kwargs'])
        self.last_features = tfeatures
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

    def loss(self, *args, **kwargs):
        raise NotImplementedError(
            "training loss (models.py:170-311) is scheduled after the "
            "inference path: see DESIGN.md, scope row a11")


def get_model(model_name):
    """models.py:313-319."""
    model_map = {
        'multi_layer_fast_local_graph_model_v2':
            MultiLayerFastLocalGraphModelV2,
    }
    return model_map[model_name]
