"""Variable naming, initialisation and loading for the Point-GNN model.

Names and shapes are exactly those of the reference's TF variables (SURVEY.md
§8c), so a `tf_bundle.load_checkpoint()` dict is a valid state dict here:

  layerK/extract_vertex_features/fully_connected[_i]/{weights,biases}
  layerK/combined_features/fully_connected[_i]/{weights,biases}
  layerK/fully_connected[_i]/{weights,biases}            (auto-offset MLP)
  output/predictor/cls/fully_connected[_i]/...
  output/predictor/loc/cls_j/fully_connected[_i]/...

`weights` is [in, out] row-major (`slim.fully_connected`, gnn.py:93-103).
"""
import numpy as np

__all__ = ["variable_specs", "init_params", "count_params", "mlp_names"]

_PREDICTOR_SHAPES = {
    # models.py:60-69: (cls hidden widths, loc hidden widths)
    'classaware_predictor': ((64,), (64, 64)),
    'classaware_predictor_128': ((128,), (128, 128)),
}


def mlp_names(scope, n_layers):
    """slim's default scope naming: fully_connected, fully_connected_1, ..."""
    return [scope + '/fully_connected' + ('' if i == 0 else '_%d' % i)
            for i in range(n_layers)]


def _mlp_specs(scope, in_dim, widths):
    specs = []
    d = in_dim
    for name, w in zip(mlp_names(scope, len(widths)), widths):
        specs.append((name + '/weights', (d, w)))
        specs.append((name + '/biases', (w,)))
        d = w
    return specs, d


def variable_specs(config, input_feature_dim=1, box_encoding_len=7):
    """[(name, shape)] for every trainable variable of `config`, in the layer
    order of models.py:119-161."""
    specs = []
    dim = input_feature_dim
    for lc in config['model_kwargs']['layer_configs'][:-1]:
        scope, kw = lc['scope'], lc['kwargs']
        if lc['type'] == 'scatter_max_point_set_pooling':
            s, d = _mlp_specs(scope + '/extract_vertex_features', dim + 3,
                              kw['point_MLP_depth_list'])
            specs += s
            s, dim = _mlp_specs(scope + '/combined_features', d,
                                kw['output_MLP_depth_list'])
            specs += s
        elif lc['type'] == 'scatter_max_graph_auto_center_net':
            if kw['auto_offset']:
                s, _ = _mlp_specs(scope, dim,
                                  kw['auto_offset_MLP_depth_list'])
                specs += s
            s, d = _mlp_specs(scope + '/extract_vertex_features', dim + 3,
                              kw['edge_MLP_depth_list'])
            specs += s
            s, d2 = _mlp_specs(scope + '/combined_features', d,
                               kw['update_MLP_depth_list'])
            specs += s
            if d2 != dim:
                raise ValueError("update MLP must preserve the feature width "
                                 "(residual, gnn.py:372)")
        else:
            raise NotImplementedError(lc['type'])
    pc = config['model_kwargs']['layer_configs'][-1]
    cls_w, loc_w = _PREDICTOR_SHAPES[pc['type']]
    nc = config['num_classes']
    s, _ = _mlp_specs(pc['scope'] + '/predictor/cls', dim,
                      list(cls_w) + [nc])
    specs += s
    for j in range(nc):
        s, _ = _mlp_specs(pc['scope'] + '/predictor/loc/cls_%d' % j, dim,
                          list(loc_w) + [box_encoding_len])
        specs += s
    return specs


def count_params(config, **kw):
    return int(sum(int(np.prod(s)) for _, s in variable_specs(config, **kw)))


def init_params(config, seed=0, bias_scale=0.0, **kw):
    """Seeded Xavier-uniform weights (slim.fully_connected's default
    initializer) and zero biases (`bias_scale` > 0 draws small random biases,
    which makes parity tests sensitive to bias handling)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in variable_specs(config, **kw):
        if name.endswith('/weights'):
            limit = np.sqrt(6.0 / (shape[0] + shape[1]))
            params[name] = rng.uniform(-limit, limit, size=shape).astype(
                np.float32)
        else:
            params[name] = (bias_scale * rng.standard_normal(shape)).astype(
                np.float32)
    return params
