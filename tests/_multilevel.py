"""A graph configuration with MORE THAN ONE pooling level (graph_gen.py:49-90,
:92-153 loop over arbitrary `levels`; no shipped config has one): scales 1,
2.5, 2.5 x base voxel 0.4 m -- two pooling levels, then a same-scale GNN level.
Shared by tests/golden/make_golden_multilevel.py and the tests."""

BASE_VOXEL = 0.4
LEVEL_CONFIGS = [
    {'graph_gen_method': 'disjointed_rnn_local_graph_v3',
     'graph_gen_kwargs': {'radius': 1.0, 'num_neighbors': -1},
     'graph_level': 0, 'graph_scale': 1},
    {'graph_gen_method': 'disjointed_rnn_local_graph_v3',
     'graph_gen_kwargs': {'radius': 2.5, 'num_neighbors': -1},
     'graph_level': 1, 'graph_scale': 2.5},
    {'graph_gen_method': 'disjointed_rnn_local_graph_v3',
     'graph_gen_kwargs': {'radius': 4.0, 'num_neighbors': -1},
     'graph_level': 2, 'graph_scale': 2.5},
]


def model_config(num_classes=4, first_width=300):
    """A model on that graph: PointSetPooling at levels 0 AND 1, one
    GraphNetAutoCenter iteration at level 2, the class-aware predictor.
    `first_width`: output width of the first pooling level (8: the second level
    then takes the fused narrow-feature kernel, on zero-padded [K, 16] rows)."""
    def pooling(level, scope, point_mlp, out_mlp):
        return {"graph_level": level, "scope": scope,
                "type": "scatter_max_point_set_pooling",
                "kwargs": {"output_MLP_activation_type": "ReLU",
                           "output_MLP_depth_list": list(out_mlp),
                           "output_MLP_normalization_type": "NONE",
                           "point_MLP_activation_type": "ReLU",
                           "point_MLP_depth_list": list(point_mlp),
                           "point_MLP_normalization_type": "NONE"}}
    gnn = {"graph_level": 2, "scope": "layer3",
           "type": "scatter_max_graph_auto_center_net",
           "kwargs": {"auto_offset": True,
                      "auto_offset_MLP_depth_list": [64, 3],
                      "auto_offset_MLP_feature_activation_type": "ReLU",
                      "auto_offset_MLP_normalization_type": "NONE",
                      "edge_MLP_activation_type": "ReLU",
                      "edge_MLP_depth_list": [300, 300],
                      "edge_MLP_normalization_type": "NONE",
                      "update_MLP_activation_type": "ReLU",
                      "update_MLP_depth_list": [300, 300],
                      "update_MLP_normalization_type": "NONE"}}
    head = {"graph_level": 2, "scope": "output", "type": "classaware_predictor",
            "kwargs": {"activation_type": "ReLU", "normalization_type": "NONE"}}
    return {
        "num_classes": num_classes,
        "loss": {"cls_loss_type": "softmax", "cls_loss_weight": 0.1,
                 "loc_loss_weight": 10.0},
        "model_name": "multi_layer_fast_local_graph_model_v2",
        "model_kwargs": {
            "layer_configs": [pooling(0, "layer1", [32, 64, 128, first_width],
                                      [first_width, first_width]),
                              pooling(1, "layer2", [300, 300], [300, 300]),
                              gnn, head],
            "regularizer_kwargs": {"scale": 5e-07}, "regularizer_type": "l1"},
    }
