"""Model / graph configurations.

The reference drives everything from JSON documents (`configs/*_config`,
loaded by util/config_util.py:5-9 and copied next to each checkpoint,
train.py:637-638).  `load_config` reads those files unchanged.  Because the
reference tree is not available on the GPU box, the shipped configurations are
also *generated* here (`car_auto_config(T)`, `car_fixed_config(T)`,
`ped_cyl_auto_config(T)`); tests/test_configs.py checks that the generated
dicts equal the reference's JSON files field by field.
"""
import json


def load_config(path):
    """util/config_util.py:5-9."""
    with open(path, 'r') as f:
        return json.load(f)


def _level(level, radius, num_neighbors, scale):
    return {
        "graph_gen_kwargs": {"num_neighbors": num_neighbors, "radius": radius},
        "graph_gen_method": "disjointed_rnn_local_graph_v3",
        "graph_level": level,
        "graph_scale": scale,
    }


def _pooling_layer(point_mlp, out_mlp):
    return {
        "graph_level": 0,
        "kwargs": {
            "output_MLP_activation_type": "ReLU",
            "output_MLP_depth_list": list(out_mlp),
            "output_MLP_normalization_type": "NONE",
            "point_MLP_activation_type": "ReLU",
            "point_MLP_depth_list": list(point_mlp),
            "point_MLP_normalization_type": "NONE",
        },
        "scope": "layer1",
        "type": "scatter_max_point_set_pooling",
    }


def _gnn_layer(idx, width, auto_offset):
    return {
        "graph_level": 1,
        "kwargs": {
            "auto_offset": auto_offset,
            "auto_offset_MLP_depth_list": [64, 3],
            "auto_offset_MLP_feature_activation_type": "ReLU",
            "auto_offset_MLP_normalization_type": "NONE",
            "edge_MLP_activation_type": "ReLU",
            "edge_MLP_depth_list": [width, width],
            "edge_MLP_normalization_type": "NONE",
            "update_MLP_activation_type": "ReLU",
            "update_MLP_depth_list": [width, width],
            "update_MLP_normalization_type": "NONE",
        },
        "scope": "layer%d" % idx,
        "type": "scatter_max_graph_auto_center_net",
    }


def _predictor():
    return {
        "graph_level": 1,
        "kwargs": {"activation_type": "ReLU", "normalization_type": "NONE"},
        "scope": "output",
        "type": "classaware_predictor",
    }


def _config(T, auto_offset, width, point_mlp, out_mlp, num_classes,
            label_method, nms_thres, r0, r1, train_scale, run_scale,
            reg_scale=5e-07):
    layers = [_pooling_layer(point_mlp, out_mlp)]
    layers += [_gnn_layer(2 + i, width, auto_offset) for i in range(T)]
    layers.append(_predictor())
    return {
        "box_encoding_method": "classaware_all_class_box_encoding",
        "downsample_by_voxel_size": None,
        "eval_is_training": True,
        "graph_gen_kwargs": {
            "add_rnd3d": True,
            "base_voxel_size": 0.8,
            "downsample_method": "random",
            "level_configs": [_level(0, r0, -1, train_scale),
                              _level(1, r1, 256, train_scale)],
        },
        "graph_gen_method": "multi_level_local_graph_v3",
        "input_features": "i",
        "label_method": label_method,
        "loss": {"cls_loss_type": "softmax", "cls_loss_weight": 0.1,
                 "loc_loss_weight": 10.0},
        "model_kwargs": {
            "layer_configs": layers,
            "regularizer_kwargs": {"scale": reg_scale},
            "regularizer_type": "l1",
        },
        "model_name": "multi_layer_fast_local_graph_model_v2",
        "nms_overlapped_thres": nms_thres,
        "num_classes": num_classes,
        "runtime_graph_gen_kwargs": {
            "add_rnd3d": False,
            "base_voxel_size": 0.8,
            "level_configs": [_level(0, r0, -1, run_scale),
                              _level(1, r1, -1, run_scale)],
        },
    }


def car_auto_config(T=3):
    """configs/car_auto_T{T}_train_config (T = 0..3 GNN iterations)."""
    return _config(T, True, 300, [32, 64, 128, 300], [300, 300], 4, "Car",
                   0.01, 1.0, 4.0, 1, 0.5)


def car_fixed_config(T=3):
    """configs/car_fixed_T3_train_config: no auto-registration offset."""
    return _config(T, False, 300, [32, 64, 128, 300], [300, 300], 4, "Car",
                   0.01, 1.0, 4.0, 1, 0.5)


def ped_cyl_auto_config(T=3):
    """configs/ped_cyl_auto_T3_trainval_config."""
    return _config(T, True, 256, [32, 64, 128, 256, 512], [256, 256], 6,
                   "Pedestrian_and_Cyclist", 0.2, 0.4, 1.6, 0.5, 0.25,
                   reg_scale=1e-06)


def get_config(name):
    table = {
        "car_auto_T0": lambda: car_auto_config(0),
        "car_auto_T1": lambda: car_auto_config(1),
        "car_auto_T2": lambda: car_auto_config(2),
        "car_auto_T3": lambda: car_auto_config(3),
        "car_fixed_T3": lambda: car_fixed_config(3),
        "ped_cyl_auto_T3": lambda: ped_cyl_auto_config(3),
    }
    return table[name]()


# ---- train / eval configs (configs/*_train_config, *_eval_config) -----------
def load_train_config(path):
    """util/config_util.py:16-20."""
    return load_config(path)


def save_config(path, config):
    """util/config_util.py:11-14 (also used for train configs, :22-25)."""
    with open(path, 'w') as f:
        json.dump(config, f, sort_keys=True, indent=4)


save_train_config = save_config


def _data_aug_configs():
    """The augmentation chain every shipped train config uses
    (preprocess.py:44-78, 239-326)."""
    return [
        {"method_name": "random_rotation_all",
         "method_kwargs": {"method_name": "normal",
                           "yaw_std": 0.39269908169872414,
                           "expend_factor": [1.0, 1.0, 1.0]}},
        {"method_name": "random_flip_all",
         "method_kwargs": {"flip_prob": 0.5}},
        {"method_name": "random_box_shift",
         "method_kwargs": {"method_name": "normal", "xyz_std": [3, 0, 3],
                           "appr_factor": 10,
                           "expend_factor": [1.1, 1.1, 1.1],
                           "max_overlap_num_allowed": 100,
                           "max_overlap_rate": 0.01, "max_trails": 100}},
    ]


_TRAIN_TABLE = {
    # name: (train_dataset, max_epoch, initial_lr, decay_factor, max_steps)
    "car_auto_T0_train": ("train_car.txt", 1718, 0.125, 0.1, 1400000),
    "car_auto_T1_train": ("train_car.txt", 1718, 0.125, 0.1, 1400000),
    "car_auto_T2_train": ("train_car.txt", 1718, 0.125, 0.1, 1400000),
    "car_auto_T3_train": ("train_car.txt", 1718, 0.125, 0.1, 1400000),
    "car_auto_T3_trainval": ("trainval_car.txt", 838, 0.125, 0.1, 1400000),
    "car_fixed_T3_train": ("train_car.txt", 1718, 0.125, 0.1, 1400000),
    "ped_cyl_auto_T3_trainval": ("trainval_pedestrian_cyclist.txt", 1611,
                                 0.32, 0.25, 1000000),
}


def get_train_config(name):
    """configs/<name>_train_config, e.g. name = 'car_auto_T3_train'."""
    dataset, max_epoch, lr, decay, max_steps = _TRAIN_TABLE[name]
    return {
        "NUM_GPU": 2, "NUM_TEST_SAMPLE": -1, "batch_size": 4, "capacity": 1,
        "checkpoint_path": "model", "config_path": "config",
        "data_aug_configs": _data_aug_configs(), "decay_factor": decay,
        "decay_step": 400000, "gpu_memusage": -1, "initial_lr": lr,
        "load_dataset_every_N_time": 0, "load_dataset_to_mem": True,
        "max_epoch": max_epoch, "max_steps": max_steps,
        "num_load_dataset_workers": 16, "optimizer": "sgd",
        "optimizer_kwargs": {}, "save_every_epoch": 20,
        "train_dataset": dataset, "train_dir": "./checkpoints/" + name,
        "unify_copies": True, "visualization": False,
    }


def get_eval_config(name):
    """configs/<name>_eval_config for the same names as get_train_config."""
    max_step = {"car_auto_T3_trainval": 1400298,
                "ped_cyl_auto_T3_trainval": 1000000}.get(name, 1400170)
    return {
        "NUM_TEST_SAMPLE": -1, "checkpoint_path": "model",
        "config_path": "config", "data_aug_configs": [],
        "eval_dataset": "val.txt",
        "eval_dir": "./checkpoints/%s_eval" % name, "eval_every_second": 60,
        "gpu_memusage": -1, "max_step": max_step,
        "train_dir": "./checkpoints/" + name, "visualization": False,
    }
