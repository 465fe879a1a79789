"""ctypes binding of libpointgnn_hip.so (include/pointgnn_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, this module raises.  torch is imported first so that the library's
`libamdhip64.so.7` dependency resolves to the HIP runtime PyTorch already
loaded (one runtime per process: torch's streams and allocations are then
valid inside the library).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGNN_LIB selects another build of the same library (kernel A/B runs in tools/)
LIB_PATH = os.environ.get("PGNN_LIB") or os.path.join(_HERE, "libpointgnn_hip.so")

c_i32, c_i64, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
c_vp, c_sz, c_u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64

PGNN_MAX_LAYERS = 8


class FcLayer(ctypes.Structure):
    """struct pgnn_fc_layer"""
    _fields_ = [("packed", c_vp), ("k_in", c_i32), ("n_out", c_i32),
                ("relu_from", c_i32)]


class DynCount(ctypes.Structure):
    """struct pgnn_dyn_count: a size that lives in device memory (`dev`, an
    int32) plus the value the host expects (`hint`)."""
    _fields_ = [("dev", c_vp), ("hint", c_i64)]


class WgradJob(ctypes.Structure):
    """struct pgnn_wgrad_job (pgnn_weight_grad_many_f32)"""
    _fields_ = [("X", c_vp), ("ld_x", c_i64), ("dZ", c_vp), ("ld_dz", c_i64),
                ("n_rows", c_i64), ("dW", c_vp), ("db", c_vp), ("k_in", c_i32),
                ("n_out", c_i32), ("accumulate", c_i32), ("reserved", c_i32)]


class MergeJob(ctypes.Structure):
    """struct pgnn_merge_job (pgnn_merge_rows)"""
    _fields_ = [("src", c_vp), ("dst", c_vp), ("n_words", c_i64),
                ("add0", c_i32), ("add1", c_i32)]


class PackJob(ctypes.Structure):
    """One record of pgnn_pack_fc_many's job table."""
    _fields_ = [("w", c_vp), ("b", c_vp), ("dst", c_vp), ("k_in", c_i32),
                ("n_out", c_i32), ("kind", c_i32), ("first_block", c_i32),
                ("ld", c_i32), ("reserved", c_i32)]


TRAIN_MAX_FC, TRAIN_MAX_STAGES, TRAIN_MAX_CLASSES, TRAIN_MAX_LEVELS = 8, 8, 16, 4


class TrainFc(ctypes.Structure):
    """struct pgnn_train_fc"""
    _fields_ = [("w_off", c_i64), ("b_off", c_i64), ("k_in", c_i32),
                ("n_out", c_i32)]


class TrainStage(ctypes.Structure):
    """struct pgnn_train_stage"""
    _fields_ = [("kind", c_i32), ("graph_level", c_i32), ("n_a", c_i32),
                ("n_b", c_i32), ("n_c", c_i32), ("reserved", c_i32),
                ("a", TrainFc * TRAIN_MAX_FC), ("b", TrainFc * TRAIN_MAX_FC),
                ("c", TrainFc * TRAIN_MAX_FC)]


class TrainModel(ctypes.Structure):
    """struct pgnn_train_model"""
    _fields_ = [("n_stages", c_i32), ("num_classes", c_i32),
                ("box_len", c_i32), ("reserved", c_i32), ("n_params", c_i64),
                ("stages", TrainStage * TRAIN_MAX_STAGES),
                ("cls", TrainFc * 2),
                ("loc", (TrainFc * 3) * TRAIN_MAX_CLASSES)]


class TrainBatch(ctypes.Structure):
    """struct pgnn_train_batch"""
    _fields_ = [("input_v", c_vp), ("n_feat", c_i32), ("n_levels", c_i32),
                ("n_vertices", c_i64 * (TRAIN_MAX_LEVELS + 1)),
                ("coords", c_vp * (TRAIN_MAX_LEVELS + 1)),
                ("keypoints", c_vp * TRAIN_MAX_LEVELS),
                ("edges", c_vp * TRAIN_MAX_LEVELS),
                ("n_edges", c_i64 * TRAIN_MAX_LEVELS),
                ("edges_sorted", c_i32 * TRAIN_MAX_LEVELS)]


class PointGnnHipError(RuntimeError):
    pass


_SIGNATURES = {
    # name: (restype, argtypes)
    "pgnn_version": (c_i32, []),
    "pgnn_last_error": (ctypes.c_char_p, []),
    "pgnn_stream_create_cu_mask": (c_i32, [c_i32, c_i32, c_i32,
                                           ctypes.POINTER(c_vp)]),
    "pgnn_stream_destroy": (c_i32, [c_vp]),
    "pgnn_check_device_pointer": (c_i32, [c_vp]),
    "pgnn_set_tunable": (c_i32, [ctypes.c_char_p, c_i32]),
    "pgnn_set_debug_buffer": (c_i32, [c_vp]),
    "pgnn_scatter_max_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                     c_vp, c_i64, c_i32, c_vp]),
    "pgnn_scatter_sum_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                     c_vp, c_i64, c_i32, c_vp, c_vp]),
    "pgnn_radius_graph_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pgnn_radius_graph_count": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64, c_vp,
                                        c_vp, c_sz, c_vp, c_vp]),
    "pgnn_radius_graph_fill": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64, c_vp,
                                       c_vp, c_sz, c_vp, c_vp, c_i64, c_vp]),
    "pgnn_radius_graph_count_f64": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64,
                                            c_vp, c_vp, c_sz, c_vp, c_vp]),
    "pgnn_radius_graph_fill_f64": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64,
                                           c_vp, c_vp, c_sz, c_vp, c_vp, c_i64,
                                           c_vp]),
    "pgnn_radius_graph_dyn_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pgnn_radius_graph_dyn": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp,
                                      c_f64, c_vp, c_vp, c_sz, c_vp, c_i64,
                                      c_vp, c_vp]),
    "pgnn_radius_graph_dyn_f64": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp,
                                          c_f64, c_vp, c_vp, c_sz, c_vp, c_i64,
                                          c_vp, c_vp]),
    "pgnn_radius_graph_dyn_grid": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64,
                                           c_vp, c_vp, c_sz, c_vp]),
    "pgnn_radius_graph_dyn_grid_f64": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_f64,
                                               c_vp, c_vp, c_sz, c_vp]),
    "pgnn_radius_graph_dyn_query": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp,
                                            c_f64, c_vp, c_vp, c_sz, c_vp,
                                            c_i64, c_vp, c_vp]),
    "pgnn_radius_graph_dyn_query_f64": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp,
                                                c_f64, c_vp, c_vp, c_sz, c_vp,
                                                c_i64, c_vp, c_vp]),
    "pgnn_cap_neighbors_count": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "pgnn_cap_neighbors_fill": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_u64, c_vp,
                                        c_vp, c_i64, c_vp]),
    "pgnn_radius_graph_dyn_cap": (c_i32, [c_vp, c_sz, c_i64, c_i64, c_vp, c_i64,
                                          c_vp, c_i32, c_u64, c_vp, c_vp, c_i64,
                                          c_vp, c_vp]),
    "pgnn_keypoints_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_voxel_keypoints_center": (c_i32, [c_vp, c_i64, c_f64, c_vp, c_sz,
                                            c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pgnn_voxel_keypoints_random": (c_i32, [c_vp, c_i64, c_f64, c_vp, c_u64,
                                            c_vp, c_sz, c_vp, c_vp, c_vp,
                                            c_vp]),
    "pgnn_voxel_keypoints_random_f64": (c_i32, [c_vp, c_i64, c_f64, c_vp, c_u64,
                                                c_vp, c_sz, c_vp, c_vp, c_vp,
                                                c_vp]),
    "pgnn_keypoints_from_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pgnn_voxel_keypoints_center_from": (c_i32, [c_vp, c_i64, c_vp, c_i64,
                                                 c_f64, c_vp, c_sz, c_vp, c_vp,
                                                 c_vp, c_vp]),
    "pgnn_voxel_keypoints_random_from": (c_i32, [c_vp, c_i64, c_vp, c_i64,
                                                 c_f64, c_vp, c_u64, c_vp, c_sz,
                                                 c_vp, c_vp, c_vp, c_vp]),
    "pgnn_voxel_keypoints_random_from_f64": (c_i32, [c_vp, c_i64, c_vp, c_i64,
                                                     c_f64, c_vp, c_u64, c_vp,
                                                     c_sz, c_vp, c_vp, c_vp,
                                                     c_vp]),
    "pgnn_kdtree_shape": (c_i32, [c_i64, c_vp, c_vp]),
    "pgnn_kdtree_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_kdtree_replica": (c_i32, [c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_vp,
                                    c_vp]),
    "pgnn_packed_fc_floats": (c_sz, [c_i32, c_i32]),
    "pgnn_pack_fc": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "pgnn_mlp_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i64,
                             ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_vp,
                             c_i64, c_vp]),
    "pgnn_mlp_fwd_dyn": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_i64,
                                 ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64,
                                 c_vp, c_i64, ctypes.POINTER(DynCount), c_vp]),
    "pgnn_point_set_pooling_fwd_dyn": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp,
                                               c_i64, c_i32,
                                               ctypes.POINTER(FcLayer), c_i32,
                                               c_i32, c_vp, c_i64, c_vp,
                                               ctypes.POINTER(DynCount),
                                               ctypes.POINTER(DynCount),
                                               c_vp]),
    "pgnn_point_set_pooling_workspace_bytes": (c_i32, [
        ctypes.POINTER(FcLayer), c_i32, c_i32, c_i64, c_i64,
        ctypes.POINTER(c_sz)]),
    "pgnn_point_set_pooling_fwd_ws": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp,
                                              c_i64, c_i32,
                                              ctypes.POINTER(FcLayer), c_i32,
                                              c_i32, c_vp, c_i64, c_vp,
                                              ctypes.POINTER(DynCount),
                                              ctypes.POINTER(DynCount),
                                              c_vp, c_sz, c_vp]),
    "pgnn_edge_mlp_scatter_max_fwd_dyn": (c_i32, [c_vp, c_vp, c_i64, c_i32,
                                                  c_vp, c_i64, c_i32,
                                                  ctypes.POINTER(FcLayer),
                                                  c_i32, c_i32, c_vp, c_i64,
                                                  c_vp,
                                                  ctypes.POINTER(DynCount),
                                                  ctypes.POINTER(DynCount),
                                                  c_vp]),
    "pgnn_vertex_pre_edge_fwd_dyn": (c_i32, [c_vp, c_i64, c_i32, c_vp,
                                             ctypes.POINTER(FcLayer), c_i32,
                                             ctypes.POINTER(FcLayer), c_vp,
                                             c_i64, c_vp, c_vp, c_i64, c_vp,
                                             c_i64, ctypes.POINTER(DynCount),
                                             c_vp]),
    "pgnn_point_set_pooling_fwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp,
                                           c_i64, c_i32,
                                           ctypes.POINTER(FcLayer), c_i32,
                                           c_i32, c_vp, c_i64, c_vp, c_vp]),
    "pgnn_edge_mlp_scatter_max_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp,
                                              c_i64, c_i32,
                                              ctypes.POINTER(FcLayer), c_i32,
                                              c_i32, c_vp, c_i64, c_vp,
                                              c_vp]),
    "pgnn_point_set_pooling_rows_fwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp,
                                                c_i64, c_i32,
                                                ctypes.POINTER(FcLayer), c_i32,
                                                c_i32, c_vp, c_i64,
                                                ctypes.POINTER(c_vp), c_i64,
                                                c_vp]),
    "pgnn_edge_mlp_scatter_max_rows_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32,
                                                   c_vp, c_i64, c_i32,
                                                   ctypes.POINTER(FcLayer),
                                                   c_i32, c_vp, c_i64, c_vp,
                                                   c_i64, c_vp, c_vp]),
    "pgnn_packed_fc_bf16x3_bytes": (c_sz, [c_i32, c_i32]),
    "pgnn_pack_fc_bf16x3": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "pgnn_edge_mlp_scatter_max_bf16x3_fwd": (c_i32, [
        c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32,
        c_i32, c_vp, c_i64, ctypes.POINTER(DynCount), ctypes.POINTER(DynCount),
        c_vp]),
    "pgnn_packed_fc_f16x2_bytes": (c_sz, [c_i32, c_i32]),
    "pgnn_pack_fc_f16x2": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "pgnn_edge_mlp_scatter_max_f16x2_fwd": (c_i32, [
        c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32,
        c_i32, c_vp, c_i64, c_vp, ctypes.POINTER(DynCount),
        ctypes.POINTER(DynCount), c_vp]),
    "pgnn_pack_fc_f16x2_acc": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "pgnn_point_set_pooling_f16x2_fwd": (c_i32, [
        c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32, ctypes.POINTER(FcLayer),
        c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(DynCount),
        ctypes.POINTER(DynCount), c_vp]),
    "pgnn_offset_apply": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp,
                                  c_i64, c_vp]),
    "pgnn_vertex_pre_edge_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp,
                                         ctypes.POINTER(FcLayer), c_i32,
                                         ctypes.POINTER(FcLayer), c_vp, c_i64,
                                         c_vp, c_vp, c_i64, c_vp, c_i64,
                                         c_vp]),
    "pgnn_vertex_update_pre_edge_fwd": (c_i32, [
        c_vp, c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_vp,
        c_i64, c_i32, c_vp, ctypes.POINTER(FcLayer), c_i32,
        ctypes.POINTER(FcLayer), c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64,
        c_vp]),
    "pgnn_vertex_update_pre_edge_fwd_dyn": (c_i32, [
        c_vp, c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_vp,
        c_i64, c_i32, c_vp, ctypes.POINTER(FcLayer), c_i32,
        ctypes.POINTER(FcLayer), c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64,
        ctypes.POINTER(DynCount), c_vp]),
    "pgnn_mlp2_fwd": (c_i32, [
        c_vp, c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_vp,
        c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_i64,
        c_vp]),
    "pgnn_mlp2_fwd_dyn": (c_i32, [
        c_vp, c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_vp,
        c_i64, c_i32, ctypes.POINTER(FcLayer), c_i32, c_vp, c_i64, c_i64,
        ctypes.POINTER(DynCount), c_vp]),
    # training step
    "pgnn_pack_fc_device": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp,
                                    c_vp]),
    "pgnn_pack_fc_many": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "pgnn_segmax_fc_bwd_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32, c_i32]),
    "pgnn_edge_segmax_fc_bwd_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64,
                                            c_i32, c_i32, c_vp, c_i64, c_vp,
                                            c_i64, c_vp, c_i64, c_vp, c_vp,
                                            c_i32, c_vp, c_i64, c_vp, c_vp,
                                            c_i64, c_vp, c_vp, c_vp, c_sz,
                                            c_vp]),
    "pgnn_segmax_fc_bwd_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                       c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                       c_i32, c_vp, c_i64, c_vp, c_i64, c_i32,
                                       c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pgnn_trainer_create": (c_i32, [ctypes.POINTER(TrainModel),
                                    ctypes.POINTER(c_vp)]),
    "pgnn_trainer_destroy": (c_i32, [c_vp]),
    "pgnn_trainer_images_bytes": (c_sz, [c_vp]),
    "pgnn_trainer_bind": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pgnn_trainer_repack": (c_i32, [c_vp, c_vp]),
    "pgnn_trainer_workspace_bytes": (c_sz, [c_vp, ctypes.POINTER(TrainBatch)]),
    "pgnn_trainer_forward": (c_i32, [c_vp, ctypes.POINTER(TrainBatch), c_vp,
                                     c_sz, ctypes.POINTER(c_vp),
                                     ctypes.POINTER(c_i64),
                                     ctypes.POINTER(c_vp), c_vp]),
    "pgnn_trainer_backward": (c_i32, [c_vp, ctypes.POINTER(TrainBatch), c_vp,
                                      c_sz, c_vp, c_vp, c_vp]),
    "pgnn_trainer_backward_sync": (c_i32, [c_vp, ctypes.POINTER(TrainBatch),
                                           c_vp, c_sz, c_vp, c_vp, c_vp, c_vp,
                                           c_i64, c_vp]),
    # collectives (RCCL)
    "pgnn_comm_unique_id": (c_i32, [c_vp]),
    "pgnn_comm_init_rank": (c_i32, [c_vp, c_i32, c_i32, ctypes.POINTER(c_vp)]),
    "pgnn_comm_destroy": (c_i32, [c_vp]),
    "pgnn_comm_info": (c_i32, [c_vp, ctypes.POINTER(c_i32),
                               ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "pgnn_comm_library": (ctypes.c_char_p, []),
    "pgnn_comm_async_error": (c_i32, [c_vp]),
    "pgnn_allreduce_sum_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "pgnn_allreduce_sum_f64": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "pgnn_allreduce_step": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "pgnn_broadcast_f32": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pgnn_edge_hidden_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                     c_vp]),
    "pgnn_edge_hidden_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp,
                                     c_vp, c_vp]),
    "pgnn_pool_features_wide_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp,
                                            c_i64, c_vp, c_i64, c_vp]),
    "pgnn_pool_features_fwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i64,
                                       c_vp, c_vp]),
    "pgnn_relu_mask_mul": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "pgnn_merge_rows": (c_i32, [ctypes.POINTER(MergeJob), c_i32, c_vp]),
    "pgnn_scatter_max_bwd_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32,
                                         c_i32, c_vp, c_i64, c_vp, c_i64, c_vp,
                                         c_vp, c_i64, c_i32, c_vp]),
    "pgnn_weight_grad_workspace_bytes": (c_sz, [c_i32, c_i32, c_i64]),
    "pgnn_weight_grad_f32": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32,
                                     c_i64, c_vp, c_vp, c_i32, c_vp, c_sz,
                                     c_vp]),
    "pgnn_weight_grad_many_workspace_bytes": (c_sz,
                                              [ctypes.POINTER(WgradJob),
                                               c_i32]),
    "pgnn_weight_grad_many_f32": (c_i32, [ctypes.POINTER(WgradJob), c_i32,
                                          c_vp, c_sz, c_vp]),
    "pgnn_pool_narrow_bwd_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_pool_narrow_bwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                         c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                         c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "pgnn_loss_fwd_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp,
                                  c_i64, c_i32, ctypes.c_float, ctypes.c_float,
                                  c_vp, c_vp, c_vp, c_vp]),
    "pgnn_loss_fwd_bwd_counts": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp,
                                         c_vp, c_i64, c_i32, ctypes.c_double,
                                         ctypes.c_double, c_vp, c_vp, c_vp,
                                         c_vp, c_vp]),
    "pgnn_loss_fwd_bwd_ex": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp,
                                     c_i64, c_i32, ctypes.c_float,
                                     ctypes.c_float, c_vp, ctypes.c_double,
                                     ctypes.c_double, c_i32, ctypes.c_float,
                                     ctypes.c_float, c_vp, c_vp, c_vp, c_vp,
                                     c_vp]),
    "pgnn_loss_fwd_bwd_sel": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp,
                                      c_vp, c_i64, c_i32, ctypes.c_float,
                                      ctypes.c_float, c_vp, ctypes.c_double,
                                      ctypes.c_double, c_i32, ctypes.c_float,
                                      ctypes.c_float, c_vp, c_vp, c_vp,
                                      ctypes.c_float, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_vp]),
    "pgnn_topk_mask_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_topk_mask_f32": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "pgnn_sgd_step": (c_i32, [c_vp, c_vp, c_vp, c_i64, ctypes.c_float,
                              ctypes.c_float, ctypes.c_float, c_vp]),
    "pgnn_optimizer_step": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64] +
                            [ctypes.c_float] * 6 + [c_vp]),
    "pgnn_l1_norm": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    # detection post-processing
    "pgnn_box_decode_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64,
                                    c_i32, c_vp, c_vp]),
    "pgnn_box_encode_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64,
                                    c_i32, c_vp, c_vp]),
    "pgnn_detection_candidates": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp,
                                          c_i64, c_vp, c_vp]),
    "pgnn_detection_candidates_dyn": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp,
                                              c_vp, c_i64, c_vp, c_vp]),
    "pgnn_nms_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_nms_boxes_3d": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64,
                                  c_f64, c_i32, ctypes.c_float, c_i64,
                                  c_vp, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp]),
    "pgnn_overlapped_boxes_3d": (c_i32, [c_vp, c_vp, c_i64, ctypes.c_float,
                                         c_vp, c_vp]),
    "pgnn_overlapped_boxes_3d_raster": (c_i32, [c_vp, c_vp, c_i64, c_vp,
                                                c_vp]),
    # training targets
    "pgnn_assign_box_labels": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp,
                                       c_vp, c_vp, c_vp]),
    "pgnn_assign_box_labels_f64": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp,
                                           c_vp, c_vp, c_vp, c_vp]),
    "pgnn_box_encode_f64": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64,
                                    c_i32, c_vp, c_vp]),
    "pgnn_box_encode_f64_xyz64": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64,
                                          c_i32, c_vp, c_vp]),
    "pgnn_points_affine_f64": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "pgnn_points_in_box_f64": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                       c_vp]),
    # KITTI frame ingest
    "pgnn_kitti_ingest_workspace_bytes": (c_sz, [c_i64]),
    "pgnn_kitti_cam_points_in_image": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_f64,
                                               c_f64, c_vp, c_i64, c_i64, c_vp,
                                               c_sz, c_vp, c_vp, c_i32, c_i64,
                                               c_vp, c_vp]),
    # streaming metrics
    "pgnn_metrics_state_bytes": (c_sz, [c_i32, c_i32]),
    "pgnn_metrics_update": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                    c_vp, c_vp]),
    "pgnn_metrics_compute": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
}

_lib = None


def exported_symbols():
    """Names every entry point include/pointgnn_hip.h declares."""
    return sorted(_SIGNATURES)


def _build_missing():
    """A source checkout without the built library (the .so is git-ignored):
    compile it once with hipcc if there is one.  Ranks of one node serialise
    on a lock file so that `torch.distributed.run` does not start N builds."""
    import fcntl
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return
    lock = os.path.join(os.path.dirname(LIB_PATH), ".build.lock")
    with open(lock, "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                from . import build as _build
                _build.build(verbose=False)
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def load():
    """Load (once) and return the ctypes library.  Raises when it is missing:
    build it with `python -m pointgnn_amd.build` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("PGNN_LIB"):
        _build_missing()
    if not os.path.exists(LIB_PATH):
        raise PointGnnHipError(
            "libpointgnn_hip.so not found at %s -- the HIP extension is "
            "required (no CPU fallback); run `python -m pointgnn_amd.build`"
            % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # PGNN_TUNE="key=value,key=value": tunables applied when the library is
    # loaded (pgnn_set_tunable) -- runs a whole test session under, e.g., the
    # frame pipeline's capped builder grids
    for kv in filter(None, os.environ.get("PGNN_TUNE", "").split(",")):
        k, v = kv.split("=")
        check(lib.pgnn_set_tunable(k.strip().encode(), int(v)),
              "PGNN_TUNE " + kv)
    return lib


E_INVALID = -1       # PGNN_E_* of include/pointgnn_hip.h
E_WORKSPACE = -2
E_UNSUPPORTED = -3


def check(rc, what=""):
    """Raise on a non-zero return code of a pgnn_* call."""
    if rc != 0:
        msg = load().pgnn_last_error()
        raise PointGnnHipError("%s failed (code %d): %s" % (
            what or "pgnn call", rc, msg.decode() if msg else "?"))


_SCHED_WS = {}
SCHED_WS_INTS = 64   # PGNN_SCHED_WS_INTS of include/pointgnn_hip.h


def sched_ws(device=None):
    """The zeroed int32 counters of the fused kernels' tile pools (sched_ws of
    pgnn_point_set_pooling_fwd / pgnn_edge_mlp_scatter_max_fwd;
    PGNN_SCHED_WS_INTS of them): one set per (device, current stream) --
    launches of one stream are serialised and the kernels hand the counters
    back zeroed."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) \
        if device is None or device.index is None else device
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _SCHED_WS.get(key)
    if t is None:
        t = _SCHED_WS[key] = torch.zeros(SCHED_WS_INTS, dtype=torch.int32,
                                         device=dev)
    return t


def stream_ptr():
    """hipStream_t of torch's current stream as void* (0 = null stream)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


class DeviceCount(object):
    """A size that exists only in device memory (capacity form, struct
    pgnn_dyn_count): `dev` is a 1-element int32 CUDA tensor, `hint` the value
    the host expects (it chooses between kernels with identical results and
    sizes strided grids, nothing else), `frame` the FrameCounts record it
    belongs to.  Tensors whose leading dimension is a capacity carry one as
    `t._pgnn_count` (tag_count / count_of)."""
    __slots__ = ("dev", "hint", "frame")

    def __init__(self, dev, hint=0, frame=None):
        self.dev = dev
        self.hint = int(hint)
        self.frame = frame

    def arg(self):
        """ctypes pgnn_dyn_count* for a *_dyn entry (valid during the call)."""
        return ctypes.byref(DynCount(self.dev.data_ptr(), self.hint))


def tag_count(t, count):
    """Mark tensor `t` as capacity-form: only the first `count` rows exist."""
    if count is not None:
        t._pgnn_count = count
    return t


def count_of(t):
    return getattr(t, "_pgnn_count", None) if t is not None else None


def set_tunable(key, value):
    check(load().pgnn_set_tunable(key.encode(), int(value)), "pgnn_set_tunable")
