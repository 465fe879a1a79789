"""Generates the committed golden fixtures.  Run in the BUILD container (it reads
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What is produced, and from what:
  * graph_tiny.npz / graph_small.npz -- inputs (seeded synthetic cloud) and the
    outputs of the REFERENCE's real `models/graph_gen.py` functions (imported
    under tensorflow/open3d stubs, tests/_refimport.py):
      - gen_disjointed_rnn_local_graph_v3 at both levels (keypoints chosen by
        the oracle's open3d-0.7 restatement, passed in as inputs),
      - gen_multi_level_local_graph_v3(downsample_method='random') with
        numpy/python RNGs seeded to 0, with and without add_rnd3d,
      - gen_disjointed_rnn_local_graph_v3 with `scale` and with the fan-in cap.
  * weights_car_auto_T0.npz / weights_car_auto_T1.npz -- the reference's
    trained variables (checkpoints/car_auto_T{0,1}_train) read by
    point-gnn_amd/tf_bundle.py, under their TF names.
  * logits_car_auto_T{0,1}_tiny.npz -- oracle/gnn_oracle.py outputs (float32 and
    float64 shadow) for those weights on the tiny cloud's centre-mode graph.
    (TensorFlow is not installable here: the GNN arithmetic itself is "parity
    unpinned", see oracle/gnn_oracle.py.)
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _refimport import reference_graph_gen  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs, tf_bundle  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
from oracle import graph_oracle as go  # noqa: E402
from oracle import gnn_oracle as gn  # noqa: E402


def graph_fixture(gg, preset, seed, name):
    xyz, inten = synthetic_cloud(seed=seed, preset=preset)
    cfg = configs.car_auto_config(3)
    out = {"xyz": xyz, "intensity": inten}
    # keypoints (centre mode) from the oracle; radius graphs from the reference
    kp_xyz, kp_idx = go.keypoints_center(xyz, xyz, 0.8 * 0.5)
    out["kp_xyz"] = kp_xyz
    out["kp_idx"] = kp_idx.astype(np.int32)
    out["ref_edges0"] = gg.gen_disjointed_rnn_local_graph_v3(
        xyz, kp_xyz, radius=1.0, num_neighbors=-1).astype(np.int32)
    out["ref_edges1"] = gg.gen_disjointed_rnn_local_graph_v3(
        kp_xyz, kp_xyz, radius=4.0, num_neighbors=-1).astype(np.int32)
    out["ref_edges1_scaled"] = gg.gen_disjointed_rnn_local_graph_v3(
        kp_xyz, kp_xyz, radius=2.0, num_neighbors=-1,
        scale=[1.0, 2.0, 0.5]).astype(np.int32)
    np.random.seed(0)
    out["ref_edges1_cap64"] = gg.gen_disjointed_rnn_local_graph_v3(
        kp_xyz, kp_xyz, radius=4.0, num_neighbors=64).astype(np.int32)
    # full multi-level call, training ('random') mode, seeded
    for tag, rnd in (("rand", False), ("randjit", True)):
        np.random.seed(0)
        random.seed(0)
        kw = dict(cfg["graph_gen_kwargs"])
        kw["add_rnd3d"] = rnd
        vc, ki, el = gg.gen_multi_level_local_graph_v3(xyz, **kw)
        out["ref_%s_kp_idx" % tag] = ki[0].astype(np.int32)
        out["ref_%s_edges0" % tag] = el[0].astype(np.int32)
        out["ref_%s_edges1" % tag] = el[1].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()})
    return out


def main():
    gg = reference_graph_gen()
    assert gg is not None, "needs /root/reference"
    tiny = graph_fixture(gg, "tiny", 1, "graph_tiny.npz")
    graph_fixture(gg, "small", 0, "graph_small.npz")
    for t in (0, 1):
        ck = tf_bundle.load_checkpoint(
            "/root/reference/checkpoints/car_auto_T%d_train" % t)
        w = {k: v for k, v in ck.items() if k != "Variable"}
        np.savez_compressed(
            os.path.join(HERE, "weights_car_auto_T%d.npz" % t), **w)
        cfg = configs.car_auto_config(t)
        coords = [tiny["xyz"], tiny["kp_xyz"], tiny["kp_xyz"]]
        k = tiny["kp_xyz"].shape[0]
        kps = [tiny["kp_idx"], np.arange(k, dtype=np.int32).reshape(-1, 1)]
        edges = [tiny["ref_edges0"], tiny["ref_edges1"]]
        lg32, bx32 = gn.predict(w, cfg, tiny["intensity"], coords, kps, edges,
                                dtype=np.float32)
        lg64, bx64 = gn.predict(w, cfg, tiny["intensity"], coords, kps, edges,
                                dtype=np.float64)
        np.savez_compressed(
            os.path.join(HERE, "logits_car_auto_T%d_tiny.npz" % t),
            logits32=lg32, boxes32=bx32, logits64=lg64, boxes64=bx64)
        print("T%d" % t, lg32.shape, bx32.shape,
              "fp32-fp64 max", np.abs(lg32 - lg64).max())


if __name__ == "__main__":
    main()
