"""Generates tests/golden/fullsize_*.npz: the reference's OWN outputs at the
BASELINE sizes (20 000-point `car` / `car_600k`, 50 000-point `ped_dense`, all
seed 0).  Run in the BUILD container (reads /root/reference):

    python tests/golden/make_golden_fullsize.py

Two things are pinned per frame:

  * EDGE LISTS.  The reference's real `gen_disjointed_rnn_local_graph_v3`
    (/root/reference/models/graph_gen.py:197-220, imported under tensorflow /
    open3d stubs, tests/_refimport.py) at both levels of the shipped inference
    kwargs, on the centre-mode keypoints the real sklearn kd-tree call of
    graph_gen.py:84-88 returns (oracle/graph_oracle.keypoints_center; only
    open3d's voxel means are restated).  A 600 k-row list is too big to
    commit, so it is stored as (E, sha256) in two forms:
      - `edges{L}_sha`: rows (src, dst) in the fixture's keypoint numbering,
        (dst, src)-sorted, int32 little-endian -- for a device build fed the
        fixture's keypoints;
      - `edges{L}_pt_sha`: rows (point index of dst keypoint, point index of
        src) -- independent of how keypoints are numbered (the reference's
        order is open3d's hash-map order, the device's its voxel-hash order),
        for the device's own end-to-end build.  Level 1's src is a keypoint
        too and is mapped the same way.
    The per-keypoint fan-in histogram digest is kept as a cheaper diagnostic.

  * CONFIG 2 AT ITS OWN SIZE WITH THE TRAINED WEIGHTS (car frames only).  The
    reference's serialized graphs `checkpoints/car_auto_T{0,1}_train/
    model-1400000.meta` (what run.py:199-201 restores and run.py:252-260
    evaluates), with the REAL blobs, evaluated by oracle/tf_meta_interp.py on
    that reference-built graph: logits [K,4] and box encodings [K,4,7].
"""
import glob
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _refimport import reference_graph_gen  # noqa: E402
import pointgnn_amd  # noqa: E402,F401
from pointgnn_amd import configs  # noqa: E402
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
from oracle import graph_oracle as go  # noqa: E402

FRAMES = (("car", "car_auto_T3", True), ("car_600k", "car_auto_T3", True),
          ("ped_dense", "ped_cyl_auto_T3", False))


def edge_digest(src, dst):
    """sha256 of the (dst, src)-sorted [E,2] int32 (src, dst) rows."""
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    order = np.lexsort((src, dst))
    rows = np.stack([src[order], dst[order]], axis=1).astype("<i4")
    return hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest()


def fanin_digest(dst, k):
    h = np.bincount(np.asarray(dst, np.int64), minlength=k).astype("<i4")
    return hashlib.sha256(np.sort(h).tobytes()).hexdigest()


def main():
    gg = reference_graph_gen()
    assert gg is not None, "needs /root/reference"
    from oracle import tf_graph_ref as tg
    for preset, cfg_name, real_weights in FRAMES:
        t0 = time.time()
        cfg = configs.get_config(cfg_name)
        rk = cfg["runtime_graph_gen_kwargs"]
        voxel = rk["base_voxel_size"] * rk["level_configs"][0]["graph_scale"]
        radii = [lc["graph_gen_kwargs"]["radius"] for lc in rk["level_configs"]]
        xyz, inten = synthetic_cloud(seed=0, preset=preset)
        kp_xyz, kp_idx = go.keypoints_center(xyz, xyz, voxel)
        kp = kp_idx[:, 0].astype(np.int64)
        k = len(kp)
        e0 = gg.gen_disjointed_rnn_local_graph_v3(
            xyz, kp_xyz, radius=radii[0], num_neighbors=-1)
        e1 = gg.gen_disjointed_rnn_local_graph_v3(
            kp_xyz, kp_xyz, radius=radii[1], num_neighbors=-1)
        out = {
            "preset": np.array(preset), "seed": np.int32(0),
            "n_points": np.int32(len(xyz)), "voxel": np.float64(voxel),
            "radii": np.asarray(radii, np.float64),
            "xyz_sha": np.array(hashlib.sha256(
                np.ascontiguousarray(xyz.astype("<f4")).tobytes()).hexdigest()),
            "kp_idx": kp_idx.astype(np.int32),
            "E0": np.int64(len(e0)), "E1": np.int64(len(e1)),
            "edges0_sha": np.array(edge_digest(e0[:, 0], e0[:, 1])),
            "edges1_sha": np.array(edge_digest(e1[:, 0], e1[:, 1])),
            "edges0_pt_sha": np.array(edge_digest(e0[:, 0], kp[e0[:, 1]])),
            "edges1_pt_sha": np.array(edge_digest(kp[e1[:, 0]], kp[e1[:, 1]])),
            "fanin0_sha": np.array(fanin_digest(e0[:, 1], k)),
            "fanin1_sha": np.array(fanin_digest(e1[:, 1], k)),
        }
        print("%s: N %d K %d E0 %d E1 %d (graph %.1f s)" % (
            preset, len(xyz), k, len(e0), len(e1), time.time() - t0))
        if real_weights:
            for t in (0, 1):
                t1 = time.time()
                name = "car_auto_T%d" % t
                meta = glob.glob("/root/reference/checkpoints/%s_train/"
                                 "model-*.meta" % name)[0]
                step = int(meta.rsplit("model-", 1)[1].split(".")[0])
                ref = tg.ReferenceGraph(meta)
                w = dict(np.load(os.path.join(HERE, "weights_%s.npz" % name)))
                ref.set_weights(w, global_step=step)
                lg, bx, _ = ref.predict(
                    features=inten, coords=[xyz, kp_xyz, kp_xyz],
                    keypoints=[kp_idx.astype(np.int32),
                               np.arange(k, dtype=np.int32).reshape(-1, 1)],
                    edges=[e0.astype(np.int32), e1.astype(np.int32)])
                out["T%d_logits" % t] = np.asarray(lg, np.float32)
                out["T%d_box_encodings" % t] = np.asarray(bx, np.float32)
                print("  %s real weights: logits %s |max| %.3g, boxes |max| "
                      "%.3g (%.1f s)" % (name, lg.shape, np.abs(lg).max(),
                                         np.abs(bx).max(), time.time() - t1))
        path = os.path.join(HERE, "fullsize_%s.npz" % preset)
        np.savez_compressed(path, **out)
        print("  -> %s (%d KB)" % (os.path.basename(path),
                                  os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
