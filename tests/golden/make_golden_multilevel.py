"""Fixture for MORE THAN ONE pooling level (graph_gen.py:49-90, :92-153 loop over
arbitrary `levels`; no shipped config uses it).  Run in the BUILD container:

    python tests/golden/make_golden_multilevel.py

Written by the REFERENCE's real `models/graph_gen.py` (imported by
tests/_refimport.py), with its `open3d` import served by a stand-in whose
`voxel_down_sample` is the oracle's restatement of open3d-python 0.7.0.0
(oracle/graph_oracle.voxel_centroids_open3d07 -- open3d is not installable
here; everything behind the centroids -- the scale loop, the sklearn
nearest-neighbour search into the PREVIOUS level's points, the index and
coordinate bookkeeping, the radius graphs -- is the reference's own code):

  * 'center': gen_multi_level_local_graph_v3(downsample_method='center') on a
    three-level configuration (scales 1, 2.5, 2.5 x base voxel 0.4 m: two
    pooling levels, then a same-scale GNN level): vertex coordinates, keypoint
    indices and edge lists of every level;
  * 'random': the same configuration with downsample_method='random', with
    and without add_rnd3d, NumPy / Python RNGs seeded -- the device draws from
    a counter RNG, so the tests compare voxel SETS (each keypoint's voxel, on
    the grid anchored at the ORIGINAL cloud's minimum) and edge lists for the
    device's own keypoints against the oracle.
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
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: E402
from oracle import graph_oracle as go  # noqa: E402

from _multilevel import BASE_VOXEL, LEVEL_CONFIGS  # noqa: E402


def install_open3d_stand_in():
    """open3d 0.7's three calls of graph_gen.py:41-45 on top of the oracle's
    restatement of voxel_down_sample."""
    o3d = sys.modules["open3d"]

    class PointCloud(object):
        points = None

    class _Down(object):
        def __init__(self, pts):
            self.points = pts

    o3d.PointCloud = PointCloud
    o3d.Vector3dVector = lambda a: np.asarray(a)

    def voxel_down_sample(pcd, voxel_size):
        cent, _ = go.voxel_centroids_open3d07(np.asarray(pcd.points), voxel_size)
        return _Down(cent)
    o3d.voxel_down_sample = voxel_down_sample


def main():
    gg = reference_graph_gen()
    assert gg is not None, "needs /root/reference"
    install_open3d_stand_in()
    out = {}
    for preset, seed in (("small", 0), ("tiny", 1)):
        xyz, _ = synthetic_cloud(seed=seed, preset=preset)
        out["%s_xyz" % preset] = xyz
        vc, ki, el = gg.gen_multi_level_local_graph_v3(
            xyz, BASE_VOXEL, LEVEL_CONFIGS, add_rnd3d=False,
            downsample_method='center')
        assert len(vc) == 4 and len(ki) == 3 and len(el) == 3
        for l in range(3):
            out["%s_center_coords%d" % (preset, l + 1)] = \
                np.asarray(vc[l + 1], np.float32)
            out["%s_center_kp%d" % (preset, l)] = np.asarray(ki[l], np.int32)
            out["%s_center_edges%d" % (preset, l)] = np.asarray(el[l], np.int32)
        for tag, rnd in (("rand", False), ("randjit", True)):
            np.random.seed(0)
            random.seed(0)
            vc, ki, el = gg.gen_multi_level_local_graph_v3(
                xyz, BASE_VOXEL, LEVEL_CONFIGS, add_rnd3d=rnd,
                downsample_method='random')
            for l in range(3):
                out["%s_%s_kp%d" % (preset, tag, l)] = np.asarray(ki[l], np.int32)
            out["%s_%s_counts" % (preset, tag)] = np.array(
                [len(v) for v in vc] + [len(e) for e in el], np.int64)
        print(preset, [len(v) for v in vc], "center K:",
              [out["%s_center_kp%d" % (preset, l)].shape[0] for l in range(3)],
              "E:", [out["%s_center_edges%d" % (preset, l)].shape[0]
                     for l in range(3)])
    np.savez_compressed(os.path.join(HERE, "graph_multilevel.npz"), **out)
    print("graph_multilevel.npz",
          os.path.getsize(os.path.join(HERE, "graph_multilevel.npz")), "bytes")


if __name__ == "__main__":
    main()
