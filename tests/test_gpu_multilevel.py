"""More than one POOLING level (graph_gen.py:49-90 and :92-153 loop over
arbitrary `levels`; every shipped config has exactly one): scales 1, 2.5, 2.5
(tests/_multilevel.py).  Against tests/golden/graph_multilevel.npz, written by
the reference's real gen_multi_level_local_graph_v3
(tests/golden/make_golden_multilevel.py), and against the oracle.

The device numbers a level's keypoints in its voxel-hash order, the reference
in open3d's / dict order, so lists are compared in numbering-free form: a
level's vertices as the (multi)set of ORIGINAL point indices they stand for,
edges through those indices."""
import os
import random

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd.synthetic import synthetic_cloud  # noqa: F401
from oracle import graph_oracle as go
from oracle import gnn_oracle as gn
from _multilevel import BASE_VOXEL, LEVEL_CONFIGS, model_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FP_TOL = 2e-4   # north-star tolerance is 1e-3


def gold():
    return dict(np.load(os.path.join(GOLD, "graph_multilevel.npz")))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from pointgnn_amd import _lib
    _lib.load()            # raises if the HIP extension is missing
    return torch.device("cuda")


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def point_ids(kps):
    """Original point index of every vertex of every level >= 1."""
    ids = [np.asarray(kps[0])[:, 0].astype(np.int64)]
    for k in kps[1:]:
        ids.append(ids[-1][np.asarray(k)[:, 0].astype(np.int64)])
    return ids


def edges_by_point(edges, src_ids, dst_ids):
    e = np.asarray(edges).astype(np.int64)
    return go.canonical_edges(np.stack([src_ids[e[:, 0]], dst_ids[e[:, 1]]], 1))


@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_second_pooling_level_center_entry(dev, preset):
    """pgnn_voxel_keypoints_center_from with the REFERENCE's first-level
    vertices as the search set: the same indices into that set (as a multiset:
    two centroids may share their nearest vertex, and the reference keeps
    both), the same coordinates."""
    from pointgnn_amd import graph_gen
    g = gold()
    xyz = g["%s_xyz" % preset]
    base = g["%s_center_coords1" % preset]
    ref = g["%s_center_kp1" % preset][:, 0]
    c, i = graph_gen.keypoints_device(T(xyz, dev), BASE_VOXEL * 2.5, 'center',
                                      base=T(base, dev))
    i = i.cpu().numpy()[:, 0]
    assert np.array_equal(np.sort(i), np.sort(ref))
    assert np.array_equal(c.cpu().numpy(), base[i])


@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_multi_level_center_end_to_end(dev, preset):
    """gen_multi_level_local_graph_v3('center') end to end, NumPy in -> NumPy
    out like the reference.  Level 1: the reference's vertex set (fixture).
    Level 2 and up: an exact distance tie (the two points of a 2-point voxel)
    goes to the point scikit-learn's query meets first in the kd-tree of the
    previous level's vertices -- which depends on the ORDER of those vertices,
    open3d's hash-map order in the reference and the device's voxel-hash order
    here -- so the level is held to the reference's own code path on the
    device's level-1 array: sklearn's kneighbors (oracle.keypoints_center),
    index for index as a multiset; away from ties it is the fixture's set."""
    from pointgnn_amd import graph_gen
    g = gold()
    xyz = g["%s_xyz" % preset]
    coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3(
        xyz, BASE_VOXEL, LEVEL_CONFIGS, add_rnd3d=False,
        downsample_method='center')
    assert isinstance(coords[1], np.ndarray) and len(coords) == 4
    ref_kps = [g["%s_center_kp%d" % (preset, l)] for l in range(3)]
    ids, ref_ids = point_ids(kps), point_ids(ref_kps)
    assert np.array_equal(np.sort(ids[0]), np.sort(ref_ids[0]))
    _, want = go.keypoints_center(xyz, coords[1], BASE_VOXEL * 2.5)
    assert np.array_equal(np.sort(kps[1][:, 0]), np.sort(want[:, 0]))
    same = len(set(ids[1]) & set(ref_ids[1]))
    # (the sparse 1 500-point cloud has many 2-point voxels: ~7 % ties)
    assert same >= 0.9 * len(ref_ids[1]) and len(ids[1]) == len(ref_ids[1])
    for l in range(3):
        assert np.array_equal(coords[l + 1], xyz[ids[l]])
    # the same-scale level keeps its vertices: identity keypoints
    assert np.array_equal(kps[2][:, 0], np.arange(len(coords[2])))
    for l, cfg in enumerate(LEVEL_CONFIGS):
        lvl = cfg['graph_level']
        ref = go.radius_graph_c(coords[lvl], coords[lvl + 1],
                                cfg['graph_gen_kwargs']['radius'])
        assert np.array_equal(go.canonical_edges(edges[l]), ref), l
    # level 0's list is the reference's (numbering-free form)
    all_ids = [np.arange(len(xyz))] + ids
    ref_all = [np.arange(len(xyz))] + ref_ids
    assert np.array_equal(
        edges_by_point(edges[0], all_ids[0], all_ids[1]),
        edges_by_point(g["%s_center_edges0" % preset], ref_all[0], ref_all[1]))
    print("%s: K %s, E %s; level 2 shares %d of %d vertices with the "
          "reference's run (ties follow the level-1 order)" % (
              preset, [len(c) for c in coords[1:]], [len(e) for e in edges],
              same, len(ref_ids[1])))


def _voxels(p, origin, voxel, jitter=None):
    """NumPy's arithmetic of graph_gen.py:121-129 on float32 rows."""
    q = (p - origin) if jitter is None else (p - origin + jitter)
    return (q // voxel).astype(np.int32)


@pytest.mark.parametrize("rnd", [False, True])
def test_multi_level_random_pooling(dev, rnd):
    """'random' keypoints, two pooling levels.  Level 1 against the reference's
    voxel set (fixture, RNGs seeded alike: the jitters are the first NumPy
    draws, in level order).  Level 2 depends on which members level 1 picked
    (the device draws from a counter RNG, the reference from Python's
    `random`), so it is held to the definition on the device's own level-1
    vertices: exactly one keypoint, a level-1 vertex, in every cell they
    occupy on the grid anchored at the ORIGINAL cloud's minimum."""
    from pointgnn_amd import graph_gen
    g = gold()
    xyz = g["tiny_xyz"]
    tag = "randjit" if rnd else "rand"
    np.random.seed(0)
    random.seed(0)
    coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3(
        xyz, BASE_VOXEL, LEVEL_CONFIGS, add_rnd3d=rnd,
        downsample_method='random')
    np.random.seed(0)
    j1 = j2 = None
    v1, v2 = BASE_VOXEL * 1, BASE_VOXEL * 2.5
    if rnd:
        j1 = v1 * np.random.random((1, 3))
        j2 = v2 * np.random.random((1, 3))
    origin = np.asarray([np.amin(xyz, axis=0)])
    k1 = kps[0][:, 0].astype(np.int64)
    ref1 = g["tiny_%s_kp0" % tag][:, 0].astype(np.int64)
    cells = _voxels(xyz[k1], origin, v1, j1)
    assert len(np.unique(cells, axis=0)) == len(k1)
    assert np.array_equal(np.unique(cells, axis=0),
                          np.unique(_voxels(xyz[ref1], origin, v1, j1), axis=0))
    assert np.array_equal(coords[1], xyz[k1])
    # level 2: one keypoint per cell occupied by the level-1 vertices
    k2 = kps[1][:, 0].astype(np.int64)
    assert k2.min() >= 0 and k2.max() < len(k1)
    assert np.array_equal(coords[2], coords[1][k2])
    c_all = _voxels(coords[1], origin, v2, j2)
    c_sel = _voxels(coords[2], origin, v2, j2)
    assert len(np.unique(c_sel, axis=0)) == len(k2)
    assert np.array_equal(np.unique(c_sel, axis=0), np.unique(c_all, axis=0))
    assert len(k2) < len(k1)
    # the reference's own level-2 count is that of ITS level-1 choice: close
    assert abs(len(k2) - int(g["tiny_%s_counts" % tag][2])) <= 0.15 * len(k2)
    # edges of every level: the oracle on the device's vertices
    for l, cfg in enumerate(LEVEL_CONFIGS):
        lvl = cfg['graph_level']
        ref = go.radius_graph_c(coords[lvl], coords[lvl + 1],
                                cfg['graph_gen_kwargs']['radius'])
        assert np.array_equal(go.canonical_edges(edges[l]), ref), l


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_anchor_of_the_second_random_level_is_the_cloud(dev, dtype):
    """graph_gen.py:108-110: the level-2 grid starts at the minimum of the
    ORIGINAL cloud, not of the level-1 vertices -- a cloud whose minimum point
    is not a level-1 vertex tells the two apart.  float64: the training path's
    clouds (train.py:88-90), NumPy's float64 `//`."""
    from pointgnn_amd import graph_gen
    rng = np.random.default_rng(3)
    xyz = rng.uniform(0.0, 6.0, (3000, 3)).astype(dtype)
    # one cell of level 1 holds the cloud's minimum AND other points: most
    # seeds pick another member, so min(level-1 vertices) > min(cloud)
    xyz[0] = [0.0, 0.0, 0.0]
    xyz[1:40] = rng.uniform(0.05, 0.39, (39, 3)).astype(dtype)
    moved = 0
    for seed in range(8):
        np.random.seed(seed)
        coords, kps = graph_gen.multi_layer_downsampling_random(
            xyz, BASE_VOXEL, levels=[1, 2.5])
        origin = np.asarray([np.amin(xyz, axis=0)])
        c_all = _voxels(coords[1], origin, BASE_VOXEL * 2.5)
        c_sel = _voxels(coords[2], origin, BASE_VOXEL * 2.5)
        assert len(np.unique(c_sel, axis=0)) == len(c_sel)
        assert np.array_equal(np.unique(c_sel, axis=0), np.unique(c_all, axis=0))
        assert coords[1].dtype == dtype and coords[2].dtype == dtype
        assert np.array_equal(coords[2], coords[1][kps[1][:, 0]])
        moved += int(np.any(np.amin(coords[1], axis=0) > origin[0]))
    assert moved > 0, "the level-1 vertices always contained the minimum"


def test_far_search_set_falls_back_to_every_point(dev):
    """A search set farther than three voxels from a centroid (never the case
    for a previous level's keypoints, but the entry takes any set): the exact
    nearest point by the every-point pass."""
    from pointgnn_amd import graph_gen
    rng = np.random.default_rng(5)
    xyz = rng.uniform(0.0, 4.0, (800, 3)).astype(np.float32)
    base = (rng.uniform(0.0, 4.0, (50, 3)) + [30.0, -20.0, 9.0]).astype(np.float32)
    base[7] = base[3]          # an exact tie: the kd order of `base` decides
    c, i = graph_gen.keypoints_device(T(xyz, dev), 1.0, 'center',
                                      base=T(base, dev))
    ref_c, ref_i = go.keypoints_center(xyz, base, 1.0)
    assert np.array_equal(np.sort(i.cpu().numpy()[:, 0]), np.sort(ref_i[:, 0]))


@pytest.mark.parametrize("first_width", [300, 8])
def test_predict_on_two_pooling_levels(dev, first_width):
    """MultiLayerFastLocalGraphModelV2.predict (models.py:79-163) on that
    graph: PointSetPooling at levels 0 AND 1 (the second one pools 300-wide
    features -- or 8-wide ones, through the fused narrow-feature kernel on the
    first level's zero-padded [K, 16] rows), one GraphNetAutoCenter iteration,
    the predictor -- against the float64 oracle, per layer."""
    from pointgnn_amd import graph_gen, models, weights
    g = gold()
    xyz = g["small_xyz"]
    _, inten = synthetic_cloud(seed=0, preset="small")
    cfg = model_config(first_width=first_width)
    params = weights.init_params(cfg, seed=2, bias_scale=0.05)
    coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3(
        T(xyz, dev), BASE_VOXEL, LEVEL_CONFIGS, downsample_method='center')
    model = models.get_model(cfg["model_name"])(
        num_classes=cfg["num_classes"], box_encoding_len=7, mode="test",
        **cfg["model_kwargs"])
    model.load_state_dict(params)
    model.keep_features = True
    logits, boxes = model.predict(T(inten, dev), coords, kps, edges, False)
    c_np = [c.cpu().numpy() for c in coords]
    k_np = [k.cpu().numpy() for k in kps]
    e_np = [e.cpu().numpy() for e in edges]
    lg, bx, feats = gn.predict(params, cfg, inten, c_np, k_np, e_np,
                               dtype=np.float64, return_features=True)
    report = []
    for n, (got, ref) in enumerate(zip(model.feature_list, feats[1:])):
        got = got.cpu().numpy()[:, :ref.shape[1]]
        assert got.shape == ref.shape
        report.append("layer%d %.2g" % (n + 1, np.abs(got - ref).max()))
        np.testing.assert_allclose(got, ref, atol=FP_TOL, rtol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), lg, atol=FP_TOL, rtol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), bx, atol=FP_TOL, rtol=0)
    print("two pooling levels, K %s: %s, max|dlogit| %.3g" % (
        [len(c) for c in c_np[1:]], ", ".join(report),
        np.abs(logits.cpu().numpy() - lg).max()))


@pytest.mark.parametrize("first_width,sparse", [(300, True), (300, False),
                                                (8, True)])
def test_training_step_through_two_pooling_levels(dev, first_width, sparse):
    """The training step (models.py:170-311, train.py:225-234 differentiate
    whatever models.py:119-149 builds) on the two-pooling-level model: every
    variable's gradient of (cls + loc) against float64 autograd
    (oracle/train_oracle.py), for the sparse and the dense scatter-max
    adjoint, and with a narrow first level.  The second stage's backward goes
    through the gather's adjoint (scatter-add of the edge rows' gradient to the
    previous level's vertices)."""
    from pointgnn_amd import graph_gen, train, weights
    from oracle import train_oracle as to
    g = gold()
    xyz = g["small_xyz"]
    _, inten = synthetic_cloud(seed=0, preset="small")
    cfg = model_config(first_width=first_width)
    params = weights.init_params(cfg, seed=4, bias_scale=0.1)
    coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3(
        T(xyz, dev), BASE_VOXEL, LEVEL_CONFIGS, downsample_method='center')
    c_np = [c.cpu().numpy() for c in coords]
    k_np = [k.cpu().numpy() for k in kps]
    e_np = [e.cpu().numpy() for e in edges]
    k = len(c_np[-1])
    rng = np.random.default_rng(9)
    labels = rng.integers(0, cfg["num_classes"], (k, 1)).astype(np.int32)
    boxes = rng.standard_normal((k, 1, 7)).astype(np.float32)
    valid = (rng.random((k, 1, 1)) < 0.5).astype(np.float32)
    batch = (inten, c_np, k_np, e_np, labels, boxes, valid)
    tr = train.Trainer(cfg, params=params, device=dev)
    assert not tr.native        # the native step orchestrates ONE pooling stage
    tr.sparse_adjoint = sparse
    out = tr.train_step(batch, apply=False)
    loss, g_ref, _ = to.step_gradients(params, cfg, [batch])
    assert abs(out['cls_loss'] - loss['cls_loss']) < 1e-4 * max(1, loss['cls_loss'])
    assert abs(out['loc_loss'] - loss['loc_loss']) < 1e-4 * max(1, loss['loc_loss'])
    got = tr.grad_dict()
    worst = (0.0, 0.0, "")
    for n, ref in g_ref.items():
        scale = np.abs(ref).max() + 1e-12
        e_max = np.abs(got[n] - ref).max() / scale
        e_fro = np.linalg.norm(got[n] - ref) / (np.linalg.norm(ref) + 1e-12)
        worst = max(worst, (e_fro, e_max, n))
        assert e_fro < 2e-3, "%s: Frobenius rel err %.3g" % (n, e_fro)
        assert e_max < 1.2e-2, "%s: max-entry rel err %.3g" % (n, e_max)
    # the first level's variables get their gradient through the second
    # level's gather: not zero, not garbage
    first = [n for n in g_ref if n.startswith("layer1/")]
    assert first and all(np.abs(got[n]).max() > 0 for n in first)
    print("two pooling levels (first width %d, %s adjoint): worst Frobenius "
          "%.3g (max-entry %.3g) at %s" % (
              first_width, "sparse" if sparse else "dense", worst[0], worst[1],
              worst[2]))


def test_capacity_form_refuses_a_second_pooling_level(dev):
    from pointgnn_amd import graph_gen
    g = gold()
    hints = graph_gen.CountHints(k=2000, edges=[100000, 100000, 100000])
    with pytest.raises(NotImplementedError):
        graph_gen.gen_multi_level_local_graph_v3(
            T(g["tiny_xyz"], dev), BASE_VOXEL, LEVEL_CONFIGS,
            downsample_method='center', deferred_counts=hints)
