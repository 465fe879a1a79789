"""The float64 training data path (train.py:88-130): the reference keeps the
augmented cloud float64 through graph generation, label assignment and box
encoding and casts last.  tests/golden/graph_f64.npz was written by the
reference's own graph_gen.py on a float64 cloud that carries points a few
float64 ulps either side of voxel faces and of the 1.0 m / 4.0 m spheres
(tests/golden/make_golden_f64.py); float32 rounding of that cloud gives a
different keypoint count and a different edge count (recorded in the fixture).
Bars: voxel sets and edge sets identical to the reference's."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from oracle import graph_oracle as go

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOXEL = 0.8


def gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from pointgnn_amd import _lib
    _lib.load()
    return torch.device("cuda")


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _vox(xyz, jitter):
    """graph_gen.py:123-128 with the reference's own NumPy expressions."""
    off = np.asarray([np.amin(xyz, axis=0)])
    if jitter is None:
        v = (xyz - off) // VOXEL
    else:
        v = (xyz - off + jitter[None, :]) // VOXEL
    assert v.dtype == np.float64
    return v.astype(np.int32)


def _canon(e):
    return go.canonical_edges(np.asarray(e))


@pytest.mark.parametrize("tag", ["rand", "randjit"])
def test_f64_random_keypoints_voxels_equal_reference(dev, tag):
    """One keypoint per reference voxel, on float64 coordinates: the probe
    pairs (two points a few ulps apart, in different voxels of this grid) and
    the radius-probe centres are alone in their voxels, so they must ALL be
    keypoints -- a device path that rounds the cloud to float32 merges every
    pair into one voxel (the fixture records K = 192 for it instead of 200)."""
    import torch
    from pointgnn_amd import graph_gen
    g = gold("graph_f64.npz")
    xyz = g["xyz"]
    assert xyz.dtype == np.float64
    jitter = g["jitter"] if tag == "randjit" else None
    vox = _vox(xyz, jitter)
    ref = g["ref_%s_kp_idx" % tag][:, 0]
    ref_vox = {tuple(v) for v in vox[ref]}
    assert len(ref_vox) == len(ref) and ref_vox == {tuple(v) for v in vox}
    c, i = graph_gen.keypoints_device(T(xyz, dev), VOXEL, 'random', jitter,
                                      seed=11)
    assert c.dtype == torch.float64      # vertex_coord_list stays float64
    i = i.cpu().numpy()[:, 0]
    assert len(i) == len(ref) != int(g["f32_rand_num_kp"])
    got = [tuple(v) for v in vox[i]]
    assert len(set(got)) == len(got) and set(got) == ref_vox
    assert np.array_equal(c.cpu().numpy(), xyz[i])      # exact coordinates
    pairs = g["probe_pairs_jit" if tag == "randjit" else "probe_pairs"]
    chosen = set(i.tolist())
    assert all(int(a) in chosen and int(b) in chosen for a, b in pairs)
    assert all(int(c0) in chosen for c0 in g["probe_centres"])


@pytest.mark.parametrize("tag", ["rand", "randjit"])
def test_f64_radius_graph_equals_reference(dev, tag):
    """The radius kernel on float64 points / centres (the reference's
    keypoints fed in as inputs): edge sets identical to what the reference's
    gen_multi_level_local_graph_v3 returned, including satellites 1 and 3 ulps
    inside / outside the sphere (float32 rounding gives 8090 level-0 edges
    instead of 8094)."""
    from pointgnn_amd import graph_gen
    g = gold("graph_f64.npz")
    xyz = g["xyz"]
    kp = xyz[g["ref_%s_kp_idx" % tag][:, 0]]
    e0 = graph_gen.gen_disjointed_rnn_local_graph_v3(xyz, kp, 1.0, -1)
    assert e0.dtype == np.int32 and np.all(np.diff(e0[:, 1]) >= 0)
    assert np.array_equal(_canon(e0), _canon(g["ref_%s_edges0" % tag]))
    if tag == "rand":
        assert len(e0) != int(g["f32_rand_num_edges0"])
    e1 = graph_gen.gen_disjointed_rnn_local_graph_v3(kp, kp, 4.0, -1)
    assert np.array_equal(_canon(e1), _canon(g["ref_%s_edges1" % tag]))
    # device tensors in (mixed precision: float32 centres are widened)
    e0d = graph_gen.gen_disjointed_rnn_local_graph_v3(
        T(xyz, dev), T(kp, dev), 1.0, -1)
    assert np.array_equal(_canon(e0d.cpu().numpy()), _canon(e0))
    # scale pre-division in float64 (graph_gen.py:203-206)
    sc = [1.0, 2.0, 0.5]
    es = graph_gen.gen_disjointed_rnn_local_graph_v3(kp, kp, 2.0, -1, scale=sc)
    assert np.array_equal(_canon(es), go.radius_graph_c(kp, kp, 2.0, scale=sc))


@pytest.mark.parametrize("rnd", [False, True])
def test_f64_multi_level_call(dev, rnd):
    """gen_multi_level_local_graph_v3 on the float64 cloud: float64 vertex
    lists back (train.py:124 casts them afterwards), one keypoint per
    reference voxel, and both levels' edges equal the float64 predicate on
    the keypoints this run chose."""
    from pointgnn_amd import graph_gen
    g = gold("graph_f64.npz")
    xyz = g["xyz"]
    kw = dict(configs.car_auto_config(3)["graph_gen_kwargs"])
    kw["add_rnd3d"] = rnd
    kw["level_configs"] = [dict(c, graph_gen_kwargs=dict(
        c["graph_gen_kwargs"], num_neighbors=-1)) for c in kw["level_configs"]]
    np.random.seed(0)       # the jitter is the first draw, as in the fixture
    coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3(xyz, **kw)
    assert [c.dtype for c in coords] == [np.float64] * 3
    i = kps[0][:, 0]
    assert np.array_equal(coords[1], xyz[i])
    vox = _vox(xyz, g["jitter"] if rnd else None)
    tag = "randjit" if rnd else "rand"
    assert {tuple(v) for v in vox[i]} == \
        {tuple(v) for v in vox[g["ref_%s_kp_idx" % tag][:, 0]]}
    assert len(i) == len(g["ref_%s_kp_idx" % tag])
    assert np.array_equal(_canon(edges[0]),
                          go.radius_graph_c(xyz, coords[1], 1.0))
    assert np.array_equal(_canon(edges[1]),
                          go.radius_graph_c(coords[1], coords[1], 4.0))


def test_f64_center_mode_needs_float32_representable(dev):
    """'center' keypoints go through the float32-keyed kd-tree replica: a
    float64 cloud holding float32 values is the same computation, a genuine
    float64 cloud has no device path and says so."""
    import torch
    from pointgnn_amd import graph_gen
    from pointgnn_amd.synthetic import synthetic_cloud
    x32, _ = synthetic_cloud(seed=1, preset="tiny")
    c32, i32 = graph_gen.keypoints_device(T(x32, dev), 0.4, 'center')
    c64, i64 = graph_gen.keypoints_device(T(x32.astype(np.float64), dev), 0.4,
                                          'center')
    assert c64.dtype == torch.float64 and torch.equal(i32, i64)
    assert torch.equal(c64, c32.to(torch.float64))
    with pytest.raises(NotImplementedError):
        graph_gen.keypoints_device(T(gold("graph_f64.npz")["xyz"], dev), 0.4,
                                   'center')


METHODS = {"yaw": (8, "assign_classaware_label_to_points"),
           "Car": (4, "assign_classaware_car_label_to_points"),
           "Pedestrian_and_Cyclist": (
               6, "assign_classaware_ped_and_cyc_label_to_points")}


@pytest.mark.parametrize("method", list(METHODS))
def test_f64_labels_and_box_encoding(dev, method):
    """Label assignment and box encoding on FLOAT64 vertices (train.py:100-122
    hands them the float64 vertex_coord_list) against the fixture the
    reference's own kitti_dataset.py / box_encoding.py wrote
    (make_golden_f64.py `labels_main`): 240 of the vertices sit 1e-12 or 3e-12 m either side
    of a box face; rounding the vertices to float32 changes ~100 labels (count
    recorded in the fixture), the float64 path must reproduce every one."""
    from pointgnn_amd import box_encoding as BE, kitti_dataset as KD
    from oracle import labels_oracle as LO
    fix = gold("labels_f64.npz")
    xyz = fix["xyz"]
    assert xyz.dtype == np.float64
    labels = LO.synthetic_labels(0, LO.synthetic_vertices(0),
                                 n_boxes=int(fix["n_labels"]))
    nc, fn = METHODS[method]
    ds = object.__new__(KD.KittiDataset)
    ds.num_classes = nc
    cls, boxes, valid, lm = getattr(ds, fn)(labels, xyz, (1.0, 1.0, 1.0))
    assert np.array_equal(cls, fix[method + "_cls"])
    assert np.array_equal(boxes, fix[method + "_boxes"])
    assert np.array_equal(valid, fix[method + "_valid"])
    cls32 = getattr(ds, fn)(labels, xyz.astype(np.float32),
                            (1.0, 1.0, 1.0))[0]
    assert int((cls32 != cls).sum()) == int(fix[method + "_n_diff_f32"]) > 0
    enc = BE.get_box_encoding_fn('classaware_all_class_box_encoding')(
        cls, xyz, boxes, lm)
    ref = fix[method + "_encoded"]
    assert enc.dtype == np.float32 and enc.shape == ref.shape
    tol = np.spacing(np.maximum(np.abs(enc), np.abs(ref)))
    assert np.all(np.abs(enc.astype(np.float64) - ref) <= tol)
    # device tensors in -> device tensors out, same values
    import torch
    cd, bd, vd, _ = getattr(ds, fn)(labels, T(xyz, dev), (1.0, 1.0, 1.0))
    assert torch.equal(cd.cpu().reshape(-1).long(),
                       torch.from_numpy(cls).reshape(-1))
    ed = BE.classaware_all_class_box_encoding(cd, T(xyz, dev), bd, lm)
    assert np.array_equal(ed.cpu().numpy(), enc)
