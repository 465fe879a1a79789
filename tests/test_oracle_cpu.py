"""Pins the oracle (oracle/*.py, oracle/*.c) against the committed golden
fixtures, which hold outputs of the REFERENCE's real graph_gen.py run in the
build container (tests/golden/make_golden.py), and against the reference
itself when /root/reference is present.  No GPU."""
import os
import random

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from pointgnn_amd.synthetic import synthetic_cloud
from oracle import graph_oracle as go
from oracle import gnn_oracle as gn
from _refimport import reference_graph_gen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("fixture", ["graph_tiny.npz", "graph_small.npz"])
def test_radius_graph_oracles_match_reference_golden(fixture):
    g = gold(fixture)
    xyz, kp = g["xyz"], g["kp_xyz"]
    for pts, ctr, r, key in ((xyz, kp, 1.0, "ref_edges0"),
                             (kp, kp, 4.0, "ref_edges1")):
        ref = go.canonical_edges(g[key])
        # same third-party call, identical row order expected
        assert np.array_equal(go.radius_graph_sklearn(pts, ctr, r), g[key])
        assert np.array_equal(go.canonical_edges(
            go.radius_graph_bruteforce(pts, ctr, r)), ref)
        assert np.array_equal(go.radius_graph_c(pts, ctr, r), ref)


def test_radius_graph_scale_and_cap_golden():
    g = gold("graph_tiny.npz")
    kp = g["kp_xyz"]
    ref = go.canonical_edges(g["ref_edges1_scaled"])
    got = go.radius_graph_c(kp, kp, 2.0, scale=[1.0, 2.0, 0.5])
    assert np.array_equal(got, ref)
    np.random.seed(0)
    capped = go.radius_graph_sklearn(kp, kp, 4.0, num_neighbors=64)
    assert np.array_equal(capped, g["ref_edges1_cap64"])
    # the cap keeps a subset of the full neighbour set, <= 64 per centre
    full = set(map(tuple, g["ref_edges1"]))
    assert all(tuple(e) in full for e in capped)
    assert np.bincount(capped[:, 1]).max() <= 64


@pytest.mark.parametrize("tag,rnd", [("rand", False), ("randjit", True)])
def test_multi_level_random_mode_golden(tag, rnd):
    g = gold("graph_tiny.npz")
    kw = dict(configs.car_auto_config(3)["graph_gen_kwargs"])
    kw["add_rnd3d"] = rnd
    np.random.seed(0)
    random.seed(0)
    coords, kps, edges = go.multi_level_graph(g["xyz"], **kw)
    assert np.array_equal(kps[0].astype(np.int32), g["ref_%s_kp_idx" % tag])
    assert np.array_equal(edges[0].astype(np.int32), g["ref_%s_edges0" % tag])
    assert np.array_equal(edges[1].astype(np.int32), g["ref_%s_edges1" % tag])


def test_oracle_against_live_reference():
    gg = reference_graph_gen()
    if gg is None:
        pytest.skip("/root/reference not present (GPU box)")
    from pointgnn_amd.synthetic import synthetic_cloud
    xyz, _ = synthetic_cloud(seed=3, preset="tiny")
    kp, _ = go.keypoints_center(xyz, xyz, 0.4)
    ref = gg.gen_disjointed_rnn_local_graph_v3(xyz, kp, 1.0, -1)
    assert np.array_equal(go.radius_graph_c(xyz, kp, 1.0),
                          go.canonical_edges(ref))
    cfg = configs.car_auto_config(3)
    np.random.seed(5)
    random.seed(5)
    rv, rk, re = gg.gen_multi_level_local_graph_v3(xyz, **cfg["graph_gen_kwargs"])
    np.random.seed(5)
    random.seed(5)
    ov, ok, oe = go.multi_level_graph(xyz, **cfg["graph_gen_kwargs"])
    for a, b in zip(rk + re, ok + oe):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_center_keypoints_properties():
    """open3d is absent (parity unpinned for this call): check the restatement's
    defining properties instead."""
    g = gold("graph_tiny.npz")
    xyz = g["xyz"]
    cent, vox = go.voxel_centroids_open3d07(xyz, 0.4)
    assert len(np.unique(vox, axis=0)) == len(vox)      # one row per voxel
    origin = xyz.astype(np.float64).min(0) - 0.2
    # every centroid lies inside its voxel
    rel = (cent - origin) / 0.4
    assert np.all(rel >= vox - 1e-9) and np.all(rel <= vox + 1 + 1e-9)
    kp_xyz, kp_idx = go.keypoints_center(xyz, xyz, 0.4)
    assert np.array_equal(kp_xyz, g["kp_xyz"])
    # the matched point is a true nearest neighbour (brute force, same float64
    # arithmetic).  Exact distance TIES are common -- a voxel holding exactly
    # two points has its centroid at their exact midpoint -- and the
    # reference's pick among tied points is kd-tree traversal order, which is
    # not a defined property: parity for keypoints is "a minimiser", and
    # "the same index wherever the minimiser is unique".
    diff = cent[:, None, :] - xyz[None].astype(np.float64)
    d = (diff[..., 0] ** 2 + diff[..., 1] ** 2) + diff[..., 2] ** 2
    dmin = d.min(1)
    chosen = d[np.arange(len(cent)), kp_idx[:, 0]]
    assert np.array_equal(chosen, dmin)
    unique = (d == dmin[:, None]).sum(1) == 1
    assert unique.sum() > 0.5 * len(cent)
    assert np.array_equal(d.argmin(1)[unique], kp_idx[unique, 0])


def test_scatter_max_oracle():
    rng = np.random.default_rng(0)
    data = rng.standard_normal((50, 7)).astype(np.float32)
    ids = rng.integers(0, 9, 50)
    out = gn.scatter_max(data, ids, 12)
    for s in range(12):
        rows = data[ids == s]
        if len(rows):
            assert np.array_equal(out[s], rows.max(0))
        else:   # TF: empty segment -> lowest()
            assert np.all(out[s] == np.finfo(np.float32).min)


@pytest.mark.parametrize("t", [0, 1])
def test_gnn_oracle_golden_logits(t):
    """Trained reference weights -> oracle -> committed logits (pins the oracle
    against drift; fp32 vs fp64 shadow bounds the rounding)."""
    g = gold("graph_tiny.npz")
    w = gold("weights_car_auto_T%d.npz" % t)
    ref = gold("logits_car_auto_T%d_tiny.npz" % t)
    cfg = configs.car_auto_config(t)
    k = g["kp_xyz"].shape[0]
    coords = [g["xyz"], g["kp_xyz"], g["kp_xyz"]]
    kps = [g["kp_idx"], np.arange(k, dtype=np.int32).reshape(-1, 1)]
    edges = [g["ref_edges0"], g["ref_edges1"]]
    lg, bx = gn.predict(w, cfg, g["intensity"], coords, kps, edges)
    assert lg.shape == (k, 4) and bx.shape == (k, 4, 7)
    np.testing.assert_allclose(lg, ref["logits32"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(bx, ref["boxes32"], atol=2e-5, rtol=0)
    assert np.abs(ref["logits32"] - ref["logits64"]).max() < 1e-4
    p = gn.softmax(lg)
    np.testing.assert_allclose(p.sum(-1), 1.0, atol=1e-6)
    # permuting the edge order must not change anything (max is order-free)
    perm = np.random.default_rng(1).permutation(edges[1].shape[0])
    edges2 = [edges[0], edges[1][perm]]
    lg2, _ = gn.predict(w, cfg, g["intensity"], coords, kps, edges2)
    assert np.array_equal(lg, lg2)


# ---------------------------------------------------------------- kd-tree ties
@pytest.mark.parametrize("preset,seed,voxel", [("tiny", 1, 0.4), ("small", 0, 0.4),
                                               ("tiny", 2, 0.2), ("car", 0, 0.4)])
def test_kdtree_oracle_is_sklearn(preset, seed, voxel):
    """oracle/kdtree_oracle.py (the rule csrc/kdtree.hip implements) against
    the real scikit-learn: idx_array / node_bounds of KDTree.get_arrays() and
    the tie winners of NearestNeighbors.kneighbors, i.e. the reference's own
    call at graph_gen.py:84-88."""
    from sklearn.neighbors import KDTree
    from oracle import kdtree_oracle as ko
    from pointgnn_amd.synthetic import synthetic_cloud
    xyz, _ = synthetic_cloud(seed=seed, preset=preset)
    data = xyz.astype(np.float64)
    tree = KDTree(data, leaf_size=30)
    _, idx_ref, node_data, node_bounds = tree.get_arrays()
    idx, ranges, bounds = ko.build(data)
    assert ko.tree_shape(len(data))[1] == node_data.shape[0]
    assert np.array_equal(idx, idx_ref)
    assert np.array_equal(bounds[:, :3], node_bounds[0])
    assert np.array_equal(bounds[:, 3:], node_bounds[1])
    cent, _ = go.voxel_centroids_open3d07(xyz, voxel)
    _, kp_idx = go.keypoints_center(xyz, xyz, voxel)      # real kneighbors
    got, ties = ko.nearest_with_ties(data, cent, voxel * 0.9)
    assert ties > 0.1 * len(cent)          # ties are common (2-point voxels)
    assert np.array_equal(got, kp_idx[:, 0])


def test_kdtree_shape_abi_matches_sklearn():
    """pgnn_kdtree_shape (host arithmetic, no GPU) == sklearn's n_nodes."""
    import ctypes
    from sklearn.neighbors import KDTree
    from pointgnn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for n in (1, 2, 30, 31, 60, 61, 62, 121, 122, 500, 961, 962, 15361, 15362,
              20000, 30721, 30722, 50000):
        lv, nodes = ctypes.c_int32(), ctypes.c_int32()
        assert lib.pgnn_kdtree_shape(n, ctypes.byref(lv), ctypes.byref(nodes)) == 0
        t = KDTree(rng.random((n, 3)), leaf_size=30)
        assert nodes.value == t.get_arrays()[2].shape[0], n


def _kernel_floor_divide(a, b, dtype):
    """csrc/graph.hip npy_floor_divide_f32 / _f64 restated with NumPy scalars
    of the same precision (divisor > 0)."""
    a = dtype(a)
    b = dtype(b)
    mod = np.fmod(a, b)
    div = dtype((a - mod) / b)
    if mod != 0 and mod < 0:
        div = dtype(div - dtype(1))
    if div == 0:
        return dtype(0)
    fl = np.floor(div)
    if dtype(div - fl) > dtype(0.5):
        fl = dtype(fl + dtype(1))
    return fl


def test_random_keypoint_voxel_rule_is_numpy_floor_divide():
    """graph_gen.py:123-131 voxelises with NumPy's `//`.  The device code
    (vox_cell in csrc/graph.hip) evaluates npy_divmod's fmod-based exact floor;
    this pins that formula to NumPy itself on cell boundaries +- a few ulps,
    where floor(a / b) in the same precision can land in the neighbouring
    cell."""
    rng = np.random.default_rng(0)
    for dtype, voxel in ((np.float32, np.float32(0.8)), (np.float64, 0.8),
                         (np.float32, np.float32(0.4)), (np.float64, 0.2)):
        vals = []
        for k in list(range(0, 130)) + [1000, 4095, 65535]:
            x = dtype(k) * dtype(voxel)
            for _ in range(4):
                vals.append(x)
                x = np.nextafter(x, dtype(np.inf), dtype=dtype)
            x = dtype(k) * dtype(voxel)
            for _ in range(4):
                x = np.nextafter(x, dtype(-np.inf), dtype=dtype)
                if x >= 0:
                    vals.append(x)
        vals += list(rng.uniform(0, 120, 4000).astype(dtype))
        vals = np.array(vals, dtype=dtype)
        ref = vals // dtype(voxel)
        assert ref.dtype == dtype
        got = np.array([_kernel_floor_divide(v, voxel, dtype) for v in vals])
        assert np.array_equal(got, ref)


# ---- float64 clouds (the training data path, train.py:88-130) ----------------
@pytest.mark.parametrize("tag,rnd", [("rand", False), ("randjit", True)])
def test_oracle_on_float64_cloud_equals_reference_golden(tag, rnd):
    """graph_f64.npz (tests/golden/make_golden_f64.py): the reference's own
    graph_gen.py on a float64 cloud with points a few ulps either side of voxel
    faces and of the search spheres.  The oracle, seeded alike, reproduces the
    keypoints and both edge lists; the tree-free C predicate gives the same
    edge sets."""
    g = gold("graph_f64.npz")
    xyz = g["xyz"]
    assert xyz.dtype == np.float64
    kw = dict(configs.car_auto_config(3)["graph_gen_kwargs"])
    kw["add_rnd3d"] = rnd
    kw["level_configs"] = [dict(c, graph_gen_kwargs=dict(
        c["graph_gen_kwargs"], num_neighbors=-1)) for c in kw["level_configs"]]
    np.random.seed(0)
    random.seed(0)
    coords, kps, edges = go.multi_level_graph(xyz, **kw)
    assert np.array_equal(kps[0], g["ref_%s_kp_idx" % tag])
    assert np.array_equal(edges[0], g["ref_%s_edges0" % tag])
    assert np.array_equal(edges[1], g["ref_%s_edges1" % tag])
    kp = xyz[g["ref_%s_kp_idx" % tag][:, 0]]
    assert np.array_equal(go.radius_graph_c(xyz, kp, 1.0),
                          go.canonical_edges(g["ref_%s_edges0" % tag]))
    assert np.array_equal(go.radius_graph_c(kp, kp, 4.0),
                          go.canonical_edges(g["ref_%s_edges1" % tag]))
    # the fixture is only worth something if float32 rounding changes it
    if not rnd:
        x32 = xyz.astype(np.float32)
        assert len(go.radius_graph_c(x32, x32[g["ref_rand_kp_idx"][:, 0]],
                                     1.0)) == int(g["f32_rand_num_edges0"]) \
            != len(g["ref_rand_edges0"])


def test_labels_oracle_on_float64_vertices_equals_reference_golden():
    from oracle import labels_oracle as LO
    fix = gold("labels_f64.npz")
    labels = LO.synthetic_labels(0, LO.synthetic_vertices(0),
                                 n_boxes=int(fix["n_labels"]))
    for m in ("yaw", "Car", "Pedestrian_and_Cyclist"):
        cls, boxes, valid, _ = LO.assign_labels(labels, fix["xyz"],
                                                (1.0, 1.0, 1.0), m)
        assert np.array_equal(cls, fix[m + "_cls"])
        assert np.array_equal(boxes, fix[m + "_boxes"])
        assert np.array_equal(valid, fix[m + "_valid"])
        cls32 = LO.assign_labels(labels, fix["xyz"].astype(np.float32),
                                 (1.0, 1.0, 1.0), m)[0]
        assert int((cls32 != cls).sum()) == int(fix[m + "_n_diff_f32"]) > 0


def test_kdtree_oracle_heap_select_fallback_equals_sklearn():
    """libstdc++'s introselect gives up after 2*floor(log2 n) partition rounds
    and finishes with std::__heap_select; about one cloud in fifty has such a
    node.  The restatement (oracle/kdtree_oracle.py heap_select) against the
    REAL scikit-learn KDTree on clouds where it runs."""
    from sklearn.neighbors import KDTree
    from oracle import kdtree_oracle as ko
    from pointgnn_amd.synthetic import synthetic_cloud
    clouds = [synthetic_cloud(seed=19, preset="small")[0]]
    rng = np.random.default_rng(0)
    for t in range(91):
        n = int(rng.integers(200, 6000))
        xyz = (rng.standard_normal((n, 3)) * np.array([20, 2, 30])).astype(
            np.float32)
        if t in (7, 39, 90):
            clouds.append(xyz)
    for xyz in clouds:
        ko.HEAP_SELECT_CALLS[0] = 0
        idx, _, bounds = ko.build(xyz)
        assert ko.HEAP_SELECT_CALLS[0] >= 1
        _, idx_ref, _, nb = KDTree(xyz.astype(np.float64),
                                   leaf_size=30).get_arrays()
        assert np.array_equal(idx, idx_ref)
        assert np.array_equal(bounds[:, :3], nb[0])


# ------------------------------------------------- BASELINE sizes (round 4)
def _fullsize_digest(src, dst):
    import hashlib
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    order = np.lexsort((src, dst))
    rows = np.stack([src[order], dst[order]], axis=1).astype("<i4")
    return hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest()


@pytest.mark.parametrize("preset", ["car", "car_600k", "ped_dense"])
def test_c_restatement_equals_reference_full_size_digest(preset):
    """oracle/radius_bruteforce.c (the full-size checker of the GPU tests)
    against the digests of the lists the reference's own
    gen_disjointed_rnn_local_graph_v3 wrote at BASELINE size
    (tests/golden/make_golden_fullsize.py)."""
    g = dict(np.load(os.path.join(GOLD, "fullsize_%s.npz" % preset)))
    xyz, _ = synthetic_cloud(seed=0, preset=preset)
    kp_xyz = xyz[g["kp_idx"][:, 0]]
    r0, r1 = (float(r) for r in g["radii"])
    e0 = go.radius_graph_c(xyz, kp_xyz, r0)
    e1 = go.radius_graph_c(kp_xyz, kp_xyz, r1)
    assert (len(e0), len(e1)) == (int(g["E0"]), int(g["E1"]))
    assert _fullsize_digest(e0[:, 0], e0[:, 1]) == str(g["edges0_sha"])
    assert _fullsize_digest(e1[:, 0], e1[:, 1]) == str(g["edges1_sha"])


def test_gnn_oracle_equals_reference_tf_graph_full_size():
    """BASELINE config 2 at its own size: oracle/gnn_oracle.predict with the
    TRAINED car_auto_T1 weights on the whole 20 000-point `car` frame against
    the reference's serialized TF graph (fixture T1_logits)."""
    g = dict(np.load(os.path.join(GOLD, "fullsize_car.npz")))
    w = dict(np.load(os.path.join(GOLD, "weights_car_auto_T1.npz")))
    xyz, inten = synthetic_cloud(seed=0, preset="car")
    kp_idx = g["kp_idx"].astype(np.int32)
    kp_xyz = xyz[kp_idx[:, 0]]
    k = len(kp_idx)
    r0, r1 = (float(r) for r in g["radii"])
    edges = [go.radius_graph_c(xyz, kp_xyz, r0),
             go.radius_graph_c(kp_xyz, kp_xyz, r1)]
    lg, bx = gn.predict(w, configs.car_auto_config(1), inten,
                        [xyz, kp_xyz, kp_xyz],
                        [kp_idx, np.arange(k, dtype=np.int32).reshape(-1, 1)],
                        edges, dtype=np.float32)
    print("max|dlogit| %.3g max|dbox| %.3g" % (
        np.abs(lg - g["T1_logits"]).max(),
        np.abs(bx - g["T1_box_encodings"]).max()))
    np.testing.assert_allclose(lg, g["T1_logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(bx, g["T1_box_encodings"], atol=1e-4, rtol=0)


# ---- more than one pooling level (graph_gen.py:49-90, :92-153) ---------------
@pytest.mark.parametrize("preset", ["tiny", "small"])
def test_multi_level_pooling_center_golden(preset):
    """Two pooling levels + a same-scale level, 'center': the oracle against
    what the reference's real gen_multi_level_local_graph_v3 returned
    (tests/golden/make_golden_multilevel.py; only open3d's voxel means are a
    restatement): vertex coordinates, keypoint indices -- the second level's
    index the FIRST level's vertices -- and the edge lists, row for row."""
    from _multilevel import BASE_VOXEL, LEVEL_CONFIGS
    g = gold("graph_multilevel.npz")
    xyz = g["%s_xyz" % preset]
    coords, kps, edges = go.multi_level_graph(
        xyz, BASE_VOXEL, LEVEL_CONFIGS, downsample_method='center')
    assert len(coords) == 4
    for l in range(3):
        assert np.array_equal(kps[l], g["%s_center_kp%d" % (preset, l)])
        assert np.array_equal(coords[l + 1],
                              g["%s_center_coords%d" % (preset, l + 1)])
        assert np.array_equal(edges[l], g["%s_center_edges%d" % (preset, l)])
    # the case is only worth something if the second level really pools
    assert len(coords[2]) < len(coords[1]) < len(xyz)
    assert np.array_equal(coords[2], coords[1][kps[1][:, 0]])


@pytest.mark.parametrize("tag,rnd", [("rand", False), ("randjit", True)])
def test_multi_level_pooling_random_golden(tag, rnd):
    """The same configuration, 'random' (RNGs seeded like the fixture's run):
    the second level voxelises the FIRST level's vertices on the grid anchored
    at the ORIGINAL cloud's minimum (graph_gen.py:108-110)."""
    from _multilevel import BASE_VOXEL, LEVEL_CONFIGS
    g = gold("graph_multilevel.npz")
    xyz = g["tiny_xyz"]
    np.random.seed(0)
    random.seed(0)
    coords, kps, edges = go.multi_level_graph(
        xyz, BASE_VOXEL, LEVEL_CONFIGS, add_rnd3d=rnd,
        downsample_method='random')
    for l in range(3):
        assert np.array_equal(kps[l], g["tiny_%s_kp%d" % (tag, l)])
    assert [len(c) for c in coords] + [len(e) for e in edges] == \
        list(g["tiny_%s_counts" % tag])


def test_multi_level_pooling_golden_is_the_reference_live():
    """With /root/reference present: the fixture equals a fresh run of the
    reference's function (and so does the oracle, by the tests above)."""
    gg = reference_graph_gen()
    if gg is None:
        pytest.skip("reference tree not present (GPU box)")
    import sys
    sys.path.insert(0, GOLD)
    try:
        import make_golden_multilevel as mk
    finally:
        sys.path.remove(GOLD)
    mk.install_open3d_stand_in()
    g = gold("graph_multilevel.npz")
    xyz = g["tiny_xyz"]
    vc, ki, el = gg.gen_multi_level_local_graph_v3(
        xyz, mk.BASE_VOXEL, mk.LEVEL_CONFIGS, add_rnd3d=False,
        downsample_method='center')
    for l in range(3):
        assert np.array_equal(np.asarray(ki[l]), g["tiny_center_kp%d" % l])
        assert np.array_equal(np.asarray(el[l]), g["tiny_center_edges%d" % l])


def test_torch_cpu_port_equals_the_numpy_oracle():
    """oracle/gnn_oracle_torch.py (what bench.py's cpu_baseline times: the same
    arithmetic in cache-sized row chunks on torch-CPU) against
    gnn_oracle.predict on the reference-built tiny graph, car and ped shapes."""
    from oracle import gnn_oracle as gn, gnn_oracle_torch as gt
    from pointgnn_amd import configs, weights
    g = dict(np.load(os.path.join(GOLD, "graph_tiny.npz")))
    k = g["kp_xyz"].shape[0]
    coords = [g["xyz"], g["kp_xyz"], g["kp_xyz"]]
    kps = [g["kp_idx"], np.arange(k, dtype=np.int32).reshape(-1, 1)]
    edges = [g["ref_edges0"], g["ref_edges1"]]
    old = gt.CHUNK_ROWS, gt.WORKERS
    try:
        for name, chunk, workers in (("car_auto_T3", 1 << 15, 0),
                                     ("car_auto_T2", 257, 0),
                                     ("car_auto_T3", 129, 4),
                                     ("ped_cyl_auto_T3", 1000, 3)):
            gt.CHUNK_ROWS = chunk      # runs cut by chunk boundaries too
            gt.WORKERS = workers       # ... and chunks on worker threads
            cfg = configs.get_config(name)
            params = weights.init_params(cfg, seed=5, bias_scale=0.05)
            l0, b0 = gn.predict(params, cfg, g["intensity"], coords, kps, edges)
            l1, b1 = gt.predict(params, cfg, g["intensity"], coords, kps, edges)
            np.testing.assert_allclose(l1, l0, atol=2e-5, rtol=0)
            np.testing.assert_allclose(b1, b0, atol=2e-5, rtol=0)
    finally:
        gt.CHUNK_ROWS, gt.WORKERS = old
