"""Host-side logic that needs no GPU: configs, variable specs, checkpoint
reader, synthetic input, C-ABI library load/symbols, weight packing layout."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs, tf_bundle, weights
from pointgnn_amd.synthetic import synthetic_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def test_param_counts_match_reference_checkpoints():
    # SURVEY.md §8c totals (include the int32 global-step scalar)
    expect = {"car_auto_T0": 344933, "car_auto_T1": 726492,
              "car_auto_T2": 1108051, "car_auto_T3": 1489610,
              "ped_cyl_auto_T3": 1357274, "car_fixed_T3": 1431233}
    for name, total in expect.items():
        assert weights.count_params(configs.get_config(name)) + 1 == total


def test_variable_names_match_golden_weights():
    for t in (0, 1):
        w = np.load(os.path.join(GOLD, "weights_car_auto_T%d.npz" % t))
        spec = dict(weights.variable_specs(configs.car_auto_config(t)))
        assert set(spec) == set(w.files)
        for k, shape in spec.items():
            assert w[k].shape == tuple(shape)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_generated_configs_equal_reference_json():
    pairs = [("car_auto_T0", "car_auto_T0_train_config"),
             ("car_auto_T1", "car_auto_T1_train_config"),
             ("car_auto_T2", "car_auto_T2_train_config"),
             ("car_auto_T3", "car_auto_T3_train_config"),
             ("car_auto_T3", "car_auto_T3_trainval_config"),
             ("car_fixed_T3", "car_fixed_T3_train_config"),
             ("ped_cyl_auto_T3", "ped_cyl_auto_T3_trainval_config")]
    for name, fname in pairs:
        ref = configs.load_config(os.path.join(REF, "configs", fname))
        assert ref == configs.get_config(name), fname


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_generated_train_and_eval_configs_equal_reference_json(tmp_path):
    for name in configs._TRAIN_TABLE:
        ref = configs.load_train_config(
            os.path.join(REF, "configs", name + "_train_config"))
        assert ref == configs.get_train_config(name), name
        ev = configs.load_config(os.path.join(REF, "configs",
                                              name + "_eval_config"))
        assert ev == configs.get_eval_config(name), name
    # save/load round trip in the reference's JSON form
    path = str(tmp_path / "train_config")
    configs.save_train_config(path, configs.get_train_config("car_auto_T3_train"))
    assert configs.load_train_config(path) == \
        configs.get_train_config("car_auto_T3_train")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_tf_bundle_reader_on_reference_checkpoints():
    for t in (0, 1):
        ck = tf_bundle.load_checkpoint(
            os.path.join(REF, "checkpoints", "car_auto_T%d_train" % t))
        assert int(ck["Variable"]) == 1400000
        gold = np.load(os.path.join(GOLD, "weights_car_auto_T%d.npz" % t))
        for k in gold.files:
            assert np.array_equal(ck[k], gold[k])
    # index-only checkpoints still list names/shapes
    names = tf_bundle.list_variables(os.path.join(
        REF, "checkpoints", "car_auto_T3_train", "model-1400000"))
    spec = dict(weights.variable_specs(configs.car_auto_config(3)))
    got = {n: s for n, _, s, _, _ in names if n != "Variable"}
    assert got == {k: tuple(v) for k, v in spec.items()}


def test_crc32c_known_answers():
    # RFC 3720 B.4 check value and TF's masking identity
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283
    assert tf_bundle.crc32c(b"") == 0
    assert tf_bundle.crc32c(b"6789", tf_bundle.crc32c(b"12345")) == 0xE3069283


def test_checkpoint_writer_round_trip(tmp_path):
    gold = np.load(os.path.join(GOLD, "weights_car_auto_T0.npz"))
    params = {k: gold[k] for k in gold.files}
    prefix = tf_bundle.save_checkpoint(str(tmp_path), params, global_step=77)
    assert os.path.basename(prefix) == "model-77"
    back = tf_bundle.load_checkpoint(str(tmp_path))   # via the `checkpoint` file
    assert int(back.pop("Variable")) == 77
    assert set(back) == set(params)
    for k in params:
        assert back[k].dtype == params[k].dtype
        assert np.array_equal(back[k], params[k])
    names = [n for n, _, _, _, _ in tf_bundle.list_variables(prefix)]
    assert names == sorted(names)                      # SSTable key order


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree absent")
def test_checkpoint_writer_is_byte_identical_to_tensorflow(tmp_path):
    """Re-encoding the variables of the reference's TF-written checkpoints
    reproduces both files bit for bit (table layout, prefix compression,
    masked crc32c of every tensor and block)."""
    for t in (0, 1):
        src = os.path.join(REF, "checkpoints", "car_auto_T%d_train" % t)
        v = tf_bundle.load_checkpoint(src)
        step = int(v.pop("Variable"))
        d = tmp_path / ("t%d" % t)
        prefix = tf_bundle.save_checkpoint(str(d), v, global_step=step)
        for ext in (".index", ".data-00000-of-00001"):
            with open(prefix + ext, "rb") as a, \
                    open(os.path.join(src, "model-%d%s" % (step, ext)), "rb") as b:
                assert a.read() == b.read(), ext
        with open(os.path.join(str(d), "checkpoint")) as a, \
                open(os.path.join(src, "checkpoint")) as b:
            assert a.read() == b.read()


def test_synthetic_cloud_is_deterministic_and_shaped():
    a, ia = synthetic_cloud(seed=4, preset="tiny")
    b, ib = synthetic_cloud(seed=4, preset="tiny")
    assert np.array_equal(a, b) and np.array_equal(ia, ib)
    assert a.dtype == np.float32 and a.shape == (1500, 3)
    assert ia.shape == (1500, 1) and ia.min() >= 0 and ia.max() < 1
    c, _ = synthetic_cloud(seed=5, preset="tiny")
    assert not np.array_equal(a, c)
    assert a[:, 2].min() > 0          # camera frame: z forward
    rng = np.sqrt((a ** 2).sum(1))
    assert rng.min() > 1.9 and rng.max() < 71


def test_init_params_shapes_and_seed():
    cfg = configs.car_auto_config(1)
    p = weights.init_params(cfg, seed=1)
    q = weights.init_params(cfg, seed=1)
    for (name, shape) in weights.variable_specs(cfg):
        assert p[name].shape == tuple(shape) and p[name].dtype == np.float32
        assert np.array_equal(p[name], q[name])
    w = p["layer2/extract_vertex_features/fully_connected/weights"]
    assert w.shape == (303, 300)
    assert np.abs(w).max() <= np.sqrt(6.0 / 603) + 1e-6


# ---- C-ABI library ---------------------------------------------------------
def _lib():
    from pointgnn_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from pointgnn_amd import build
        build.build(verbose=False)
    return L, L.load()


def test_library_exports_every_declared_symbol():
    L, lib = _lib()
    header = open(os.path.join(ROOT, "include", "pointgnn_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pgnn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    # the Python binding covers the header (plus the tuning hook)
    assert declared <= set(L.exported_symbols())
    assert lib.pgnn_version() >= 100


def test_error_reporting_without_gpu():
    L, lib = _lib()
    # argument validation happens before any HIP call
    rc = lib.pgnn_scatter_max_f32(None, 4, None, 5, 8, 3, None, 4, 0, None)
    assert rc == -1
    assert b"stride" in lib.pgnn_last_error()
    assert lib.pgnn_pack_fc(None, None, 4, 4, None) == -1
    assert lib.pgnn_packed_fc_floats(303, 300) == 19 * 19 * 256 + 19 * 16
    with pytest.raises(L.PointGnnHipError):
        L.check(-1, "demo")
    # the widened entries validate before touching HIP as well
    assert lib.pgnn_nms_boxes_3d(None, None, None, None, -1, 0.5, 1, 10.0, -1,
                                 None, 0, None, None, None, None, None,
                                 None) == -1
    assert lib.pgnn_nms_boxes_3d(None, None, None, None, 5, 0.5, 9, 10.0, -1,
                                 None, 0, None, None, None, None, None,
                                 None) == -1          # mode out of range
    assert b"nms_boxes_3d" in lib.pgnn_last_error()
    assert lib.pgnn_nms_workspace_bytes(-1) == 0
    assert 0 < lib.pgnn_nms_workspace_bytes(100) < \
        lib.pgnn_nms_workspace_bytes(10000)
    assert lib.pgnn_box_decode_f32(None, None, None, None, 0, 5, 0, None,
                                   None) == -1        # boxes_per_row = 0
    assert lib.pgnn_kitti_cam_points_in_image(
        None, 10, None, None, 1242.0, 375.0, None, 0, 0, None, 0, None, None,
        1, 10, None, None) == -1                       # no count pointer
    assert lib.pgnn_kitti_ingest_workspace_bytes(120000) > 0
    # collectives: arguments are checked before RCCL is entered
    import ctypes
    h = ctypes.c_void_p()
    assert lib.pgnn_comm_unique_id(None) == -1
    assert lib.pgnn_comm_init_rank(None, 1, 0, ctypes.byref(h)) == -1
    assert lib.pgnn_comm_init_rank(b"\0" * 128, 2, 2, ctypes.byref(h)) == -1
    assert lib.pgnn_allreduce_sum_f32(None, None, 4, None) == -1
    assert lib.pgnn_allreduce_step(None, None, 0, None, 0, None) == -1
    assert lib.pgnn_comm_destroy(None) == 0
    assert b"rccl" in lib.pgnn_comm_library().lower()
    assert lib.pgnn_assign_box_labels(None, -1, None, 0, None, None, None,
                                      None, None) == -1
    assert lib.pgnn_vertex_pre_edge_fwd(None, 0, 0, None, None, 0, None, None,
                                        5, None, None, 0, None, 0, None) == -1
    assert lib.pgnn_points_in_box_f64(None, 5, None, None, None, None,
                                      None) == -1
    assert lib.pgnn_metrics_state_bytes(4, 200) == 4 * (3 + 2 * 201) * 8
    assert lib.pgnn_metrics_state_bytes(0, 200) == 0
    assert lib.pgnn_metrics_update(None, 4, None, 10, 4, 5000, None,
                                   None) == -1        # too many thresholds
    assert lib.pgnn_metrics_update(None, 2, None, 10, 4, 200, None,
                                   None) == -1        # stride < classes
    assert lib.pgnn_metrics_compute(None, 4, 200, None, None) == -1


def test_capacity_form_host_logic():
    """The host side of the capacity form (graph_gen deferred_counts): size
    hints and capacities, the FrameCounts record, the pgnn_dyn_count struct
    and the *_dyn entries' argument checks -- no GPU involved."""
    import ctypes
    from pointgnn_amd import graph_gen as G
    L, lib = _lib()
    # struct pgnn_dyn_count {const int32_t *dev; int64_t hint;}
    assert ctypes.sizeof(L.DynCount) == 16
    assert L.DynCount.dev.offset == 0 and L.DynCount.hint.offset == 8
    h = G.CountHints()
    assert h.cap(0) == G.CountHints.MIN_EDGE_CAP and h.edge_hint(1) == 0
    h.update(3000, [350000, 600000])
    assert h.k == 3000 and h.edges == [350000, 600000]
    assert h.cap(0) >= 2 * 350000 and h.cap(1) >= 2 * 600000
    caps = list(h.edge_caps)
    h.update(2500, [10, 20])                  # hints follow, capacities stay
    assert h.k == 2500 and h.edges == [10, 20] and h.edge_caps == caps
    h.update(2500, [900000, 20])              # ... and only ever grow
    assert h.cap(0) >= 1800000 and h.cap(1) == caps[1]
    c = G.FrameCounts(None, [100, 50])
    c._host = [7, 0, 100, 130, 40, 40]        # level 0 needed 130 rows, had 100
    assert (c.k, c.kd_status, c.edges, c.overflowed) == (7, 0, [130, 40], [0])
    c._host = [7, 0, 90, 90, 40, 40]
    assert c.overflowed == [] and c.raw_edges is None
    # levels with a fan-in cap (training kwargs): the uncapped list has its
    # own capacity and its own record behind the levels' (written, required)
    assert h.raw_cap(1) == 2 * h.cap(1)       # nothing seen yet
    h.update(2500, [900000, 20], raw_edges=[900000, 5000])
    assert h.raw_cap(0) >= 1800000 and h.raw_cap(1) == G.CountHints.MIN_EDGE_CAP
    c._host = [7, 0, 90, 90, 0, 40, 0, 0, 300, 500]   # level 1: uncapped list cut
    assert c.raw_edges == [0, 500] and c.edges == [90, 40]
    assert c.overflowed == [1]                # flagged as {0, required}
    assert lib.pgnn_radius_graph_dyn_cap(None, 0, 10, 10, None, 0, None, 8, 1,
                                         None, None, 0, None, None) == -1
    assert lib.pgnn_radius_graph_dyn_cap(None, 0, 10, 10, None, 0, None, 0, 1,
                                         None, None, 0, None, None) == -1
    # the f16x2 pooling entry: null layers / image are argument errors
    assert lib.pgnn_point_set_pooling_f16x2_fwd(
        None, 1, None, None, None, 0, 0, None, 4, None, None, 1, None, 304,
        None, None, None, None, None) == -1
    # the capacity-form entries validate before any HIP call
    assert lib.pgnn_radius_graph_dyn_workspace_bytes(-1, 5) == 0
    assert lib.pgnn_radius_graph_dyn_workspace_bytes(20000, 20000) > \
        lib.pgnn_radius_graph_workspace_bytes(20000, 20000)
    assert lib.pgnn_radius_graph_dyn(None, 10, None, None, 10, None, 1.0, None,
                                     None, 0, None, 100, None, None) == -1
    # ... and so do its two stages (grid: points only; query: centres too)
    assert lib.pgnn_radius_graph_dyn_grid(None, 10, None, 10, 1.0, None, None,
                                          0, None) == -1
    assert lib.pgnn_radius_graph_dyn_grid(None, 0, None, 0, -1.0, None, None,
                                          0, None) == -1
    assert lib.pgnn_radius_graph_dyn_query(None, 0, None, 10, None, 1.0, None,
                                           None, 0, None, 100, None,
                                           None) == -1
    assert lib.pgnn_mlp_fwd_dyn(None, 0, 4, None, 0, 0, 16, None, 1, None, 0,
                                None, 0, None, None) == -1   # null count
    assert b"count" in lib.pgnn_last_error()
    assert lib.pgnn_vertex_pre_edge_fwd_dyn(None, 0, 0, None, None, 0, None,
                                            None, 5, None, None, 0, None, 0,
                                            None, None) == -1
    # builder tunables of the frame pipeline exist and are range-checked
    for key, ok, bad in (("graph_max_wgs", 8, -1), ("graph_lds_pad", 32768,
                                                    1 << 20),
                         ("ws_reserve", 8, 3)):
        assert lib.pgnn_set_tunable(key.encode(), ok) == 0
        assert lib.pgnn_set_tunable(key.encode(), bad) != 0
        assert lib.pgnn_set_tunable(key.encode(), 0) == 0


def _emulate_mfma_layer(x, packed, k_in, n_out):
    """NumPy model of mlp_engine.h: A fragment = x[row][16q + 4g + s], B
    fragment = packed[q][t][lane][s]; v_mfma_f32_16x16x4 sums the four k-slots
    g = lane>>4; D[row][col] lives in lane = col + 16*(row//4), reg row%4."""
    kq, nt = (k_in + 15) // 16, (n_out + 15) // 16
    rows = x.shape[0]
    assert rows % 16 == 0
    xp = np.zeros((rows, 16 * kq), np.float32)
    xp[:, :k_in] = x
    w = packed[:kq * nt * 256].reshape(kq, nt, 64, 4)
    bias = packed[kq * nt * 256:]
    out = np.zeros((rows, 16 * nt), np.float32)
    lanes = np.arange(64)
    for m in range(rows // 16):
        for t in range(nt):
            acc = np.zeros((16, 16), np.float32)
            for q in range(kq):
                for s in range(4):
                    a = np.zeros((16, 4), np.float32)   # A[i][kslot]
                    b = np.zeros((4, 16), np.float32)   # B[kslot][j]
                    a[lanes & 15, lanes >> 4] = xp[16 * m + (lanes & 15),
                                                   16 * q + 4 * (lanes >> 4) + s]
                    b[lanes >> 4, lanes & 15] = w[q, t, lanes, s]
                    acc += a @ b
            out[16 * m:16 * m + 16, 16 * t:16 * t + 16] = acc
    return out + bias[None, :]


@pytest.mark.parametrize("k_in,n_out", [(4, 32), (303, 300), (64, 3), (20, 17)])
def test_pack_fc_layout_reproduces_matmul(k_in, n_out):
    L, lib = _lib()
    rng = np.random.default_rng(k_in * 1000 + n_out)
    w = rng.standard_normal((k_in, n_out)).astype(np.float32)
    b = rng.standard_normal(n_out).astype(np.float32)
    n = lib.pgnn_packed_fc_floats(k_in, n_out)
    packed = np.full(n, np.nan, np.float32)
    assert lib.pgnn_pack_fc(w.ctypes.data, b.ctypes.data, k_in, n_out,
                            packed.ctypes.data) == 0
    assert np.isfinite(packed).all()
    x = rng.standard_normal((16, k_in)).astype(np.float32)
    got = _emulate_mfma_layer(x, packed, k_in, n_out)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    np.testing.assert_allclose(got[:, :n_out], ref, atol=1e-4, rtol=1e-4)
    assert np.all(got[:, n_out:] == 0)       # zero padding stays zero


def test_product_path_has_no_oracle_import():
    """The shipped package must never reach into oracle/ (or any CPU
    fallback)."""
    pkg = os.path.join(ROOT, "point-gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_input_features_switch_matches_run_py():
    """run.py:226-241 / train.py:91-104: the `input_features` config key."""
    import torch
    from pointgnn_amd.kitti_dataset import Points
    from pointgnn_amd.run import _input_features
    attr = torch.arange(20, dtype=torch.float32).reshape(5, 4) + 1
    pts = Points(xyz=torch.zeros(5, 3), attr=attr)
    a = attr.numpy()
    want = {
        'irgb': a,
        '0rgb': np.hstack([np.zeros((5, 1)), a[:, 1:]]),
        '0000': np.zeros_like(a),
        'i000': np.hstack([a[:, [0]], np.zeros((5, 3))]),
        'i': a[:, [0]],
        '0': np.zeros((5, 1)),
    }
    for kind, ref in want.items():
        got = _input_features({'input_features': kind}, pts).numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), kind
    assert np.array_equal(attr.numpy(), a)        # the cloud is not modified
    with pytest.raises(ValueError):
        _input_features({'input_features': 'rgb'}, pts)


def test_learning_rate_schedule_matches_tf_exponential_decay():
    """train.py:375-378: tf.train.exponential_decay(staircase=True)."""
    from pointgnn_amd.train import learning_rate
    tc = configs.get_train_config("car_auto_T3_train")
    assert learning_rate(tc, 0) == 0.125
    assert learning_rate(tc, 399999) == 0.125
    assert learning_rate(tc, 400000) == pytest.approx(0.0125)
    assert learning_rate(tc, 1399999) == pytest.approx(0.125 * 0.1 ** 3)
    ped = configs.get_train_config("ped_cyl_auto_T3_trainval")
    assert learning_rate(ped, 800000) == pytest.approx(0.32 * 0.25 ** 2)


def test_checkpoint_state_file_lists_and_prunes(tmp_path):
    """tf.train.Saver bookkeeping (train.py:496,516,634-636): the `checkpoint`
    state file lists the checkpoints kept so far, oldest first; beyond
    max_to_keep (TF default 5) the oldest are deleted; the prefix comes from
    train_config['checkpoint_path']."""
    import numpy as np
    from pointgnn_amd import tf_bundle
    d = str(tmp_path)
    var = {"layer1/w": np.arange(6, dtype=np.float32).reshape(2, 3)}
    for step in range(1, 8):
        tf_bundle.save_checkpoint(d, var, global_step=step, name="mymodel",
                                  max_to_keep=3)
    lines = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert lines == ['model_checkpoint_path: "mymodel-7"',
                     'all_model_checkpoint_paths: "mymodel-5"',
                     'all_model_checkpoint_paths: "mymodel-6"',
                     'all_model_checkpoint_paths: "mymodel-7"']
    files = sorted(f for f in os.listdir(d) if f.endswith(".index"))
    assert files == ["mymodel-5.index", "mymodel-6.index", "mymodel-7.index"]
    ck = tf_bundle.load_checkpoint(d)
    assert int(ck["Variable"]) == 7 and np.array_equal(ck["layer1/w"], var["layer1/w"])
    # saving the same step again does not duplicate its entry
    tf_bundle.save_checkpoint(d, var, global_step=7, name="mymodel", max_to_keep=3)
    assert open(os.path.join(d, "checkpoint")).read().count("mymodel-7") == 2


def test_sched_ws_size_matches_header():
    """The Python loader allocates the scheduling counters the header
    promises (PGNN_SCHED_WS_INTS), and the kernels' own layout fits in them."""
    from pointgnn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pointgnn_hip.h")).read()
    m = re.search(r"#define\s+PGNN_SCHED_WS_INTS\s+(\d+)", hdr)
    assert m and int(m.group(1)) == _lib.SCHED_WS_INTS
    src = open(os.path.join(ROOT, "point-gnn_amd", "csrc", "edge_ws.h")).read()
    slices = int(re.search(r"kWsMaxSlices\s*=\s*(\d+)", src).group(1))
    groups = int(re.search(r"kWsMaxGroups\s*=\s*(\d+)", src).group(1))
    assert 2 + slices * groups <= _lib.SCHED_WS_INTS


def test_variable_scope_is_thread_local():
    """gnn.parameters / gnn.variable_scope mirror tf.variable_scope: the stack
    belongs to the calling thread, so two threads driving operators do not see
    (or corrupt) each other's scope."""
    import threading
    from pointgnn_amd import gnn
    seen = {}
    gate_a, gate_b = threading.Event(), threading.Event()

    def worker():
        seen["store_in_thread"] = gnn._state.store
        with gnn.variable_scope("other"):
            gate_a.set()
            gate_b.wait(5)
            seen["scope_in_thread"] = gnn._scope("x")

    store = gnn.ParamStore({})
    with gnn.parameters(store), gnn.variable_scope("layer1"):
        t = threading.Thread(target=worker)
        t.start()
        assert gate_a.wait(5)
        assert gnn._scope("w") == "layer1/w"       # untouched by the thread
        gate_b.set()
        t.join()
    assert seen["store_in_thread"] is None
    assert seen["scope_in_thread"] == "other/x"
    assert gnn._state.scope == [] and gnn._state.store is None


def test_bf16x3_weight_image_is_an_exact_split():
    """pgnn_pack_fc_bf16x3 (host side of csrc/edge_ws_bf16.h): the three bf16
    parts of every weight sum to the fp32 weight EXACTLY, sit where the
    kernel's A-operand layout expects them ([kb][t][part][lane][8]), pad with
    zeros, and the bias follows in fp32."""
    import numpy as np
    from pointgnn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    k_in, n_out = 300, 300
    w = (rng.standard_normal((k_in, n_out)) *
         np.exp(rng.uniform(-20, 5, (k_in, n_out)))).astype(np.float32)
    b = rng.standard_normal(n_out).astype(np.float32)
    nbytes = lib.pgnn_packed_fc_bf16x3_bytes(k_in, n_out)
    kb, nt = 10, 19
    assert nbytes == kb * nt * 3 * 1024 + nt * 16 * 4
    host = np.empty(nbytes, np.uint8)
    _lib.check(lib.pgnn_pack_fc_bf16x3(w.ctypes.data, b.ctypes.data, k_in,
                                       n_out, host.ctypes.data))
    img = host[:kb * nt * 3 * 1024].view(np.uint16).reshape(kb, nt, 3, 64, 8)
    parts = (img.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    total = parts.sum(axis=2)                       # [kb, nt, lane, j]
    lane = np.arange(64)
    kk = (32 * np.arange(kb)[:, None, None, None] +
          8 * (lane >> 4)[None, None, :, None] + np.arange(8)[None, None, None])
    nn = (16 * np.arange(nt)[None, :, None, None] +
          (lane & 15)[None, None, :, None]) + 0 * kk
    kk = kk + 0 * nn
    inside = (kk < k_in) & (nn < n_out)
    want = np.where(inside, w.astype(np.float64)[np.minimum(kk, k_in - 1),
                                                 np.minimum(nn, n_out - 1)], 0.0)
    assert np.array_equal(total, want)              # exact, not approximate
    # each part is what remains after the ones before it, rounded to 8 bits
    assert np.all(np.abs(parts[:, :, 1]) <= np.abs(parts[:, :, 0]) * 2.0 ** -8 + 1e-300)
    bias = host[kb * nt * 3 * 1024:].view(np.float32)
    assert np.array_equal(bias[:n_out], b) and np.all(bias[n_out:] == 0)


def test_f16x2_weight_images_natural_and_accumulator_order():
    """pgnn_pack_fc_f16x2 / _acc (host side of csrc/edge_ws_f16.h and
    pool_ws_f16.h): w0 = fp16(w), w1' = fp16((w - w0) 2^11) at
    [kb][t][part][lane][8]; the two entries differ only in WHICH row of W slot
    (g, j) of a block holds -- 8 g + j, or the order in which a lane of the
    previous layer's fp32 accumulators holds a row's features (4 g + j for
    j < 4, 16 + 4 g + j - 4 otherwise); w0 + w1' / 2^11 is within 2^-22 of w."""
    import numpy as np
    from pointgnn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for k_in, n_out in ((64, 128), (128, 300), (300, 300)):
        w = rng.standard_normal((k_in, n_out)).astype(np.float32)
        b = rng.standard_normal(n_out).astype(np.float32)
        kb, nt = (k_in + 31) // 32, (n_out + 15) // 16
        nbytes = lib.pgnn_packed_fc_f16x2_bytes(k_in, n_out)
        assert nbytes == kb * nt * 2 * 1024 + nt * 16 * 4
        lane, j = np.arange(64)[:, None], np.arange(8)[None, :]
        g = lane >> 4
        slot = {"pgnn_pack_fc_f16x2": 8 * g + j,
                "pgnn_pack_fc_f16x2_acc":
                    np.where(j < 4, 4 * g + j, 16 + 4 * g + (j - 4))}
        for name, within in slot.items():
            assert sorted(set(within.reshape(-1).tolist())) == list(range(32))
            host = np.empty(nbytes, np.uint8)
            _lib.check(getattr(lib, name)(w.ctypes.data, b.ctypes.data, k_in,
                                          n_out, host.ctypes.data))
            img = host[:kb * nt * 2 * 1024].view(np.float16).reshape(
                kb, nt, 2, 64, 8).astype(np.float64)
            kk = 32 * np.arange(kb)[:, None, None, None] + within[None, None]
            nn = 16 * np.arange(nt)[None, :, None, None] + \
                (lane & 15)[None, None] + 0 * kk
            kk = kk + 0 * nn
            inside = (kk < k_in) & (nn < n_out)
            want = np.where(inside, w[np.minimum(kk, k_in - 1),
                                      np.minimum(nn, n_out - 1)], 0.0)
            w0 = want.astype(np.float16).astype(np.float64)
            assert np.array_equal(img[:, :, 0], w0)
            assert np.array_equal(
                img[:, :, 1],
                ((want - w0) * 2048.0).astype(np.float16).astype(np.float64))
            err = np.abs(img[:, :, 0] + img[:, :, 1] / 2048.0 - want)
            assert np.all(err <= np.abs(want) * 2.0 ** -22 + 2.0 ** -36)
            bias = host[kb * nt * 2 * 1024:].view(np.float32)
            assert np.array_equal(bias[:n_out], b) and np.all(bias[n_out:] == 0)
    w = np.full((32, 16), 70000.0, np.float32)
    host = np.empty(lib.pgnn_packed_fc_f16x2_bytes(32, 16), np.uint8)
    assert lib.pgnn_pack_fc_f16x2_acc(w.ctypes.data, None, 32, 16,
                                      host.ctypes.data) == _lib.E_UNSUPPORTED


def test_batch_norm_layers_are_folded_and_training_refuses_them():
    """ParamStore.fc folds slim.batch_norm's inference map into a layer that
    has BatchNorm statistics instead of biases (gnn.py:17-23, 60-103); the
    folded layer equals the unfolded expression in float64.  A Trainer given
    such a config refuses it instead of training without the normalizer."""
    import copy
    from pointgnn_amd import gnn, train
    from oracle import gnn_oracle as gn
    rng = np.random.default_rng(0)
    w = rng.standard_normal((7, 5)).astype(np.float32)
    mean = rng.standard_normal(5).astype(np.float32)
    var = rng.uniform(0.3, 3.0, 5).astype(np.float32)
    beta = rng.standard_normal(5).astype(np.float32)
    x = rng.standard_normal((11, 7))
    for with_beta in (True, False):
        params = {"s/fully_connected/weights": w,
                  "s/fully_connected/BatchNorm/moving_mean": mean,
                  "s/fully_connected/BatchNorm/moving_variance": var}
        if with_beta:
            params["s/fully_connected/BatchNorm/beta"] = beta
        wf, bf = gnn.ParamStore(params).fc("s/fully_connected")
        assert wf.dtype == np.float32 and bf.dtype == np.float32
        (w64, b64), = gn._layers(params, "s", np.float64)
        ref = gn.fully_connected(x, w64, b64, relu=False)
        np.testing.assert_allclose(x @ wf.astype(np.float64) + bf, ref,
                                   rtol=0, atol=5e-6)
    # scale=True (not one of the registry's kinds, but what a checkpoint of
    # slim.batch_norm(scale=True) would hold): gamma multiplies the scale
    gamma = rng.uniform(0.5, 1.5, 5).astype(np.float32)
    params["s/fully_connected/BatchNorm/beta"] = beta
    params["s/fully_connected/BatchNorm/gamma"] = gamma
    wf, bf = gnn.ParamStore(params).fc("s/fully_connected")
    (w64, b64), = gn._layers(params, "s", np.float64)
    np.testing.assert_allclose(x @ wf.astype(np.float64) + bf,
                               gn.fully_connected(x, w64, b64, relu=False),
                               rtol=0, atol=5e-6)
    with pytest.raises(KeyError, match="neither biases nor BatchNorm"):
        gnn.ParamStore({"s/fully_connected/weights": w}).fc("s/fully_connected")
    gnn._check_kinds("ReLU", "fused_BN_center")
    with pytest.raises(NotImplementedError, match="normalization 'IN'"):
        gnn._check_kinds("ReLU", "IN")
    with pytest.raises(NotImplementedError, match="activation 'ELU'"):
        gnn._check_kinds("ELU", "NONE")
    cfg = copy.deepcopy(configs.get_config("car_auto_T3"))
    cfg["model_kwargs"]["layer_configs"][1]["kwargs"][
        "edge_MLP_normalization_type"] = "BN"
    import torch
    if not torch.cuda.is_available():
        # the check sits behind the Trainer's no-CPU-fallback guard: call it on
        # the config the way __init__ does
        with pytest.raises(NotImplementedError, match="no training path"):
            train.check_trainable_kinds(cfg)
    train.check_trainable_kinds(configs.get_config("car_auto_T3"))
