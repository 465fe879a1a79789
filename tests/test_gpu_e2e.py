"""One frame through every row of SURVEY.md §8 on the device, the way run.py
strings them together (run.py:203-433): KITTI files -> camera-frame crop ->
graph -> GNN with the reference's TRAINED car_auto_T1 weights (read from a TF
checkpoint written by our writer) -> softmax -> decode + NMS -> KITTI txt; the
oracle walks the same chain on the host from the same files."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs
from oracle import detect_oracle as DO
from oracle import gnn_oracle as gn
from oracle import graph_oracle as go
from oracle import ingest_oracle as IO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _velodyne_scan(seed):
    """The ray-cast synthetic scene (camera frame) mapped back into the
    velodyne frame, plus returns behind the car that the crop must drop."""
    from pointgnn_amd.synthetic import synthetic_cloud
    xyz_cam, inten = synthetic_cloud(seed=seed, preset="small")
    calib = IO.get_calib(IO.CALIB_LINES)
    homo = np.hstack([xyz_cam.astype(np.float64), np.ones((len(xyz_cam), 1))])
    front = (homo @ calib["cam_to_velo"].T)[:, :3]
    back = IO.synthetic_velo_scan(seed, n=15000)
    back = back[back[:, 0] < -1.0]
    scan = np.vstack([np.hstack([front, inten]), back]).astype(np.float32)
    return scan[np.random.default_rng(seed).permutation(len(scan))]


@pytest.mark.parametrize("weights_kind", ["trained", "seeded"])
def test_files_to_kitti_txt(tmp_path, weights_kind):
    import torch
    from test_ingest_cpu import _write_png_header_only
    from pointgnn_amd import weights
    from pointgnn_amd import (kitti_dataset as KD, graph_gen, models, nms,
                              kitti_output as KO, tf_bundle)
    cfg = configs.get_config("car_auto_T1")
    # ---- a KITTI-layout directory and a TF checkpoint on disk
    for d in ("image_2", "velodyne", "calib", "ckpt"):
        (tmp_path / d).mkdir()
    velo = _velodyne_scan(21)
    velo.tofile(str(tmp_path / "velodyne" / "000042.bin"))
    (tmp_path / "calib" / "000042.txt").write_text("".join(IO.CALIB_LINES))
    _write_png_header_only(str(tmp_path / "image_2" / "000042.png"), 375, 1242)
    if weights_kind == "trained":
        gold_w = np.load(os.path.join(GOLD, "weights_car_auto_T1.npz"))
        gold_w = {k: gold_w[k] for k in gold_w.files}
    else:   # near-uniform probabilities: the detection tail has work to do
        gold_w = weights.init_params(cfg, seed=3, bias_scale=0.05)
    tf_bundle.save_checkpoint(str(tmp_path / "ckpt"), gold_w,
                              global_step=1400000)
    # ---- device chain: pointgnn_amd.run = run.py's frame loop
    from pointgnn_amd import run as RUN
    ds = KD.KittiDataset(str(tmp_path / "image_2"), str(tmp_path / "velodyne"),
                         str(tmp_path / "calib"))
    td = RUN.run_dataset(ds, cfg, str(tmp_path / "ckpt"), str(tmp_path / "out"))
    assert td['frames'] == 1 and td['gnn inference'] > 0
    out_file = str(tmp_path / "out" / "data" / (ds.get_filename(0) + ".txt"))
    model = RUN.build_model(cfg, str(tmp_path / "ckpt"))
    rows, st = RUN.detect_frame(ds, 0, model, cfg)
    params = tf_bundle.load_checkpoint(str(tmp_path / "ckpt"))
    params.pop("Variable")
    pts, coords, kps, edges = st['points'], st['coords'], st['kps'], st['edges']
    logits, box_enc, probs = st['logits'], st['box_encodings'], st['probs']
    labels, boxes, scores, idx = (st['class_labels'], st['boxes_3d'],
                                  st['scores'], st['nms_indices'])
    cand_idx = st['candidate_indices']
    lmap = KO.LABEL_MAPS[cfg["label_method"]]
    # the file run_dataset wrote holds exactly these rows
    import io
    buf = str(tmp_path / "again.txt")
    KO.write_kitti_txt(buf, rows)
    assert open(buf).read() == open(out_file).read()

    # ---- the same chain in the oracle (GNN on the device-built graph: the
    # 'center' keypoint pick has legitimate ties)
    o_xyz, o_attr, _ = IO.cam_points_in_image(velo, IO.get_calib(IO.CALIB_LINES),
                                              (375, 1242))
    assert np.array_equal(pts.attr[:, :1].cpu().numpy(), o_attr)
    assert np.array_equal(pts.xyz.cpu().numpy(), o_xyz)
    c_np = [c.cpu().numpy() for c in coords]
    k_np = [k.cpu().numpy() for k in kps]
    e_np = [e.cpu().numpy() for e in edges]
    lcfg = cfg["runtime_graph_gen_kwargs"]["level_configs"]
    for lvl in (0, 1):
        ref_e = go.radius_graph_c(c_np[lvl], c_np[lvl + 1],
                                  lcfg[lvl]["graph_gen_kwargs"]["radius"])
        assert np.array_equal(go.canonical_edges(e_np[lvl]),
                              go.canonical_edges(ref_e))
    o_logits, o_enc = gn.predict(dict(params), cfg, o_attr, c_np, k_np, e_np,
                                 dtype=np.float64)
    np.testing.assert_allclose(logits.cpu().numpy(), o_logits, atol=2e-4,
                               rtol=1e-4)
    o_probs = gn.softmax(o_logits).astype(np.float32)
    if weights_kind == "seeded":
        # probabilities sit right at the 1/nc threshold here: give the oracle
        # tail the device's float32 outputs, so that it checks the detection
        # stage and not which side of 0.25 a 1e-7 difference falls
        o_probs = probs.cpu().numpy()
        o_enc = box_enc.cpu().numpy()
    w_idx, w_lab = DO.select_candidates(o_probs)
    assert np.array_equal(cand_idx.cpu().numpy(), w_idx)
    nc = cfg["num_classes"]
    dec = DO.box_decoding(np.tile(np.arange(nc), len(c_np[-1])).reshape(-1, 1),
                          np.repeat(c_np[-1], nc, axis=0),
                          o_enc.astype(np.float32).reshape(-1, 1, 7), lmap)
    want = DO.nms_boxes_3d(w_lab, dec[w_idx, 0], o_probs.reshape(-1)[w_idx],
                           cfg["nms_overlapped_thres"], "uncertainty")
    assert np.array_equal(idx.cpu().numpy(), want[3])
    assert np.array_equal(labels.cpu().numpy(), want[0])
    np.testing.assert_allclose(boxes.cpu().numpy(), want[1], atol=1e-3)
    np.testing.assert_allclose(scores.cpu().numpy(), want[2], rtol=1e-3)
    o_rows = DO.kitti_labels(want[0], want[1], want[2],
                             IO.get_calib(IO.CALIB_LINES)["cam_to_image"],
                             cfg["label_method"],
                             c_np[-1][w_idx // nc])
    assert [r[0] for r in rows] == [r[0] for r in o_rows]
    if rows:
        np.testing.assert_allclose(
            np.array([r[4:] for r in rows], np.float64),
            np.array([r[4:] for r in o_rows], np.float64), rtol=2e-3,
            atol=2e-3)
    # ---- the file is KITTI-shaped
    lines = [l for l in open(out_file).read().split("\n") if l.strip()]
    assert len(lines) == len(rows)
    for l in lines:
        f = l.split()
        assert len(f) == 16 and f[0] in ("Car", "DontCare", "Background")
    if weights_kind == "seeded":
        assert len(w_idx) > 100 and 0 < len(rows) <= len(want[3])
    print("N %d K %d candidates %d kept %d lines %d" % (
        len(o_xyz), len(c_np[-1]), len(w_idx), len(want[3]), len(rows)))


def test_frame_loop_with_the_f16x2_arithmetic(tmp_path):
    """run.py's frame loop with edge_arith='f16x2' (extension of build_model /
    run_dataset): the same detections as the fp32 loop to fp32 noise, and a
    frame whose activations leave fp16's safe range is rerun in fp32."""
    import torch
    from test_ingest_cpu import _write_png_header_only
    from pointgnn_amd import _lib, kitti_dataset as KD, tf_bundle
    from pointgnn_amd import run as RUN
    cfg = configs.get_config("car_auto_T1")
    for d in ("image_2", "velodyne", "calib", "ckpt"):
        (tmp_path / d).mkdir()
    _velodyne_scan(21).tofile(str(tmp_path / "velodyne" / "000042.bin"))
    (tmp_path / "calib" / "000042.txt").write_text("".join(IO.CALIB_LINES))
    _write_png_header_only(str(tmp_path / "image_2" / "000042.png"), 375, 1242)
    gold_w = np.load(os.path.join(GOLD, "weights_car_auto_T1.npz"))
    tf_bundle.save_checkpoint(str(tmp_path / "ckpt"),
                              {k: gold_w[k] for k in gold_w.files},
                              global_step=1400000)
    ds = KD.KittiDataset(str(tmp_path / "image_2"), str(tmp_path / "velodyne"),
                         str(tmp_path / "calib"))
    m32 = RUN.build_model(cfg, str(tmp_path / "ckpt"))
    m16 = RUN.build_model(cfg, str(tmp_path / "ckpt"), edge_arith="f16x2")
    assert m16.edge_arith == "f16x2" and m32.edge_arith == "f32"
    _lib.set_tunable("b16_force", 1)      # (a small frame: below the kernels'
    try:                                  # size threshold otherwise)
        rows32, st32 = RUN.detect_frame(ds, 0, m32, cfg)
        td = {}
        rows16, st16 = RUN.detect_frame(ds, 0, m16, cfg, time_dict=td)
        assert 'f16x2 range reruns' not in td
        l32, l16 = st32['logits'], st16['logits']
        assert not torch.equal(l32, l16), "the f16x2 kernels did not run"
        assert float((l32 - l16).abs().max()) <= 2e-5 * float(l32.abs().max())
        assert [r[0] for r in rows16] == [r[0] for r in rows32]
        if rows32:
            np.testing.assert_allclose(
                np.array([r[4:] for r in rows16], np.float64),
                np.array([r[4:] for r in rows32], np.float64), rtol=2e-3,
                atol=2e-3)
        # a frame flagged by the range guard: rerun in fp32, model unchanged
        m16.edge_range_ok = lambda: False
        td = {}
        _, st = RUN.detect_frame(ds, 0, m16, cfg, time_dict=td)
        assert td['f16x2 range reruns'] == 1 and m16.edge_arith == "f16x2"
        assert torch.equal(st['logits'], l32)
    finally:
        _lib.set_tunable("b16_force", 0)


def test_training_sample_from_kitti_files(tmp_path):
    """train.py:78-133 (`fetch_data`) on the device: KITTI files + label file
    -> crop -> augmentations -> training-mode graph -> label assignment ->
    box encoding -> one Trainer step; targets checked against the oracle on
    the same augmented cloud."""
    import copy
    import torch
    from test_ingest_cpu import _write_png_header_only
    from oracle import labels_oracle as LO
    from pointgnn_amd import (kitti_dataset as KD, graph_gen, preprocess as PP,
                              box_encoding as BE, train, weights)
    cfg = configs.get_config("car_auto_T1")
    for d in ("image_2", "velodyne", "calib", "label_2"):
        (tmp_path / d).mkdir()
    velo = _velodyne_scan(5)
    velo.tofile(str(tmp_path / "velodyne" / "000001.bin"))
    (tmp_path / "calib" / "000001.txt").write_text("".join(IO.CALIB_LINES))
    _write_png_header_only(str(tmp_path / "image_2" / "000001.png"), 375, 1242)
    cam, _, _ = IO.cam_points_in_image(velo, IO.get_calib(IO.CALIB_LINES),
                                       (375, 1242))
    gt = LO.synthetic_labels(5, cam, n_boxes=12)
    LO.write_label_file(str(tmp_path / "label_2" / "000001.txt"), gt)
    ds = KD.KittiDataset(str(tmp_path / "image_2"), str(tmp_path / "velodyne"),
                         str(tmp_path / "calib"), str(tmp_path / "label_2"),
                         is_training=True, num_classes=cfg["num_classes"])
    pts = ds.get_cam_points_in_image_with_rgb(0)
    labels = ds.get_label(0)
    assert labels == gt
    aug = PP.get_data_aug([
        {"method_name": "random_rotation_all",
         "method_kwargs": {"method_name": "normal", "yaw_std": 0.39,
                           "expend_factor": (1.0, 1.0, 1.0)}},
        {"method_name": "random_flip_all", "method_kwargs": {"flip_prob": 0.5}},
        {"method_name": "random_box_shift",
         "method_kwargs": {"appr_factor": 10, "expend_factor": (1.1, 1.1, 1.1),
                           "max_overlap_num_allowed": 100,
                           "max_overlap_rate": 0.01, "max_trails": 100,
                           "method_name": "normal", "xyz_std": (3, 0, 3)}}])
    np.random.seed(5)
    pts, labels = aug(pts, copy.deepcopy(labels))
    pts = PP.finish(pts)
    fn = graph_gen.get_graph_generate_fn(cfg["graph_gen_method"])
    coords, kps, edges = fn(pts.xyz, **cfg["graph_gen_kwargs"])
    last = coords[-1]
    cls, boxes3d, valid, lmap = ds.assign_classaware_car_label_to_points(
        labels, last, expend_factor=(1.0, 1.0, 1.0))
    enc = BE.get_box_encoding_fn(cfg["box_encoding_method"])(
        cls, last, boxes3d, lmap)
    # oracle targets on the same vertices / labels
    o_cls, o_boxes, o_valid, _ = LO.assign_labels(labels, last.cpu().numpy(),
                                                  (1.0, 1.0, 1.0), "Car")
    assert np.array_equal(cls.cpu().numpy(), o_cls)
    assert np.array_equal(boxes3d.cpu().numpy(), o_boxes)
    assert np.array_equal(valid.cpu().numpy(), o_valid)
    o_enc = DO.box_encoding(o_cls, last.cpu().numpy(), o_boxes, lmap
                            ).astype(np.float32)
    tol = np.spacing(np.maximum(np.abs(o_enc), 1e-30))
    assert np.all(np.abs(enc.cpu().numpy().astype(np.float64) - o_enc) <= tol)
    assert int((o_cls > 0).sum()) > 0
    # one optimisation step on this sample
    tr = train.Trainer(cfg, params=weights.init_params(cfg, seed=1),
                       device=last.device)
    batch = (pts.attr[:, :1].contiguous(), coords, kps, edges, cls, enc, valid)
    out = tr.train_step(batch, num_valid=float(o_valid.sum()))
    assert np.isfinite([out['cls_loss'], out['loc_loss'], out['reg_loss']]).all()
    assert out['num_endpoint'] == len(o_cls)
    assert out['num_valid_endpoint'] == float(o_valid.sum())


def test_train_epochs_on_kitti_files(tmp_path):
    """train.py's epoch loop (pointgnn_amd.train.train_epochs) on a three-frame
    KITTI directory: fetch_data -> batch_data -> train_step -> metrics ->
    checkpoints; a second call resumes from the checkpoint it finds."""
    import torch
    from test_ingest_cpu import _write_png_header_only
    from oracle import labels_oracle as LO
    from pointgnn_amd import kitti_dataset as KD, tf_bundle, train
    cfg = configs.get_config("car_auto_T1")
    for d in ("image_2", "velodyne", "calib", "label_2"):
        (tmp_path / d).mkdir()
    for i in range(3):
        name = "%06d" % i
        velo = _velodyne_scan(30 + i)
        velo.tofile(str(tmp_path / "velodyne" / (name + ".bin")))
        (tmp_path / "calib" / (name + ".txt")).write_text("".join(IO.CALIB_LINES))
        _write_png_header_only(str(tmp_path / "image_2" / (name + ".png")),
                               375, 1242)
        cam, _, _ = IO.cam_points_in_image(velo, IO.get_calib(IO.CALIB_LINES),
                                           (375, 1242))
        LO.write_label_file(str(tmp_path / "label_2" / (name + ".txt")),
                            LO.synthetic_labels(30 + i, cam, n_boxes=10))
    ds = KD.KittiDataset(str(tmp_path / "image_2"), str(tmp_path / "velodyne"),
                         str(tmp_path / "calib"), str(tmp_path / "label_2"),
                         is_training=True, num_classes=cfg["num_classes"])
    tcfg = {
        'train_dir': str(tmp_path / "ckpt"), 'batch_size': 1, 'max_epoch': 2,
        'save_every_epoch': 1, 'initial_lr': 0.05, 'decay_step': 4,
        'decay_factor': 0.5, 'optimizer': 'sgd', 'unify_copies': True,
        'NUM_TEST_SAMPLE': -1, 'config_path': 'config',
        'data_aug_configs': [
            {"method_name": "random_rotation_all",
             "method_kwargs": {"method_name": "normal", "yaw_std": 0.39,
                               "expend_factor": (1.0, 1.0, 1.0)}},
            {"method_name": "random_flip_all",
             "method_kwargs": {"flip_prob": 0.5}}]}
    np.random.seed(3)
    # one sample by itself: the 7-tuple of train.py:78-133
    sample = train.fetch_data(ds, 1, cfg, tcfg)
    k = int(sample[1][-1].shape[0])
    assert sample[0].shape[1] == 1 and sample[4].shape == (k, 1)
    assert sample[5].shape == (k, 1, 7) and sample[6].shape == (k, 1, 1)
    assert sample[5].dtype == torch.float32 and sample[4].dtype == torch.int32
    # the same sample with its graph built in capacity form (one host read per
    # frame; train_epochs fetches this way): identical tensors for the same
    # RNG state, from cold hints (overflow -> host-sized rebuild) and learned
    from pointgnn_amd import graph_gen
    hints = graph_gen.CountHints()
    for trial in range(2):
        np.random.seed(3)
        again = train.fetch_data(ds, 1, cfg, tcfg, graph_hints=hints)
        for a, b in zip(sample, again):
            for x, y in zip(a if isinstance(a, (list, tuple)) else [a],
                            b if isinstance(b, (list, tuple)) else [b]):
                assert x.dtype == y.dtype and torch.equal(x, y)
    assert hints.k == int(sample[1][1].shape[0])
    lines = []
    tr, res = train.train_epochs(ds, cfg, tcfg, log=lines.append)
    assert tr.global_step == 6 and res['step'] == 6
    assert res['learning_rate'] == pytest.approx(0.05 * 0.5)   # step 5 // 4
    for key in ('cls_loss', 'loc_loss', 'reg_loss', 'total_loss', 'recall_0',
                'precision_1', 'mAP_1'):
        assert np.isfinite(res[key]), key
    assert any(l.startswith('STEP: 6, epoch_idx: 1') for l in lines)
    assert any('Class_1: recall=' in l for l in lines)
    ck = tf_bundle.load_checkpoint(tcfg['train_dir'])
    assert int(ck['Variable']) == 6
    assert os.path.exists(os.path.join(tcfg['train_dir'], 'config'))
    assert os.path.exists(os.path.join(tcfg['train_dir'], 'train_config'))
    w6 = tr.state_dict()
    for name, v in w6.items():
        assert np.array_equal(ck[name], v), name
    # resume: a new trainer picks the checkpoint up and runs epoch 2 only
    tcfg2 = dict(tcfg, max_epoch=3)
    tr2, res2 = train.train_epochs(ds, cfg, tcfg2)
    assert tr2.global_step == 9
    assert int(tf_bundle.load_checkpoint(tcfg['train_dir'])['Variable']) == 9
    # ---- eval.py's pass over the same files with the last checkpoint
    from pointgnn_amd import eval as EV
    ev_lines = []
    ev = EV.eval_once(ds, cfg, {'data_aug_configs': [], 'NUM_TEST_SAMPLE': -1},
                      checkpoint_dir=tcfg['train_dir'], log=ev_lines.append)
    assert ev['step'] == 9
    for c in range(cfg['num_classes']):
        for key in ('recall_%d', 'precision_%d', 'mAP_%d', 'loc_loss_cls_%d'):
            assert np.isfinite(ev[key % c])
        assert 'loc_loss_cls_%d_box_6' % c in ev
    assert np.isfinite(ev['total_loss']) and ev['total_loss'] > 0
    assert ev_lines[0].startswith('STEP: 9')


def _kitti_tree(tmp_path, presets):
    """A KITTI-object tree of synthetic frames (one preset per frame) written
    by the product's own generator."""
    from pointgnn_amd import synthetic as S
    root = str(tmp_path / "kitti")
    for i, preset in enumerate(presets):
        S.write_kitti_frames(root, [i], preset=preset, behind_points=4000)
    return [os.path.join(root, d) for d in ("image_2", "velodyne", "calib")]


@pytest.mark.parametrize("edge_arith", ["f32", "f16x2"])
def test_pipelined_frame_loop_writes_the_sequential_loops_files(tmp_path,
                                                                edge_arith):
    """run_dataset with frames in flight (loader thread, capacity-form graph +
    GNN on 1..3 streams, decode + NMS behind them, host-side rows in a writer
    thread) against the strictly sequential loop of run.py:203-433: every
    output file byte for byte, for frames of different sizes, including a
    frame that outgrows the capacity its predecessors set (sequential
    fallback).  Under 'f16x2' (a secondary arithmetic) the edge stage of a
    SMALL graph runs the fp32 kernel, and the capacity form takes that decision
    from the previous frame's size: the files then agree field by field to
    1e-3 instead of byte for byte."""
    import torch
    from pointgnn_amd import kitti_dataset as KD, run as RUN, weights
    cfg = configs.get_config("car_auto_T1")
    presets = ["tiny", "tiny", "small", "tiny", "small", "car", "small", "tiny",
               "small"]
    dirs = _kitti_tree(tmp_path, presets)
    ds = KD.KittiDataset(*dirs)
    assert ds.num_files == len(presets)
    # near-uniform probabilities: hundreds of candidates, dozens of rows
    params = weights.init_params(cfg, seed=3, bias_scale=0.05)
    seq_dir = str(tmp_path / "seq")
    td0 = RUN.run_dataset(ds, cfg, None, seq_dir, params=params,
                          edge_arith=edge_arith, pipelined=False)
    assert td0['frames'] == len(presets)
    want = {}
    for i in range(ds.num_files):
        with open(os.path.join(seq_dir, "data",
                               ds.get_filename(i) + ".txt"), "rb") as f:
            want[i] = f.read()
    assert sum(len(w) > 1 for w in want.values()) >= 3, "no detections at all"
    for in_flight in (1, 2, 3):
        out = str(tmp_path / ("pipe%d" % in_flight))
        td = RUN.run_dataset(ds, cfg, None, out, params=params,
                             edge_arith=edge_arith, in_flight=in_flight,
                             prefetch=2)
        torch.cuda.synchronize()
        assert td['frames'] == len(presets)
        for key in ('fetch input', 'gen graph', 'gnn inference',
                    'decode box + nms', 'kitti rows', 'write txt', 'wall'):
            assert td[key] > 0, key
        # 'car' after tiny / small frames outgrows their edge capacities
        assert td['sequential fallbacks'] >= 1
        for i in range(ds.num_files):
            with open(os.path.join(out, "data",
                                   ds.get_filename(i) + ".txt"), "rb") as f:
                got = f.read()
            if edge_arith == "f32":
                assert got == want[i], (in_flight, i)
                continue
            a = [l.split() for l in got.decode().split("\n") if l.strip()]
            b = [l.split() for l in want[i].decode().split("\n") if l.strip()]
            assert [r[0] for r in a] == [r[0] for r in b], (in_flight, i)
            if a:
                np.testing.assert_allclose(
                    np.array([r[4:] for r in a], np.float64),
                    np.array([r[4:] for r in b], np.float64), rtol=1e-3,
                    atol=1e-3)


def test_pipelined_frame_loop_ped_cyl_split_pooling(tmp_path):
    """The same for `ped_cyl_auto_T3` (configs/ped_cyl_auto_T3_*: two classes
    with their own box codec, 4-32-64-128-256-512 point MLP): its pooling stage
    runs as two launches through a workspace of hidden rows
    (csrc/pool_split.h; forced below its size threshold here), in the
    sequential loop with host-sized counts and in the pipelined one in capacity
    form -- every output file byte for byte."""
    import torch
    from pointgnn_amd import _lib, kitti_dataset as KD, run as RUN, weights
    cfg = configs.get_config("ped_cyl_auto_T3")
    presets = ["small", "tiny", "car", "small", "car"]
    dirs = _kitti_tree(tmp_path, presets)
    ds = KD.KittiDataset(*dirs)
    params = weights.init_params(cfg, seed=5, bias_scale=0.05)
    try:
        _lib.set_tunable("mlp_debug", 16384)
        seq_dir = str(tmp_path / "seq")
        td0 = RUN.run_dataset(ds, cfg, None, seq_dir, params=params,
                              pipelined=False)
        assert td0['frames'] == len(presets)
        out = str(tmp_path / "pipe")
        td = RUN.run_dataset(ds, cfg, None, out, params=params, in_flight=3,
                             prefetch=2)
        torch.cuda.synchronize()
        assert td['frames'] == len(presets)
    finally:
        _lib.set_tunable("mlp_debug", 0)
    rows = 0
    for i in range(ds.num_files):
        name = ds.get_filename(i) + ".txt"
        with open(os.path.join(seq_dir, "data", name), "rb") as f:
            want = f.read()
        with open(os.path.join(out, "data", name), "rb") as f:
            assert f.read() == want, i
        rows += want.count(b"\n")
        assert all(l.split()[0] in (b"Pedestrian", b"Cyclist")
                   for l in want.split(b"\n") if l.strip())
    assert rows >= 3, "no detections at all"
    # the tile kernel writes the same files
    try:
        _lib.set_tunable("mlp_debug", 8192)
        ref_dir = str(tmp_path / "tile")
        RUN.run_dataset(ds, cfg, None, ref_dir, params=params, pipelined=False)
    finally:
        _lib.set_tunable("mlp_debug", 0)
    for i in range(ds.num_files):
        name = ds.get_filename(i) + ".txt"
        with open(os.path.join(ref_dir, "data", name), "rb") as f, \
                open(os.path.join(seq_dir, "data", name), "rb") as g:
            assert f.read() == g.read(), i


def test_inside_box_host_equals_device_kernel():
    """kitti_output.inside_box_host (the pipelined loop's occlusion test) ==
    kitti_dataset.sel_xyz_in_box3d on the device, including points placed ON
    the box faces (strict inequalities)."""
    import torch
    from pointgnn_amd import kitti_dataset as KD, kitti_output as KO
    rng = np.random.default_rng(4)
    for trial in range(20):
        label = {'x3d': rng.uniform(-10, 10), 'y3d': rng.uniform(0, 2),
                 'z3d': rng.uniform(5, 40), 'yaw': rng.uniform(-3.2, 3.2),
                 'height': rng.uniform(1, 3), 'width': rng.uniform(1, 3),
                 'length': rng.uniform(2, 6)}
        corners = KD.box3d_to_cam_points(label).xyz
        mids = (corners[None] + corners[:, None]).reshape(-1, 3) / 2
        pts = np.vstack([
            mids, corners,
            np.array([label['x3d'], label['y3d'], label['z3d']]) +
            rng.normal(0, 2.0, (2000, 3))]).astype(np.float32)
        dev = KD.sel_xyz_in_box3d(label, torch.from_numpy(pts).cuda())
        host = KO.inside_box_host(label, pts)
        assert np.array_equal(dev.cpu().numpy(), host)
        assert 0 < host.sum() < len(pts)


def test_pipelined_frame_loop_propagates_a_loader_failure(tmp_path):
    """A frame the loader cannot read (truncated velodyne file) ends the
    pipelined loop with that error -- no hang, no half-alive threads -- and the
    frames before it are on disk."""
    import threading
    from pointgnn_amd import kitti_dataset as KD, run as RUN, weights
    cfg = configs.get_config("car_auto_T1")
    dirs = _kitti_tree(tmp_path, ["tiny"] * 8)
    ds = KD.KittiDataset(*dirs)
    bad = os.path.join(dirs[1], ds.get_filename(5) + ".bin")
    with open(bad, "r+b") as f:
        f.truncate(4 * 4 * 100 + 6)          # not a whole number of points
    params = weights.init_params(cfg, seed=3, bias_scale=0.05)
    before = threading.active_count()
    with pytest.raises(ValueError):
        RUN.run_dataset(ds, cfg, None, str(tmp_path / "out"), params=params,
                        in_flight=2, prefetch=2)
    assert threading.active_count() == before
    assert os.path.exists(os.path.join(str(tmp_path / "out"), "data",
                                       ds.get_filename(0) + ".txt"))
    # the same tree without the bad frame runs through
    td = RUN.run_dataset(ds, cfg, None, str(tmp_path / "out2"), params=params,
                         frame_indices=[0, 1, 2, 3, 4, 6, 7], in_flight=2)
    assert td['frames'] == 7
