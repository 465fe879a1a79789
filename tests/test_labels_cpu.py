"""Training-target oracle (oracle/labels_oracle.py) and the host half of the
product (label parsing, box -> normals records) against the fixture written
by the reference's own code (tests/golden/make_golden_labels.py).  No GPU."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from oracle import detect_oracle as DO
from oracle import labels_oracle as LO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXPEND = {"e1": (1.0, 1.0, 1.0), "e2": (1.1, 1.1, 1.1)}


@pytest.fixture(scope="module")
def fix():
    return np.load(os.path.join(GOLD, "labels_kitti.npz"))


@pytest.fixture(scope="module")
def scene():
    xyz = LO.synthetic_vertices(0)
    return xyz, LO.synthetic_labels(0, xyz, n_boxes=60)


def test_label_file_round_trip_and_difficulty_filter(fix, scene, tmp_path):
    from pointgnn_amd import kitti_dataset as KD
    _, labels = scene
    path = str(tmp_path / "000000.txt")
    LO.write_label_file(path, labels)
    for reader in (LO.get_label, KD.read_label_file):
        assert reader(path) == labels
        for d in (0, 1, 2):
            assert len(reader(path, d)) == int(fix["label_count_d%d" % d])
    assert int(fix["label_count_d-1"]) == len(labels)
    assert int(fix["label_count_d0"]) < int(fix["label_count_d2"]) < len(labels)


def test_normals_and_mask_equal_reference(fix, scene):
    from pointgnn_amd import kitti_dataset as KD
    xyz, labels = scene
    for i in (0, 5, 11):
        for fn in (LO.box_normals, KD.box3d_to_normals):
            n, lo, up = fn(labels[i], (1.1, 1.2, 1.3))
            np.testing.assert_allclose(n, fix["normals_%d" % i], rtol=0,
                                       atol=1e-13)
            np.testing.assert_allclose(lo, fix["lower_%d" % i], rtol=1e-13)
            np.testing.assert_allclose(up, fix["upper_%d" % i], rtol=1e-13)
        m = LO.sel_xyz_in_box3d(labels[i], xyz, (1.1, 1.2, 1.3))
        assert np.array_equal(m, fix["mask_%d" % i])
    corners = KD.box3d_to_cam_points(labels[0]).xyz
    np.testing.assert_allclose(corners, LO.box_corners(labels[0]), atol=1e-13)
    assert corners.shape == (8, 3)
    # the box bottom sits at y3d (camera y points down), the top at y3d - h
    assert corners[:, 1].max() == pytest.approx(labels[0]['y3d'])
    assert corners[:, 1].min() == pytest.approx(
        labels[0]['y3d'] - labels[0]['height'])


@pytest.mark.parametrize("method", ["yaw", "Car", "Pedestrian_and_Cyclist"])
@pytest.mark.parametrize("tag", ["e1", "e2"])
def test_assign_oracle_equals_reference(fix, scene, method, tag):
    xyz, labels = scene
    cls, boxes, valid, lm = LO.assign_labels(labels, xyz, EXPEND[tag], method)
    pre = "%s_%s_" % (method, tag)
    assert np.array_equal(cls, fix[pre + "cls"])
    assert np.array_equal(boxes, fix[pre + "boxes"])
    assert np.array_equal(valid, fix[pre + "valid"])
    assert cls.dtype == np.int64 and boxes.dtype == np.float64
    enc = DO.box_encoding(cls, xyz, boxes, lm).astype(np.float32)
    assert np.array_equal(enc, fix[pre + "encoded"])
    # overlapping boxes exist: some vertex lies in more than one box
    hits = sum(LO.sel_xyz_in_box3d(l, xyz, EXPEND[tag]).astype(int)
               for l in labels if l['name'] != 'DontCare')
    assert hits.max() >= 2


def test_host_records_follow_the_reference_rules(scene):
    from pointgnn_amd import kitti_dataset as KD
    _, labels = scene
    lm = LO.LABEL_MAPS["Car"][0]
    rec = KD._label_records(labels, (1.1, 1.1, 1.1), lm)
    assert rec.shape == (len(labels), 24)
    for r, lab in zip(rec, labels):
        if lab['name'] == 'DontCare':
            assert r[15] == 0.0 and not r.any()
        elif lab['name'] == 'Car':
            assert r[15] == 1.0 and r[16] in (1.0, 2.0)
            assert -0.25 * np.pi <= r[23] <= 0.75 * np.pi
            assert (r[16] == 1.0) == (r[23] < 0.25 * np.pi)
            n, lo, up = LO.box_normals(lab, (1.1, 1.1, 1.1))
            np.testing.assert_allclose(r[:9], n.reshape(-1), atol=1e-13)
            np.testing.assert_allclose(r[9:12], lo, rtol=1e-13)
            np.testing.assert_allclose(r[12:15], up, rtol=1e-13)
        else:                       # Van, Pedestrian, ... under the Car map
            assert r[15] == 2.0 and r[16] == 3.0
