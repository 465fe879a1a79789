"""Detections -> KITTI label lines (pointgnn_amd.kitti_output, run.py:360-433)
against the fixture composed from the reference's own functions; the inside
test runs on the device (pgnn_assign_box_labels)."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from oracle import ingest_oracle as IO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kitti_labels_and_txt(tmp_path):
    import torch
    from pointgnn_amd import kitti_output as KO, kitti_dataset as KD
    fix = np.load(os.path.join(GOLD, "detect_output.npz"))
    calib = KD.parse_calib(IO.CALIB_LINES)
    rows = KO.detections_to_kitti_labels(
        torch.from_numpy(fix["labels"]).cuda(),
        torch.from_numpy(fix["boxes"]).cuda(),
        torch.from_numpy(fix["scores"]).cuda(), calib, 'Car',
        candidate_xyz=torch.from_numpy(fix["cand_xyz"]).cuda())
    assert len(rows) == len(fix["rows"])
    assert [r[0] for r in rows] == list(fix["names"])
    assert all(r[1:4] == (-1, -1, 0) for r in rows)
    got = np.array([r[4:] for r in rows], np.float64)
    np.testing.assert_allclose(got, fix["rows"][:, :12], rtol=1e-12)
    plain = KO.detections_to_kitti_labels(
        fix["labels"], fix["boxes"], fix["scores"], calib, 'Car',
        use_box_score=False)
    np.testing.assert_array_equal(
        np.array([r[15] for r in plain]), fix["rows"][:, 12].astype(np.float32))
    # file format of run.py:423-433
    path = str(tmp_path / "data" / "000007.txt")
    KO.write_kitti_txt(path, rows)
    lines = open(path).read().split('\n')
    assert lines[-1] == '' and lines[-2] == '' and len(lines) == len(rows) + 2
    f0 = lines[0].split(' ')
    assert f0[0] == rows[0][0] and f0[1:4] == ['-1', '-1', '0']
    assert len(f0) == 17 and f0[16] == ''
    assert float(f0[15]) == pytest.approx(float(rows[0][15]))
    KO.write_kitti_txt(str(tmp_path / "empty.txt"), [])
    assert open(str(tmp_path / "empty.txt")).read() == '\n'
