"""KITTI ingest: oracle restatement and the host-side calibration algebra
against the fixture written by the reference's own KittiDataset methods
(tests/golden/make_golden_ingest.py).  No GPU."""
import hashlib
import os
import struct
import zlib

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from oracle import ingest_oracle as IO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CALIB_KEYS = ("P2", "R0_rect", "Tr_velo_to_cam", "velo_to_rect",
              "cam_to_image", "rect_to_cam", "velo_to_cam", "cam_to_velo",
              "velo_to_image")


@pytest.fixture(scope="module")
def fix():
    return np.load(os.path.join(GOLD, "ingest_kitti.npz"))


def _inputs(fix):
    velo = IO.synthetic_velo_scan(0, n=60000)
    image = IO.synthetic_image(0)
    assert hashlib.sha1(velo.tobytes()).digest() == fix["velo_sha1"].tobytes()
    assert hashlib.sha1(image.tobytes()).digest() == fix["image_sha1"].tobytes()
    return velo, image


def test_oracle_calib_equals_reference(fix):
    calib = IO.get_calib(IO.CALIB_LINES)
    for k in CALIB_KEYS:
        assert calib[k].dtype == fix["calib_" + k].dtype, k
        assert np.array_equal(calib[k], fix["calib_" + k]), k


def test_host_parse_calib_equals_reference(fix, tmp_path):
    from pointgnn_amd import kitti_dataset as KD
    p = tmp_path / "000000.txt"
    p.write_text("".join(IO.CALIB_LINES) + "\n")      # KITTI files end blank
    for src in (str(p), IO.CALIB_LINES):
        calib = KD.parse_calib(src)
        for k in CALIB_KEYS:
            assert calib[k].dtype == fix["calib_" + k].dtype, k
            assert np.array_equal(calib[k], fix["calib_" + k]), k


def test_oracle_ingest_equals_reference(fix):
    velo, image = _inputs(fix)
    calib = IO.get_calib(IO.CALIB_LINES)
    cam = IO.velo_to_cam(velo[:, :3], calib)
    assert np.array_equal(cam[::16], fix["cam_every16"])
    xyz, attr, kept = IO.cam_points_in_image(velo, calib, image.shape[:2],
                                             image=image)
    assert np.array_equal(kept, fix["kept"])
    assert np.array_equal(xyz, fix["xyz"])
    assert np.array_equal(attr, fix["attr_rgb"])
    xyz1, attr1, _ = IO.cam_points_in_image(velo, calib, image.shape[:2])
    assert np.array_equal(xyz1, xyz) and np.array_equal(attr1, attr[:, :1])
    # about a sixth of a 360-degree scan is seen by the camera
    assert 0.1 < len(kept) / len(velo) < 0.35
    assert xyz[:, 2].min() > 0.1


def _write_png_header_only(path, height, width):
    ihdr = struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)
    chunk = b"IHDR" + ihdr
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + struct.pack(">I", len(ihdr)) + chunk +
                struct.pack(">I", zlib.crc32(chunk) & 0xffffffff))


def test_png_size_and_dataset_index(tmp_path):
    from pointgnn_amd import kitti_dataset as KD
    img = tmp_path / "image_2"
    pts = tmp_path / "velodyne"
    cal = tmp_path / "calib"
    for d in (img, pts, cal):
        d.mkdir()
    for name in ("000003", "000001"):
        _write_png_header_only(str(img / (name + ".png")), 375, 1242)
        IO.synthetic_velo_scan(1, n=100).tofile(str(pts / (name + ".bin")))
        (cal / (name + ".txt")).write_text("".join(IO.CALIB_LINES))
    assert KD.png_size(str(img / "000001.png")) == (375, 1242)
    with pytest.raises(ValueError):
        KD.png_size(str(cal / "000001.txt"))
    ds = KD.KittiDataset(str(img), str(pts), str(cal))
    assert ds.num_files == 2 and ds.get_filename(0) == "000001"
    v = ds.get_velo_points(1)
    assert v.xyz.shape == (100, 3) and v.attr.shape == (100, 1)
    cropped = ds.get_velo_points(1, xyz_range=((0, 80), (-40, 40), (-3, 1)))
    assert 0 < len(cropped.xyz) < 100 and cropped.xyz[:, 0].min() > 0
    assert np.array_equal(ds.get_calib(0)["P2"],
                          IO.get_calib(IO.CALIB_LINES)["P2"])
    with pytest.raises(NotImplementedError):
        KD.KittiDataset(str(img), str(pts), str(cal), is_raw=True)
    with pytest.raises(ValueError):
        KD.KittiDataset(str(img), str(pts), str(cal), is_training=True)
