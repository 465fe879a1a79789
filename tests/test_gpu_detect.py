"""HIP detection post-processing (pgnn_box_*_f32, pgnn_detection_candidates,
pgnn_nms_boxes_3d, pgnn_overlapped_boxes_3d through the C-ABI) against the
fixtures written by the reference's own code and against the oracle.

Bars: labels / kept indices / candidate indices bit-exact; decoded boxes within
2 float32 ulp of NumPy (exp/log/cos/sin are evaluated in float64 and rounded
once on the device, NumPy uses float32 routines that are ~1.5 ulp accurate);
overlaps within 5e-7 relative (one float32 ulp in cos/sin of the corners;
float64 clipping vs the oracle's vertex-collection algorithm beyond that);
merged boxes exact (medians of the inputs); accumulated scores within 1e-6
relative.  A pair whose overlap sits within that distance of the threshold
could in principle be decided differently; none of the seeded cases has one."""
import os

import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from oracle import detect_oracle as DO

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LABEL_MAPS = {
    "car": {'Background': 0, 'Car': 1, 'DontCare': 3},
    "ped": {'Background': 0, 'Pedestrian': 1, 'Cyclist': 3, 'DontCare': 5},
}
MODES = ("plain", "uncertainty", "merge_only", "score_only")


def _fn(mode):
    from pointgnn_amd import nms
    return {"plain": nms.nms_boxes_3d,
            "uncertainty": nms.nms_boxes_3d_uncertainty,
            "merge_only": nms.nms_boxes_3d_merge_only,
            "score_only": nms.nms_boxes_3d_score_only}[mode]


def _ulp_close(a, b, ulps):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return np.all(np.abs(a.astype(np.float64) - b) <= tol)


@pytest.mark.parametrize("name", ["car", "ped"])
def test_codec_matches_reference_fixture(name):
    from pointgnn_amd import box_encoding as BE
    codec = np.load(os.path.join(GOLD, "detect_codec.npz"))
    lm = LABEL_MAPS[name]
    g = lambda k: codec["codec_%s_%s" % (name, k)]  # noqa: E731
    dec_fn = BE.get_box_decoding_fn('classaware_all_class_box_encoding')
    enc_fn = BE.get_box_encoding_fn('classaware_all_class_box_encoding')
    dec = dec_fn(g("labels"), g("xyz"), g("encoded"), lm)
    assert dec.dtype == np.float32 and dec.shape == g("decoded").shape
    assert _ulp_close(dec, g("decoded"), 2)
    enc = enc_fn(g("labels"), g("xyz"), g("boxes"), lm)
    ref = g("boxes_encoded")
    # log() of a ratio near 1 is ill-conditioned in ulps; bound it absolutely
    np.testing.assert_allclose(enc, ref, rtol=3e-7, atol=3e-7)
    # untouched labels (background / don't care) are pure offset arithmetic
    plain = ~np.isin(g("labels")[:, 0], np.nonzero(BE.class_table(lm)[:, 4])[0])
    assert np.array_equal(enc[plain], ref[plain])
    assert np.array_equal(dec[plain], g("decoded")[plain])
    back = dec_fn(g("labels"), g("xyz"), enc, lm)
    np.testing.assert_allclose(back, g("boxes"), rtol=2e-5, atol=2e-5)
    assert BE.get_encoding_len('classaware_all_class_box_encoding') == 7
    with pytest.raises(NotImplementedError):
        BE.get_box_decoding_fn('voxelnet_box_encoding')


def test_codec_multi_column_and_tensor_io():
    import torch
    from pointgnn_amd import box_encoding as BE
    rng = np.random.default_rng(5)
    lm = LABEL_MAPS["ped"]
    labels = rng.integers(0, 6, (300, 1)).astype(np.int32)
    xyz = rng.uniform(-5, 5, (300, 3)).astype(np.float32)
    enc = rng.normal(0, 0.5, (300, 3, 7)).astype(np.float32)
    want = DO.box_decoding(labels, xyz, enc, lm)
    got = BE.classaware_all_class_box_decoding(
        torch.from_numpy(labels).cuda(), torch.from_numpy(xyz).cuda(),
        torch.from_numpy(enc).cuda(), lm)
    assert got.is_cuda
    assert _ulp_close(got.cpu().numpy(), want, 2)
    # columns 1.. only receive the xyz offset (box_encoding.py:293-298)
    assert np.array_equal(got.cpu().numpy()[:, 1:], want[:, 1:])


def test_pairwise_overlap_matches_reference_and_oracle():
    from pointgnn_amd import nms
    fix = np.load(os.path.join(GOLD, "detect_nms.npz"))
    boxes = fix["geom_boxes"]
    ov = nms.overlapped_boxes_3d_fast_poly(boxes[0], boxes[1:])
    assert ov.dtype == np.float64
    # the reference fixture's corners come from NumPy's float32 cos/sin (within
    # ~1.5 ulp, CPU-dispatch dependent); the device rounds the float64 value
    # once.  One float32 ulp in cos/sin moves the overlap by ~1e-7 relative.
    np.testing.assert_allclose(ov, fix["geom_overlap_row0"], rtol=5e-7,
                               atol=1e-15)
    assert (ov > 0).sum() >= 3
    # a denser random set against the oracle, every row as the single box
    labels, b2, _ = DO.synthetic_detections(11, n_objects=6, half_width=5.0,
                                            depth=(5.0, 15.0))
    corners = DO.boxes_3d_to_corners(b2)
    for i in range(0, len(b2), 17):
        want = DO.overlapped_boxes_3d_fast_poly(corners[i], corners)
        got = nms.overlapped_boxes_3d_fast_poly(b2[i], b2)
        # small overlaps are differences of nearly equal areas: absolute bound
        np.testing.assert_allclose(got, want, rtol=5e-7, atol=5e-9)
        assert got[i] == pytest.approx(1.0, rel=1e-6)   # a box with itself
    # integer-corner variant (nms.py:113-115)
    ci = np.int32(corners * 100.0)
    want = DO.overlapped_boxes_3d_fast_poly(ci[0], ci)
    got = nms.overlapped_boxes_3d_fast_poly(b2[0], b2, appr_factor=100.0)
    # integer corners: a last-ulp difference can move a corner by one unit
    # (0.01 m) when the scaled value sits on an integer; bound it that way
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-2)
    assert np.mean(np.isclose(got, want, rtol=1e-9, atol=1e-15)) > 0.9


@pytest.mark.parametrize("case", range(5))
@pytest.mark.parametrize("mode", MODES)
def test_nms_matches_reference_fixture(case, mode):
    from pointgnn_amd import nms
    fix = np.load(os.path.join(GOLD, "detect_nms.npz"))
    thres, top_k = fix["nms%d_params" % case]
    labels = fix["nms%d_labels" % case]
    boxes = fix["nms%d_boxes" % case]
    scores = fix["nms%d_scores" % case]
    keep_in = (labels.copy(), boxes.copy(), scores.copy())
    lab, box, sco, att = _fn(mode)(
        labels, boxes, scores, overlapped_thres=float(thres),
        overlapped_fn=nms.overlapped_boxes_3d_fast_poly, appr_factor=100.0,
        top_k=int(top_k), attributes=np.arange(len(labels)))
    pre = "nms%d_%s_" % (case, mode)
    assert np.array_equal(att, fix[pre + "attrs"])
    assert np.array_equal(lab, fix[pre + "labels"])
    assert lab.dtype == labels.dtype and att.dtype == np.arange(3).dtype
    assert np.array_equal(box, fix[pre + "boxes"])
    np.testing.assert_allclose(sco, fix[pre + "scores"], rtol=1e-6)
    for a, b in zip(keep_in, (labels, boxes, scores)):
        assert np.array_equal(a, b)          # inputs untouched
    assert np.all(np.diff(scores[att]) < 0)  # score order of the originals


def test_nms_edge_cases():
    import torch
    from pointgnn_amd import nms
    e = nms.nms_boxes_3d_uncertainty(
        np.zeros((0,), np.int64), np.zeros((0, 7), np.float32),
        np.zeros((0,), np.float32), attributes=np.zeros((0,), np.int64))
    assert [len(x) for x in e] == [0, 0, 0, 0] and e[1].shape == (0, 7)
    one = nms.nms_boxes_3d_uncertainty(
        np.array([1]), np.array([[0, 1, 10, 4, 1.5, 1.6, 0.3]], np.float32),
        np.array([0.9], np.float32))
    assert one[0].tolist() == [1] and one[3] is None
    assert one[2][0] == np.float32(0.9)
    # identical boxes collapse to the best-scored one; scores add up with
    # overlap 1 (nms.py:160-163); other classes are left alone
    b = np.tile(np.array([[0, 1, 10, 4, 1.5, 1.6, 0.3]], np.float32), (5, 1))
    lab = np.array([1, 1, 1, 3, 1])
    sc = np.array([0.5, 0.9, 0.4, 0.8, 0.3], np.float32)
    l2, b2, s2, a2 = nms.nms_boxes_3d_uncertainty(
        lab, b, sc, overlapped_thres=0.01, attributes=np.arange(5))
    assert a2.tolist() == [1, 3] and l2.tolist() == [1, 3]
    assert s2[0] == pytest.approx(0.9 + 0.5 + 0.4 + 0.3, rel=1e-6)
    assert s2[1] == np.float32(0.8)
    assert np.array_equal(b2, b[:2])
    # CUDA tensors in -> CUDA tensors out
    t = nms.nms_boxes_3d_merge_only(
        torch.from_numpy(lab).cuda(), torch.from_numpy(b).cuda(),
        torch.from_numpy(sc).cuda(), overlapped_thres=0.01)
    assert t[0].is_cuda and t[1].shape == (2, 7) and t[3] is None
    with pytest.raises(NotImplementedError):
        nms.nms_boxes_3d(lab, b, sc, overlapped_fn=nms.overlapped_boxes_3d)


def test_nms_large_removed_set_and_even_median():
    """One object with more votes than the in-LDS list holds (1024): the
    spill path must give the oracle's medians and score sum."""
    from pointgnn_amd import nms
    rng = np.random.default_rng(2)
    n = 2500                       # 2499 removed + the winner: even-count median
    ctr = np.array([1.0, 1.2, 20.0, 3.9, 1.5, 1.6, 0.4])
    boxes = (ctr + rng.normal(0, 0.03, (n, 7))).astype(np.float32)
    labels = np.ones(n, np.int64)
    scores = rng.permutation(n).astype(np.float32) / n * 0.7 + 0.26
    far = np.array([[40, 1, 50, 4, 1.5, 1.6, 0.0]], np.float32)
    boxes = np.concatenate([boxes, far])
    labels = np.concatenate([labels, [1]])
    scores = np.concatenate([scores, [0.27]]).astype(np.float32)
    for mode in ("uncertainty", "merge_only", "score_only"):
        want = DO.nms_boxes_3d(labels, boxes, scores, 0.01, mode)
        got = _fn(mode)(labels, boxes, scores, overlapped_thres=0.01,
                        attributes=np.arange(len(labels)))
        assert np.array_equal(got[3], want[3]) and len(got[3]) == 2
        assert np.array_equal(got[1], want[1])
        np.testing.assert_allclose(got[2], want[2], rtol=1e-6)


def test_nms_many_boxes_properties():
    """70 000 boxes (keep bits beyond the LDS-resident 65 536): isolated boxes
    all survive in score order; planted duplicates are absorbed."""
    from pointgnn_amd import nms
    rng = np.random.default_rng(3)
    n = 70000
    gx, gz = np.meshgrid(np.arange(280), np.arange(250))
    boxes = np.zeros((n, 7), np.float32)
    boxes[:, 0] = gx.reshape(-1)[:n] * 8.0
    boxes[:, 2] = gz.reshape(-1)[:n] * 8.0
    boxes[:, 1] = 1.0
    boxes[:, 3:6] = (3.9, 1.5, 1.6)
    boxes[:, 6] = rng.uniform(-3, 3, n)
    labels = rng.choice([1, 3], n)
    scores = (rng.permutation(n).astype(np.float32) + 1) / (n + 1)
    dup = rng.choice(n, 500, replace=False)
    boxes2 = np.concatenate([boxes, boxes[dup]])
    labels2 = np.concatenate([labels, labels[dup]])
    scores2 = np.concatenate([scores, scores[dup] * 0.5]).astype(np.float32)
    lab, box, sco, att = nms.nms_boxes_3d_uncertainty(
        labels2, boxes2, scores2, overlapped_thres=0.01,
        attributes=np.arange(n + 500))
    assert len(att) == n and att.max() < n
    order = np.argsort(-scores, kind="stable")
    assert np.array_equal(att, order)
    assert np.array_equal(lab, labels[order])
    assert np.array_equal(box, boxes[order])
    bumped = np.zeros(n, np.float32)
    bumped[dup] = scores[dup] * 0.5
    np.testing.assert_allclose(sco, (scores + bumped)[order], rtol=1e-6)


def test_candidates_and_detect_boxes_pipeline():
    import torch
    from pointgnn_amd import nms
    rng = np.random.default_rng(7)
    for nc, lm in ((4, LABEL_MAPS["car"]), (6, LABEL_MAPS["ped"])):
        k = 1500
        logits = rng.normal(0, 1.5, (k, nc))
        probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
        probs = probs.astype(np.float32)
        probs[:7, 1] = np.float32(1.0 / nc)          # boundary: not selected
        idx, lab = nms.select_candidates(probs)
        widx, wlab = DO.select_candidates(probs)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(lab.cpu().numpy(), wlab)
        # keypoints on a coarse grid so that neighbouring votes overlap
        xyz = np.stack([rng.uniform(-20, 20, k), rng.uniform(0.8, 1.6, k),
                        rng.uniform(5, 45, k)], 1).astype(np.float32)
        enc = rng.normal(0, 0.25, (k, nc, 7)).astype(np.float32)
        got = nms.detect_boxes(probs, enc, xyz, lm, overlapped_thres=0.01)
        # the same stage in the oracle (run.py:264-326)
        dec = DO.box_decoding(np.tile(np.arange(nc), k).reshape(-1, 1),
                              np.repeat(xyz, nc, axis=0),
                              enc.reshape(-1, 1, 7), lm)
        want = DO.nms_boxes_3d(wlab, dec[widx, 0], probs.reshape(-1)[widx],
                               0.01, "uncertainty")
        assert got[0].is_cuda
        assert np.array_equal(got[3].cpu().numpy(), want[3])
        assert np.array_equal(got[0].cpu().numpy(), want[0])
        np.testing.assert_allclose(got[1].cpu().numpy(), want[1], rtol=2e-6,
                                   atol=2e-6)
        np.testing.assert_allclose(got[2].cpu().numpy(), want[2], rtol=1e-5)
        assert 0 < len(want[3]) < len(widx)
    none = nms.detect_boxes(np.full((10, 4), 0.25, np.float32),
                            np.zeros((10, 4, 7), np.float32),
                            np.zeros((10, 3), np.float32), LABEL_MAPS["car"],
                            0.01)
    assert none[1].shape == (0, 7)
