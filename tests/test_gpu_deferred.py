"""Capacity form of the per-frame path: graph build and model with every size
(K, E0, E1) left in device memory (graph_gen `deferred_counts`, the *_dyn
entries of include/pointgnn_hip.h).  The reference's builder never waits for a
size (graph_gen.py:155-220 hands NumPy arrays on); the bar here is that taking
the host reads out changes nothing: keypoints, edge rows (order included),
logits and box encodings are BIT-identical to the host-sized path, whatever
the hints and capacities are."""
import numpy as np
import pytest

import pointgnn_amd  # noqa: F401
from pointgnn_amd import configs, weights
from pointgnn_amd.synthetic import synthetic_cloud

pytestmark = pytest.mark.gpu


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


def _graph_kwargs(cfg):
    return cfg['runtime_graph_gen_kwargs']


@pytest.mark.parametrize("wide", [False, True])
def test_radius_graph_dyn_equals_count_fill(dev, wide):
    """pgnn_radius_graph_dyn on capacity-form points / centres (valid rows
    followed by garbage, counts on the device) == pgnn_radius_graph_count +
    _fill on the exact arrays, row for row."""
    import torch
    from pointgnn_amd import _lib, graph_gen as G
    rng = np.random.default_rng(5)
    dt = np.float64 if wide else np.float32
    pts = (rng.random((6000, 3)) * [40, 3, 40]).astype(dt)
    ctr = pts[rng.permutation(6000)[:900]].copy()
    ctr[::7] += 0.01
    for r, scale in ((1.0, None), (2.5, (1.0, 2.0, 1.0))):
        want, _ = G.radius_graph_device(T(pts, dev), T(ctr, dev), r, scale)
        # capacity form: 30 % more rows of garbage behind the valid ones
        pcap = np.concatenate([pts, rng.random((1800, 3)).astype(dt) * 40])
        ccap = np.concatenate([ctr, rng.random((300, 3)).astype(dt) * 40])
        counts = torch.tensor([6000, 900, 0, 0], dtype=torch.int32, device=dev)
        p = _lib.tag_count(T(pcap, dev), _lib.DeviceCount(counts[0:1], 6000))
        c = _lib.tag_count(T(ccap, dev), _lib.DeviceCount(counts[1:2], 900))
        cap = int(want.shape[0]) + 1000
        got = G.radius_graph_dyn_device(p, c, r, scale, cap, counts[2:4])
        n_written, n_required = counts[2:4].tolist()
        assert n_written == n_required == int(want.shape[0])
        assert torch.equal(got[:n_written], want)
        # too small a buffer: the prefix is written, the required size reported
        small = int(want.shape[0]) // 2
        got = G.radius_graph_dyn_device(p, c, r, scale, small, counts[2:4])
        assert counts[2:4].tolist() == [small, int(want.shape[0])]
        assert torch.equal(got, want[:small])


def test_radius_graph_dyn_empty_and_zero_counts(dev):
    import torch
    from pointgnn_amd import _lib, graph_gen as G
    pts = T(np.random.default_rng(0).random((500, 3)).astype(np.float32), dev)
    counts = torch.tensor([0, 0, 7, 7], dtype=torch.int32, device=dev)
    p = _lib.tag_count(pts, _lib.DeviceCount(counts[0:1], 0))
    c = _lib.tag_count(pts.clone(), _lib.DeviceCount(counts[1:2], 0))
    G.radius_graph_dyn_device(p, c, 0.5, None, 4096, counts[2:4])
    assert counts[2:4].tolist() == [0, 0]
    counts[0] = 500          # points, but no centre
    G.radius_graph_dyn_device(p, c, 0.5, None, 4096, counts[2:4])
    assert counts[2:4].tolist() == [0, 0]
    counts[0], counts[1] = 0, 500   # centres, but no point
    G.radius_graph_dyn_device(p, c, 0.5, None, 4096, counts[2:4])
    assert counts[2:4].tolist() == [0, 0]


@pytest.mark.parametrize("wide", [False, True])
def test_fan_in_cap_in_capacity_form_equals_count_fill(dev, wide):
    """pgnn_radius_graph_dyn + pgnn_radius_graph_dyn_cap (the training-time
    cap, graph_gen.py:210-214, with every size on the device) == the
    host-sized radius graph + pgnn_cap_neighbors_count / _fill for the same
    seed: same surviving rows, same order; an uncapped list that does not fit
    ITS buffer flags the level."""
    import torch
    from pointgnn_amd import _lib, graph_gen as G
    rng = np.random.default_rng(11)
    dt = np.float64 if wide else np.float32
    pts = (rng.random((5000, 3)) * [30, 3, 30]).astype(dt)
    ctr = pts[rng.permutation(5000)[:700]].copy()
    for r, k, seed in ((1.5, 8, 1234), (2.5, 64, 7), (1.0, 100000, 3)):
        want, off = G.radius_graph_device(T(pts, dev), T(ctr, dev), r, None,
                                          k, seed)
        raw, _ = G.radius_graph_device(T(pts, dev), T(ctr, dev), r, None)
        if k < 100:
            assert want.shape[0] < raw.shape[0]       # the cap bites
        fan = torch.diff(off.to(torch.int64))
        assert int(fan.max()) <= k
        pcap = np.concatenate([pts, rng.random((900, 3)).astype(dt) * 30])
        ccap = np.concatenate([ctr, rng.random((200, 3)).astype(dt) * 30])
        counts = torch.tensor([5000, 700, 0, 0, 0, 0], dtype=torch.int32,
                              device=dev)
        p = _lib.tag_count(T(pcap, dev), _lib.DeviceCount(counts[0:1], 5000))
        c = _lib.tag_count(T(ccap, dev), _lib.DeviceCount(counts[1:2], 700))
        cap = int(want.shape[0]) + 500
        raw_cap = int(raw.shape[0]) + 300
        got = G.radius_graph_dyn_device(
            p, c, r, None, cap, counts[2:4],
            fan_in=(k, seed, raw_cap, counts[4:6]))
        assert counts[2:6].tolist() == [want.shape[0], want.shape[0],
                                        raw.shape[0], raw.shape[0]]
        assert torch.equal(got[:want.shape[0]], want)
        # capped list too small: prefix + the size required
        small = int(want.shape[0]) // 2
        got = G.radius_graph_dyn_device(
            p, c, r, None, small, counts[2:4],
            fan_in=(k, seed, raw_cap, counts[4:6]))
        assert counts[2:4].tolist() == [small, want.shape[0]]
        assert torch.equal(got, want[:small])
        # uncapped list too small: level flagged (rows written 0 < required)
        got = G.radius_graph_dyn_device(
            p, c, r, None, cap, counts[2:4],
            fan_in=(k, seed, int(raw.shape[0]) // 2, counts[4:6]))
        assert counts[2:6].tolist() == [0, want.shape[0],
                                        raw.shape[0] // 2, raw.shape[0]]


@pytest.mark.parametrize("method", ["center", "random"])
def test_training_graph_in_capacity_form_equals_host_sized(dev, method):
    """The training graph kwargs (random keypoints on the float64 cloud,
    fan-in cap 256 -- tightened to 24 here so that it bites) through
    deferred_counts == the host-sized call for the same NumPy RNG state; the
    one-read wrapper returns exactly the host-sized lists, also when its
    first frame overflows every capacity."""
    import copy
    import torch
    from pointgnn_amd import _lib, graph_gen as G
    cfg = configs.car_auto_config(3)
    kw = copy.deepcopy(cfg['graph_gen_kwargs'])
    kw['downsample_method'] = method
    kw['add_rnd3d'] = method == 'random'   # (no 'center' form of the jitter)
    for lc in kw['level_configs']:
        lc['graph_gen_kwargs']['num_neighbors'] = 24
    xyz, _ = synthetic_cloud(seed=5, preset="car")
    x = T(xyz.astype(np.float64 if method == "random" else np.float32), dev)
    np.random.seed(42)
    coords, kps, edges = G.gen_multi_level_local_graph_v3(x, **kw)
    after = np.random.get_state()[1][:8].tolist()
    np.random.seed(42)
    _, _, uncapped = G.gen_multi_level_local_graph_v3(
        x, **dict(kw, level_configs=[
            dict(lc, graph_gen_kwargs=dict(lc['graph_gen_kwargs'],
                                           num_neighbors=-1))
            for lc in kw['level_configs']]))
    assert all(a.shape[0] < b.shape[0] for a, b in zip(edges, uncapped))
    k = int(coords[1].shape[0])
    hints = G.CountHints().update(
        k, [int(e.shape[0]) for e in edges], [int(e.shape[0]) for e in uncapped])
    np.random.seed(42)
    c2, k2, e2 = G.gen_multi_level_local_graph_v3(x, deferred_counts=hints, **kw)
    assert np.random.get_state()[1][:8].tolist() == after   # same draws
    frame = _lib.count_of(e2[0]).frame
    assert frame.k == k and not frame.overflowed
    assert frame.edges == [int(e.shape[0]) for e in edges]
    assert frame.raw_edges == [int(e.shape[0]) for e in uncapped]
    for want, got in zip(list(coords) + list(kps) + list(edges),
                         list(c2) + list(k2) + list(e2)):
        assert got.dtype == want.dtype
        assert torch.equal(want, got[:int(want.shape[0])])
    # one read per frame, from cold hints (everything overflows -> rebuilt
    # host-sized from the same random state), then from the learned ones
    cold = G.CountHints()
    for trial in range(2):
        np.random.seed(42)
        c3, k3, e3 = G.gen_multi_level_local_graph_v3_one_read(x, cold, **kw)
        assert np.random.get_state()[1][:8].tolist() == after
        for want, got in zip(list(coords) + list(kps) + list(edges),
                             list(c3) + list(k3) + list(e3)):
            assert got.shape == want.shape and torch.equal(want, got)
        assert all(getattr(e, '_pgnn_sorted', 0) for e in e3)
    assert cold.k == k and cold.raw_caps[0] >= uncapped[0].shape[0]


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("cfg_name,preset", [
    ("car", "small"), ("car", "car"), ("ped", "small")])
def test_deferred_graph_equals_host_sized_graph(dev, cfg_name, preset,
                                                overlap):
    """gen_multi_level_local_graph_v3(deferred_counts=...) == the host-sized
    call: same keypoints in the same order, same edge rows in the same
    order -- also with the independent parts of the build on side streams
    (overlap_build)."""
    import torch
    from pointgnn_amd import _lib, graph_gen as G
    cfg = configs.car_auto_config(3) if cfg_name == "car" else \
        configs.ped_cyl_auto_config(3)
    xyz, _ = synthetic_cloud(seed=3, preset=preset)
    x = T(xyz, dev)
    kw = _graph_kwargs(cfg)
    coords, kps, edges = G.gen_multi_level_local_graph_v3(x, **kw)
    k = int(coords[1].shape[0])
    hints = G.CountHints().update(k, [int(e.shape[0]) for e in edges])
    for trial in range(3):
        if trial == 1:      # hints far off: only speed may change
            hints.k, hints.edges = 7, [11, 13]
        if trial == 2:
            hints.k, hints.edges = 10 * k, [10 ** 8, 10 ** 8]
        c2, k2, e2 = G.gen_multi_level_local_graph_v3(
            x, deferred_counts=hints, overlap_build=overlap, **kw)
        frame = _lib.count_of(e2[0]).frame
        assert frame.k == k and frame.kd_status == 0
        assert frame.edges == [int(e.shape[0]) for e in edges]
        assert not frame.overflowed
        assert int(c2[1].shape[0]) == int(x.shape[0])      # capacity = N
        for a, b in zip(coords, c2):
            n = int(a.shape[0])
            assert torch.equal(a, b[:n])
        for a, b in zip(kps, k2):
            assert torch.equal(a, b[:int(a.shape[0])])
        for a, b in zip(edges, e2):
            assert torch.equal(a, b[:int(a.shape[0])])


@pytest.mark.parametrize("cfg_name,preset,hint_scale", [
    ("car", "car", 1.0), ("car", "car", 0.01), ("car", "car", 50.0),
    ("car", "small", 1.0), ("car", "small", 100.0),
    ("ped", "small", 1.0), ("ped", "car", 1.0), ("ped", "car", 0.01)])
def test_deferred_frame_is_bit_identical(dev, cfg_name, preset, hint_scale,
                                         edge_arith):
    """Graph build + model in capacity form == the host-sized frame, bit for
    bit, also when the hints point at the other kernel of a pair (8-wave vs
    4-wave rows kernel, weights-stationary vs LDS-tile edge / pooling
    kernels): the choice is made on the hint and may not change a result.
    (edge_arith 'bf16x3': under the fixture's `b16_force` the split-bf16 kernel
    runs whatever the hint says; without it a low hint sends the frame to the
    fp32 kernel -- another arithmetic, 1e-6 apart, see gnn.EDGE_ARITHS.)"""
    import torch
    from pointgnn_amd import graph_gen as G
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(3) if cfg_name == "car" else \
        configs.ped_cyl_auto_config(3)
    params = weights.init_params(cfg, seed=4, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev, edge_arith=edge_arith)
    xyz, inten = synthetic_cloud(seed=1, preset=preset)
    x, f = T(xyz, dev), T(inten, dev)
    lg, bx = eng.run_frame(x, f)
    if edge_arith != "f32":
        eng.model.edge_arith = "f32"
        assert not torch.equal(eng.run_frame(x, f)[0], lg), \
            "the split-bf16 kernel did not run"
        eng.model.edge_arith = edge_arith
        eng.frame_shapes.pop()
    k, e0, e1 = eng.frame_shapes[-1]
    eng._hints = G.CountHints(
        max(1, int(k * hint_scale)),
        [max(1, int(e0 * hint_scale)), max(1, int(e1 * hint_scale))],
        [e0 + 4096, e1 + 4096])
    d = eng.run_frame_deferred(x, f)
    assert d.counts._host is None          # nothing was read while enqueuing
    lg2, bx2 = d.result()
    assert eng.deferred_overflows == 0
    assert eng.frame_shapes[-1] == (k, e0, e1)
    assert lg2.shape == lg.shape and bx2.shape == bx.shape
    assert torch.isfinite(lg2).all()
    assert torch.equal(lg, lg2) and torch.equal(bx, bx2)


@pytest.mark.parametrize("cfg_name,preset", [
    ("car", "car"), ("ped", "car"), ("car", "small")])
def test_overlapped_build_frames_are_bit_identical(dev, cfg_name, preset):
    """run_frame_deferred(overlap_build=True): several frames enqueued back to
    back on a non-default stream with nothing waited for in between (the side
    streams, their events and the allocator's blocks are reused from frame to
    frame) give the host-sized frames' logits and boxes, bit for bit; an
    overflowing level is still detected and rebuilt."""
    import torch
    from pointgnn_amd import graph_gen as G
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(3) if cfg_name == "car" else \
        configs.ped_cyl_auto_config(3)
    params = weights.init_params(cfg, seed=4, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    clouds = []
    for seed in (1, 2, 3, 4):
        xyz, inten = synthetic_cloud(seed=seed, preset=preset)
        clouds.append((T(xyz, dev), T(inten, dev)))
    want = [eng.run_frame(x, f) for x, f in clouds]
    shapes = list(eng.frame_shapes[-4:])
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for rep in range(2):
            frames = [eng.run_frame_deferred(x, f, overlap_build=True)
                      for x, f in clouds]
            assert all(d.counts._host is None for d in frames)
            got = [d.result() for d in frames]
            for (lg, bx), (lg2, bx2) in zip(want, got):
                assert torch.equal(lg, lg2) and torch.equal(bx, bx2)
        assert eng.deferred_overflows == 0
        k, e0, e1 = shapes[0]
        eng._hints = G.CountHints(k, [e0, e1], [e0 // 2, e1 + 10])
        d = eng.run_frame_deferred(*clouds[0], overlap_build=True)
        assert d.counts.overflowed == [0]
        lg2, bx2 = d.result()
        assert eng.deferred_overflows == 1
        assert torch.equal(want[0][0], lg2) and torch.equal(want[0][1], bx2)
    torch.cuda.synchronize()


def test_radius_graph_stages_on_two_streams(dev):
    """pgnn_radius_graph_dyn_grid on one stream while the centres are still
    being produced on another, _query after both: the rows of
    pgnn_radius_graph_dyn."""
    import torch
    from pointgnn_amd import graph_gen as G
    rng = np.random.RandomState(5)
    pts = T((rng.rand(6000, 3) * 20).astype(np.float32), dev)
    src = T((rng.rand(900, 3) * 20).astype(np.float32), dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    want = G.radius_graph_dyn_device(pts, src * 1.0, 1.5, None, 200000,
                                     counts[0:2])
    n_want = int(counts[0].item())
    assert 0 < n_want < 200000
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    job = G._RadiusDynJob(pts, None, 900, 1.5, None, 200000, counts[2:4])
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        job.grid()
        ev.record()
    ctr = src * 1.0                      # produced on the main stream
    torch.cuda.current_stream().wait_event(ev)
    got = job.query(ctr, None)
    torch.cuda.synchronize()
    assert counts.tolist() == [n_want, n_want, n_want, n_want]
    assert torch.equal(want[:n_want], got[:n_want])


def test_deferred_overflow_is_detected_and_rebuilt(dev):
    """An edge list larger than its capacity: the frame reports it (required
    > written) and the engine rebuilds it with host-read sizes; the next
    frame's capacity covers it."""
    import torch
    from pointgnn_amd import graph_gen as G
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(2)
    params = weights.init_params(cfg, seed=6, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    xyz, inten = synthetic_cloud(seed=2, preset="car")
    x, f = T(xyz, dev), T(inten, dev)
    lg, bx = eng.run_frame(x, f)
    k, e0, e1 = eng.frame_shapes[-1]
    eng._hints = G.CountHints(k, [e0, e1], [e0 // 2, e1 + 10])
    d = eng.run_frame_deferred(x, f)
    assert d.counts.overflowed == [0]
    lg2, bx2 = d.result()
    assert eng.deferred_overflows == 1
    assert torch.equal(lg, lg2) and torch.equal(bx, bx2)
    assert eng._hints.cap(0) >= e0 and eng._hints.cap(1) >= e1
    d = eng.run_frame_deferred(x, f)
    lg3, bx3 = d.result()
    assert eng.deferred_overflows == 1 and not d.counts.overflowed
    assert torch.equal(lg, lg3) and torch.equal(bx, bx3)


def test_deferred_pipeline_equals_sequential(dev):
    """run_frames_pipelined(deferred=True): frames of different sizes through
    the multi-stream schedule without a host wait == frame-at-a-time."""
    import torch
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(2)
    params = weights.init_params(cfg, seed=9, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    frames = []
    for s in range(6):
        xyz, inten = synthetic_cloud(
            seed=s, preset=("small", "tiny", "car")[s % 3])
        frames.append((T(xyz, dev), T(inten, dev)))
    seq = [eng.run_frame(x, f) for x, f in frames]
    shapes = list(eng.frame_shapes)
    torch.cuda.synchronize()
    for rep in range(6):
        eng.frame_shapes = []
        # reps 2..5: the graphs of consecutive frames on two / three builder
        # streams, each build running that many frames ahead of its GNN
        pip = eng.run_frames_pipelined(frames, compute_streams=1 + rep % 2,
                                       deferred=True,
                                       graph_streams=1 + rep // 2)
        torch.cuda.synchronize()
        assert eng.frame_shapes == shapes[-len(frames):]
        for (l0, b0), (l1, b1) in zip(seq, pip):
            assert torch.equal(l0, l1) and torch.equal(b0, b1)


def test_count_record_pool_wraps_inside_a_deferred_batch(dev, monkeypatch):
    """The pool of zeroed count records rotates its segments while frames of
    the batch are still unread (a bulk re-zero handed K = 0 / E = 0
    to every frame built before the wrap).  Segments of 3 records, batches of
    7 frames: every batch crosses two rotations, and the lookahead builders
    run ahead of the GNN.  A live record is never wiped; idle segments are
    the ones reused."""
    import torch
    from pointgnn_amd import graph_gen as G
    from pointgnn_amd.engine import InferenceEngine
    monkeypatch.setattr(G._ZeroPool, "kRecords", 3)
    monkeypatch.setattr(G, "_ZERO_POOLS", {})
    cfg = configs.car_auto_config(1)
    params = weights.init_params(cfg, seed=5, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    frames = []
    for s in range(7):
        xyz, inten = synthetic_cloud(seed=s, preset=("small", "tiny")[s % 2])
        frames.append((T(xyz, dev), T(inten, dev)))
    seq = [eng.run_frame(x, f) for x, f in frames]
    torch.cuda.synchronize()
    for rep in range(3):
        got = eng.run_frames_on_streams(frames, 3)
        pip = eng.run_frames_pipelined(frames, compute_streams=2,
                                       deferred=True, graph_streams=2)
        torch.cuda.synchronize()
        for (l0, b0), (l1, b1), (l2, b2) in zip(seq, got, pip):
            assert l0.shape[0] > 0
            assert torch.equal(l0, l1) and torch.equal(b0, b1)
            assert torch.equal(l0, l2) and torch.equal(b0, b2)
    dev = G._device()
    pool = G._ZERO_POOLS[dev.index]
    assert pool.kRecords == 3
    # rotations happened, and dead segments were recycled rather than leaked
    assert len(pool.retired) <= 8
    # held records survive any number of later rotations
    held = [G._zero_counts(6, dev) for _ in range(4)]
    for h in held:
        h[0] = 77
    for _ in range(20):
        G._zero_counts(6, dev)
    torch.cuda.synchronize()
    assert all(int(h[0]) == 77 for h in held)
    fresh = G._zero_counts(6, dev)
    assert fresh.tolist() == [0] * 6


def test_frames_on_streams_equal_sequential(dev, edge_arith):
    """run_frames_on_streams: whole frames (graph build + GNN, capacity form)
    round-robin on 1..4 streams == frame-at-a-time execution, bit for bit,
    for frames of different sizes and for both shipped inference configs, on
    both arithmetics of the edge stage."""
    import torch
    from pointgnn_amd.engine import InferenceEngine
    for cfg in (configs.car_auto_config(2), configs.ped_cyl_auto_config(1)):
        params = weights.init_params(cfg, seed=11, bias_scale=0.05)
        eng = InferenceEngine(cfg, params, device=dev, edge_arith=edge_arith)
        frames = []
        for s in range(7):
            xyz, inten = synthetic_cloud(
                seed=s, preset=("small", "car", "tiny")[s % 3])
            frames.append((T(xyz, dev), T(inten, dev)))
        seq = [eng.run_frame(x, f) for x, f in frames]
        shapes = list(eng.frame_shapes)
        torch.cuda.synchronize()
        for n in (1, 2, 3, 4):
            eng.frame_shapes = []
            got = eng.run_frames_on_streams(frames, n)
            assert eng.frame_shapes == shapes
            for (l0, b0), (l1, b1) in zip(seq, got):
                assert torch.equal(l0, l1) and torch.equal(b0, b1)
        assert eng.deferred_overflows == 0


@pytest.mark.parametrize("n_points", [0, 1, 2, 37])
def test_deferred_frame_on_degenerate_clouds(dev, n_points):
    """Empty, single-point and tiny clouds through the capacity form: the same
    (possibly empty) outputs as the host-sized frame, no overflow, no
    out-of-range access (capacities a few rows above the need)."""
    import torch
    from pointgnn_amd import graph_gen as G
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(1)
    params = weights.init_params(cfg, seed=3, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev)
    rng = np.random.default_rng(n_points)
    xyz = (rng.random((n_points, 3)) * [3.0, 1.0, 3.0]).astype(np.float32)
    inten = rng.random((n_points, 1)).astype(np.float32)
    x, f = T(xyz, dev), T(inten, dev)
    lg, bx = eng.run_frame(x, f)
    shape = eng.frame_shapes[-1]
    eng._hints = G.CountHints(shape[0], shape[1:],
                              [shape[1] + 8, shape[2] + 8])
    d = eng.run_frame_deferred(x, f)
    assert (d.counts is None) == (n_points == 0)   # an empty cloud is host-known
    lg2, bx2 = d.result()
    torch.cuda.synchronize()
    assert eng.deferred_overflows == 0 and eng.frame_shapes[-1] == shape
    assert lg2.shape == lg.shape and bx2.shape == bx.shape
    assert torch.equal(lg, lg2) and torch.equal(bx, bx2)


@pytest.mark.parametrize("cfg_name,overlap", [
    ("car", False), ("ped", False), ("car", True)])
def test_captured_frame_replays_bit_identically(dev, cfg_name, overlap,
                                                edge_arith):
    """A whole frame in ONE hipGraph (engine.capture_frame; `overlap`: with the
    graph build's side streams as branches of the graph): replays give the
    eager results bit for bit, for the captured cloud and for other clouds of
    the same point count (K and the edge counts differ: they live on the
    device), replay after replay (ped_cyl: the LDS-tile pooling kernel's tile
    pool counters are re-armed by a kernel -- a hipMemsetAsync node did not
    take effect on later replays); another point count is refused."""
    import torch
    from pointgnn_amd.engine import InferenceEngine
    cfg = configs.car_auto_config(3) if cfg_name == "car" else \
        configs.ped_cyl_auto_config(3)
    params = weights.init_params(cfg, seed=5, bias_scale=0.05)
    eng = InferenceEngine(cfg, params, device=dev, edge_arith=edge_arith)
    clouds = []
    for s in (0, 1, 2):
        xyz, inten = synthetic_cloud(seed=s, preset="car")
        clouds.append((T(xyz, dev), T(inten, dev)))
    eager = [eng.run_frame(x, f) for x, f in clouds]
    shapes = list(eng.frame_shapes)
    assert len({sh[0] for sh in shapes}) > 1       # different K per cloud
    cap = eng.capture_frame(*clouds[0], overlap_build=overlap)
    for rep in range(3):
        for (x, f), (lg, bx), sh in zip(clouds, eager, shapes):
            out = cap.replay(x, f)
            lg2, bx2 = out.result()
            assert eng.frame_shapes[-1] == sh
            assert torch.equal(lg, lg2) and torch.equal(bx, bx2)
    eng.check_edge_range()      # (f16x2: the range flag survives replays)
    small = synthetic_cloud(seed=0, preset="small")
    with pytest.raises(ValueError):
        cap.replay(T(small[0], dev), T(small[1], dev))


def test_deferred_reports_kd_status(dev):
    """The kd-tree replica's tie-order status travels with the counts: a
    cloud outside what the replica reproduces raises when the frame's result
    is taken, like the host-sized path does when it reads K."""
    from pointgnn_amd import _lib, graph_gen as G
    c = G.FrameCounts(None, [10, 10])
    c._host = [5, 1, 3, 3, 4, 4]
    assert c.kd_status == 1 and c.k == 5 and c.edges == [3, 4]
    with pytest.raises(_lib.PointGnnHipError):
        G.check_kd_status(c.kd_status)
