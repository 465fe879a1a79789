"""Graph construction -- the reference's `models/graph_gen.py` operator surface on
the MI355X kernels of csrc/graph.hip.

Same function names, keyword arguments and registry keys as the reference
(graph_gen.py:155-157, 197-200, 222-227), so `run.py`'s
`get_graph_generate_fn(config['graph_gen_method'])(xyz, **config[
'runtime_graph_gen_kwargs'])` call (run.py:219-222) works unchanged.

Array contract: NumPy in -> NumPy out (host round trip, drop-in); torch CUDA
tensors in -> torch CUDA tensors out (device-resident fast path used by
bench.py and the model).  Coordinates keep their precision like in the
reference: a float64 cloud (what train.py:88-90 passes after the
augmentations) is voxelised and searched as float64 and its vertex lists come
back float64; anything else is the float32 cloud run.py:219-222 feeds.  Edges are int32 [E,2] rows (point_idx, centre_idx)
grouped by ascending centre -- the reference emits the same grouping
(graph_gen.py:215-219) with an unspecified order inside a centre.  Keypoint
order is ascending voxel-hash bucket (the reference's is open3d's hash-map
order / dict order: only the *set* is defined).

Device tensors are sized on the host by default (two reads per frame: K, then
the edge totals).  `gen_multi_level_local_graph_v3(..., deferred_counts=hints)`
is the capacity form: no read at all, capacity-sized outputs tagged with their
device-side counts (CountHints, FrameCounts, _lib.DeviceCount).
"""
import ctypes
import threading
import weakref

import numpy as np
import torch

from . import _lib

__all__ = ["gen_disjointed_rnn_local_graph_v3",
           "gen_multi_level_local_graph_v3", "get_graph_generate_fn",
           "multi_layer_downsampling_select", "multi_layer_downsampling_random",
           "gen_multi_level_local_graph_v3_one_read",
           "CountHints", "FrameCounts"]


class CountHints(object):
    """What the host expects a frame's sizes to be, for the capacity form
    (`gen_multi_level_local_graph_v3(..., deferred_counts=hints)`): `k` the
    keypoint count, `edges[l]` the edge count of level l -- both only steer
    kernel choice and grid sizes -- and `edge_caps[l]` the rows allocated for
    level l's edge list (a frame that needs more reports it, see FrameCounts).
    `update(k, edges)` folds a finished frame's sizes in: hints follow the
    last frame, capacities keep 2 x the largest list seen (8 bytes a row).
    Levels with a fan-in cap (training kwargs) also keep `raw_caps[l]`, the
    rows allocated for the list BEFORE the cap."""

    MIN_EDGE_CAP = 1 << 16

    def __init__(self, k=0, edges=(), edge_caps=(), raw_caps=()):
        self.k = int(k)
        self.edges = [int(e) for e in edges]
        self.edge_caps = [int(c) for c in edge_caps]
        self.raw_caps = [int(c) for c in raw_caps]

    def _grown(self, caps, sizes):
        caps = list(caps) + [0] * (len(sizes) - len(caps))
        return [max(c, self.MIN_EDGE_CAP, 2 * int(e) + 1024)
                for c, e in zip(caps, sizes)]

    def update(self, k, edges, raw_edges=None):
        self.k = int(k)
        self.edges = [int(e) for e in edges]
        self.edge_caps = self._grown(self.edge_caps, edges)
        if raw_edges is not None:
            self.raw_caps = self._grown(self.raw_caps, raw_edges)
        return self

    def raw_cap(self, level):
        """rows for level's uncapped list: what was seen, else 2 x the capped
        capacity (most centres of a training graph sit below the cap)"""
        if level < len(self.raw_caps) and self.raw_caps[level] > 0:
            return self.raw_caps[level]
        return 2 * self.cap(level)

    def cap(self, level):
        if level < len(self.edge_caps) and self.edge_caps[level] > 0:
            return self.edge_caps[level]
        return self.MIN_EDGE_CAP

    def edge_hint(self, level):
        return self.edges[level] if level < len(self.edges) else 0


class FrameCounts(object):
    """The sizes of one capacity-form frame, in device memory until `read()`:
    tensor = int32 [2 + 2 L]: K, kd-tree tie-order status, then per level
    (rows written, rows required); with a fan-in cap on any level L more pairs
    follow, the same record for every level's list BEFORE the cap (zeros for
    levels without one).  `read()` is the frame's ONE host read (it
    waits for the stream that built the graph); the caller does it when it
    takes the frame's results, not before the model runs."""

    def __init__(self, tensor, edge_caps):
        self.tensor = tensor
        self.edge_caps = list(edge_caps)
        self._host = None

    def read(self):
        if self._host is None:
            self._host = [int(v) for v in self.tensor.tolist()]
        return self._host

    @property
    def k(self):
        return self.read()[0]

    @property
    def kd_status(self):
        return self.read()[1]

    @property
    def edges(self):
        """Edge rows each level needs (== rows written unless overflowed)."""
        v = self.read()
        return [v[3 + 2 * l] for l in range(len(self.edge_caps))]

    @property
    def raw_edges(self):
        """Rows each level's uncapped list needs (None without any cap)."""
        v = self.read()
        n = len(self.edge_caps)
        if len(v) < 2 + 4 * n:
            return None
        return [v[3 + 2 * n + 2 * l] for l in range(n)]

    @property
    def overflowed(self):
        """Levels whose edge list did not fit its capacity (their tail was
        dropped: the frame has to be rebuilt with a larger capacity)."""
        v = self.read()
        return [l for l in range(len(self.edge_caps))
                if v[3 + 2 * l] > v[2 + 2 * l]]


class _ZeroPool(object):
    """Zeroed int32 count records for capacity-form frames, handed out from
    segments of `kRecords` records that are zeroed ONCE per segment (a fill
    launch per frame otherwise: ~4 us of a 2-3 ms frame, and one more node on
    every frame's dependency chain).

    A record stays the frame's own for as long as the tensor `take()`
    returned is alive (FrameCounts holds it, and every graph tensor of the
    frame holds the FrameCounts): a segment is zeroed again only when all the
    records it served are dead AND the device has been synchronised since, so
    neither an unread FrameCounts nor a kernel that has yet to be enqueued
    (a lookahead builder runs frames ahead of the GNN) can meet a wiped
    record.  A segment with live records is left alone and a fresh one is
    allocated instead.  Thread-safe: builder threads share the pool."""
    kRecords, kInts = 512, 16

    class _Segment(object):
        def __init__(self, dev, n_records, n_ints):
            self.buf = torch.zeros(n_records * n_ints, dtype=torch.int32,
                                   device=dev)
            self.owners = []

        def idle(self):
            return all(w() is None for w in self.owners)

    def __init__(self, dev):
        self.dev = dev
        self.lock = threading.Lock()
        self.cur = self._Segment(dev, self.kRecords, self.kInts)
        torch.cuda.synchronize(dev)   # records are used on any stream
        self.retired = []
        self.next = 0

    def _rotate(self):
        # everything enqueued so far has run once this returns: a retired
        # segment whose records are all dead has no reader left anywhere
        torch.cuda.synchronize(self.dev)
        self.retired.append(self.cur)
        seg = None
        for i, s in enumerate(self.retired):
            if s.idle():
                seg = self.retired.pop(i)
                seg.owners = []
                seg.buf.zero_()
                break
        if seg is None:
            seg = self._Segment(self.dev, self.kRecords, self.kInts)
        torch.cuda.synchronize(self.dev)
        self.cur, self.next = seg, 0

    def take(self, n_ints):
        if n_ints > self.kInts:
            return torch.zeros(n_ints, dtype=torch.int32, device=self.dev)
        with self.lock:
            if self.next == self.kRecords:
                self._rotate()
            i = self.next
            self.next += 1
            rec = self.cur.buf[i * self.kInts:i * self.kInts + n_ints]
            self.cur.owners.append(weakref.ref(rec))
            return rec


_ZERO_POOLS = {}
_ZERO_POOLS_LOCK = threading.Lock()


def _zero_counts(n_ints, dev):
    # (inside a stream capture the record has to be zeroed by the captured
    # work itself: every replay reuses it)
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(n_ints, dtype=torch.int32, device=dev)
    with _ZERO_POOLS_LOCK:
        pool = _ZERO_POOLS.get(dev.index)
        if pool is None:
            pool = _ZERO_POOLS[dev.index] = _ZeroPool(dev)
    return pool.take(n_ints)


def _device():
    if not torch.cuda.is_available():
        raise _lib.PointGnnHipError(
            "pointgnn_amd.graph_gen needs a GPU: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a):
    """-> (contiguous CUDA tensor [n,3], was_numpy): float64 stays float64
    (the reference's arithmetic on such a cloud is float64 throughout),
    every other dtype becomes float32."""
    if isinstance(a, torch.Tensor):
        dt = torch.float64 if a.dtype == torch.float64 else torch.float32
        return a.to(device=_device(), dtype=dt).contiguous(), False
    arr = np.asarray(a)
    dt = np.float64 if arr.dtype == np.float64 else np.float32
    arr = np.ascontiguousarray(arr, dtype=dt)
    return torch.from_numpy(arr).to(_device()), True


def _to_dev_f32(a):
    """-> (contiguous float32 CUDA tensor [n,3], was_numpy)"""
    t, was_np = _to_dev(a)
    return t.to(torch.float32), was_np


def _same_precision(points, centers):
    """Both operands of a radius query in one dtype: float64 as soon as one of
    them is (widening float32 is exact, so this is what NumPy's promotion in
    sklearn's float64 tree does), float32 otherwise."""
    if points.dtype == torch.float64 or centers.dtype == torch.float64:
        return (points.to(torch.float64).contiguous(),
                centers.to(torch.float64).contiguous(), True)
    return points, centers, False


# what to do when the kd-tree replica reports that libstdc++'s heap-select
# fallback would have run (pgnn_kdtree_replica `status`): exact 1-NN ties of
# that frame may then be broken differently from the reference's sklearn call
KD_STATUS_POLICY = 'raise'   # or 'warn'


def _scale3(scale):
    if scale is None:
        return None, ctypes.c_void_p(0)
    s = np.broadcast_to(np.asarray(scale, dtype=np.float64).reshape(-1), (3,))
    s = np.ascontiguousarray(s)
    return s, ctypes.c_void_p(s.ctypes.data)


def radius_graph_device(points, centers, radius, scale=None, num_neighbors=-1,
                        seed=0):
    """Device tensors in/out.  Returns (edges int32 [E,2], offsets int32
    [Q+1]).  One host sync (reading E) per call, two when capping."""
    lib = _lib.load()
    dev = points.device
    points, centers, wide = _same_precision(points, centers)
    count_fn = lib.pgnn_radius_graph_count_f64 if wide else \
        lib.pgnn_radius_graph_count
    fill_fn = lib.pgnn_radius_graph_fill_f64 if wide else \
        lib.pgnn_radius_graph_fill
    n_p, n_c = int(points.shape[0]), int(centers.shape[0])
    ws_bytes = lib.pgnn_radius_graph_workspace_bytes(n_p, n_c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    offsets = torch.empty(n_c + 1, dtype=torch.int32, device=dev)
    keep, sp = _scale3(scale)
    st = _lib.stream_ptr()
    _lib.check(count_fn(
        _lib.ptr(points), n_p, _lib.ptr(centers), n_c, float(radius), sp,
        _lib.ptr(ws), ws_bytes, _lib.ptr(offsets), st),
        "pgnn_radius_graph_count")
    n_e = int(offsets[-1].item())  # the one host sync
    edges = torch.empty((n_e, 2), dtype=torch.int32, device=dev)
    _lib.check(fill_fn(
        _lib.ptr(points), n_p, _lib.ptr(centers), n_c, float(radius), sp,
        _lib.ptr(ws), ws_bytes, _lib.ptr(offsets), _lib.ptr(edges), n_e, st),
        "pgnn_radius_graph_fill")
    del keep
    if num_neighbors is not None and num_neighbors > 0:
        new_off = torch.empty(n_c + 1, dtype=torch.int32, device=dev)
        _lib.check(lib.pgnn_cap_neighbors_count(
            _lib.ptr(offsets), n_c, int(num_neighbors), _lib.ptr(new_off), st),
            "pgnn_cap_neighbors_count")
        n_e2 = int(new_off[-1].item())
        if n_e2 != n_e:
            new_edges = torch.empty((n_e2, 2), dtype=torch.int32, device=dev)
            _lib.check(lib.pgnn_cap_neighbors_fill(
                _lib.ptr(offsets), _lib.ptr(edges), n_c, int(num_neighbors),
                int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(new_off),
                _lib.ptr(new_edges), n_e2, st), "pgnn_cap_neighbors_fill")
            edges, offsets = new_edges, new_off
    edges._pgnn_sorted = 1  # grouped by ascending centre (see gnn.mark_sorted)
    return edges, offsets


def radius_graphs_device(queries):
    """Several independent radius graphs with ONE wait between them: every
    count pass is enqueued first, then the edge totals are read (the first read
    waits for all the count passes, the rest return at once), then every fill
    pass.  queries: list of (points, centers, radius,
    scale); returns the list of edge tensors (no fan-in cap).  Used by
    gen_multi_level_local_graph_v3 for the levels of a frame (two host reads
    per frame -- K, then (E0, E1) -- instead of three)."""
    lib = _lib.load()
    st = _lib.stream_ptr()
    pend = []
    for points, centers, radius, scale in queries:
        dev = points.device
        points, centers, wide = _same_precision(points, centers)
        n_p, n_c = int(points.shape[0]), int(centers.shape[0])
        ws_bytes = lib.pgnn_radius_graph_workspace_bytes(n_p, n_c)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        offsets = torch.empty(n_c + 1, dtype=torch.int32, device=dev)
        keep, sp = _scale3(scale)
        _lib.check((lib.pgnn_radius_graph_count_f64 if wide else
                    lib.pgnn_radius_graph_count)(
            _lib.ptr(points), n_p, _lib.ptr(centers), n_c, float(radius), sp,
            _lib.ptr(ws), ws_bytes, _lib.ptr(offsets), st),
            "pgnn_radius_graph_count")
        pend.append((points, centers, radius, keep, sp, ws, ws_bytes, offsets,
                     wide))
    # the host read: the first one waits for every count pass enqueued above,
    # the others find their value already computed
    totals = [int(q[7][-1].item()) for q in pend]
    out = []
    for (points, centers, radius, keep, sp, ws, ws_bytes, offsets,
         wide), n_e in zip(pend, totals):
        n_p, n_c = int(points.shape[0]), int(centers.shape[0])
        edges = torch.empty((int(n_e), 2), dtype=torch.int32,
                            device=points.device)
        _lib.check((lib.pgnn_radius_graph_fill_f64 if wide else
                    lib.pgnn_radius_graph_fill)(
            _lib.ptr(points), n_p, _lib.ptr(centers), n_c, float(radius), sp,
            _lib.ptr(ws), ws_bytes, _lib.ptr(offsets), _lib.ptr(edges),
            int(n_e), st), "pgnn_radius_graph_fill")
        edges._pgnn_sorted = 1
        out.append(edges)
    return out


def radius_graph_dyn_device(points, centers, radius, scale, edge_cap,
                            n_edges_out, edge_hint=0, fan_in=None):
    """Capacity form (pgnn_radius_graph_dyn): no host read.  `points` /
    `centers` may be capacity-form tensors (tagged with a DeviceCount);
    `n_edges_out` is the int32 [2] device slice that receives (rows written,
    rows required).  `fan_in` = (num_neighbors, seed, raw_cap, n_raw_out): the
    training-time cap (graph_gen.py:210-214), also without a read.  Returns
    the [edge_cap, 2] edge tensor tagged with its count."""
    cp, cc = _lib.count_of(points), _lib.count_of(centers)
    points, centers, _ = _same_precision(points, centers)
    job = _RadiusDynJob(points, cp, int(centers.shape[0]), radius, scale,
                        edge_cap, n_edges_out, edge_hint, fan_in)
    job.grid()
    return job.query(centers, cc)


class _RadiusDynJob(object):
    """One capacity-form radius graph in its two stages
    (pgnn_radius_graph_dyn_grid / _query).  Every buffer is allocated here, on
    the stream current at construction; grid() and query() enqueue on the
    stream current at THEIR call, so a caller can put the grid stage -- which
    needs the points only -- on a side stream while the centres are still
    being computed.  `points` and the later `centers` must have one dtype.
    With `fan_in` = (num_neighbors, seed, raw_cap, n_raw_out) the query stage
    writes the uncapped list into its own [raw_cap, 2] buffer (record ->
    n_raw_out) and pgnn_radius_graph_dyn_cap selects from it into `edges`."""

    def __init__(self, points, points_count, centers_cap, radius, scale,
                 edge_cap, n_edges_out, edge_hint=0, fan_in=None):
        self.lib = _lib.load()
        self.points, self.cp = points, points_count
        self.wide = points.dtype == torch.float64
        self.n_p = int(points.shape[0])
        self.n_c = int(centers_cap)
        self.radius = float(radius)
        self.keep, self.sp = _scale3(scale)
        dev = points.device
        self.ws_bytes = self.lib.pgnn_radius_graph_dyn_workspace_bytes(
            self.n_p, self.n_c)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.edge_cap = int(edge_cap)
        self.edges = torch.empty((self.edge_cap, 2), dtype=torch.int32,
                                 device=dev)
        self.n_edges_out = n_edges_out
        self.edge_hint = edge_hint
        self.fan_in = None
        if fan_in is not None and int(fan_in[0]) > 0:
            k, seed, raw_cap, n_raw_out = fan_in
            self.fan_in = (int(k), int(seed) & 0xFFFFFFFFFFFFFFFF)
            self.raw_cap = int(raw_cap)
            self.raw = torch.empty((self.raw_cap, 2), dtype=torch.int32,
                                   device=dev)
            self.n_raw_out = n_raw_out
            self.new_off = torch.empty(self.n_c + 1, dtype=torch.int32,
                                       device=dev)

    def grid(self):
        _lib.check((self.lib.pgnn_radius_graph_dyn_grid_f64 if self.wide else
                    self.lib.pgnn_radius_graph_dyn_grid)(
            _lib.ptr(self.points), self.n_p,
            _lib.ptr(self.cp.dev if self.cp else None), self.n_c, self.radius,
            self.sp, _lib.ptr(self.ws), self.ws_bytes, _lib.stream_ptr()),
            "pgnn_radius_graph_dyn_grid")

    def query(self, centers, centers_count):
        if centers.dtype != self.points.dtype or \
                int(centers.shape[0]) != self.n_c:
            raise ValueError("radius graph: centres do not match the job")
        _lib.check((self.lib.pgnn_radius_graph_dyn_query_f64 if self.wide else
                    self.lib.pgnn_radius_graph_dyn_query)(
            _lib.ptr(self.points), self.n_p, _lib.ptr(centers), self.n_c,
            _lib.ptr(centers_count.dev if centers_count else None),
            self.radius, self.sp, _lib.ptr(self.ws), self.ws_bytes,
            _lib.ptr(self.raw if self.fan_in else self.edges),
            self.raw_cap if self.fan_in else self.edge_cap,
            _lib.ptr(self.n_raw_out if self.fan_in else self.n_edges_out),
            _lib.stream_ptr()),
            "pgnn_radius_graph_dyn_query")
        if self.fan_in:
            _lib.check(self.lib.pgnn_radius_graph_dyn_cap(
                _lib.ptr(self.ws), self.ws_bytes, self.n_p, self.n_c,
                _lib.ptr(self.raw), self.raw_cap, _lib.ptr(self.n_raw_out),
                self.fan_in[0], self.fan_in[1], _lib.ptr(self.new_off),
                _lib.ptr(self.edges), self.edge_cap,
                _lib.ptr(self.n_edges_out), _lib.stream_ptr()),
                "pgnn_radius_graph_dyn_cap")
        self.edges._pgnn_sorted = 1
        return _lib.tag_count(
            self.edges,
            _lib.DeviceCount(self.n_edges_out[0:1], self.edge_hint))


def gen_disjointed_rnn_local_graph_v3(
        points_xyz, center_xyz, radius, num_neighbors,
        neighbors_downsample_method='random', scale=None, seed=None):
    """graph_gen.py:197-220.  `seed` (extension) keys the random fan-in cap;
    default: drawn from numpy's global RNG like the reference's choice."""
    if num_neighbors > 0 and neighbors_downsample_method != 'random':
        # the reference silently skips the cap for any other method name
        num_neighbors = -1
    p, was_np = _to_dev(points_xyz)
    c, _ = _to_dev(center_xyz)
    if seed is None and num_neighbors > 0:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    edges, _ = radius_graph_device(p, c, radius, scale, num_neighbors,
                                   seed or 0)
    if was_np:
        return edges.cpu().numpy()
    return edges


_AUX_STREAMS = {}
_IDENTITY = {}


def _identity_indices(n, dev):
    """[n,1] int32 0..n-1 for the levels that keep their vertices
    (graph_gen.py:76-81): a view of one cached ramp per device, so the frame
    loop launches no torch kernel for it.  Read-only by contract."""
    ramp = _IDENTITY.get(dev.index)
    if ramp is None or ramp.shape[0] < n:
        size = max(int(n), 1 << 16, 2 * (ramp.shape[0] if ramp is not None else 0))
        ramp = torch.arange(size, dtype=torch.int32, device=dev)
        # other streams read views of this ramp without an event: make sure
        # the fill has run before anyone can (a rare, one-off wait)
        torch.cuda.current_stream(dev).synchronize()
        _IDENTITY[dev.index] = ramp
    return ramp[:n].reshape(-1, 1)


_OVERLAP = {}


class _OverlapSet(object):
    """The side stream and the four events of an overlapped graph build, per
    (device, main stream): made once, re-recorded every frame (creating and
    destroying events per frame is not free in the HIP runtime)."""

    def __init__(self, dev):
        # streams on hardware queues of their own: two streams that HIP put
        # on one queue run strictly in turn (engine.concurrent_streams probes
        # for that once per main stream)
        from .engine import concurrent_streams
        self.side, self.aux = concurrent_streams(2, dev)
        self.fork, self.grid, self.kp, self.join = (
            torch.cuda.Event() for _ in range(4))


def _overlap_set(dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    o = _OVERLAP.get(key)
    if o is None:
        o = _OVERLAP[key] = _OverlapSet(dev)
    return o


def _aux_stream(dev):
    """A side stream per (device, current stream): the kd-tree replica of the
    'center' keypoints is forked onto it (pgnn_voxel_keypoints_center)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    s = _AUX_STREAMS.get(key)
    if s is None:
        s = _AUX_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def keypoints_device(points, voxel_size, method='center', jitter=None, seed=0,
                     fork_kdtree=False, num_out=None, k_hint=0, base=None):
    """One pooling level.  Returns (coords [K,3] in the dtype of `points`,
    indices int32 [K,1]) as device tensors.  One host sync (reading K and the
    tie-order status together) -- unless `num_out` (an int32 [2] device
    tensor) is given: then K and the status stay there, nothing is read, and
    the two arrays come back in capacity form ([n,3] / [n,1], tagged with the
    DeviceCount of K).

    `base` (the second and later pooling levels, graph_gen.py:49-90, :92-153):
    the previous level's vertex coordinates, `points` then being the ORIGINAL
    cloud -- 'center': centroids of the cloud's voxels, nearest neighbour among
    `base`; 'random': `base` is voxelised on the grid anchored at the cloud's
    minimum.  The indices refer to `base`."""
    if base is not None:
        return _keypoints_from(points, base, voxel_size, method, jitter, seed)
    lib = _lib.load()
    dev = points.device
    n = int(points.shape[0])
    wide = points.dtype == torch.float64
    if wide and method == 'center':
        # the kd-tree replica keys are float32; a float64 cloud whose values
        # are float32-representable (a widened float32 cloud) is the same
        # computation, anything else has no device path (no shipped config:
        # training uses 'random', run.py feeds the float32 cloud)
        if num_out is not None:
            # capacity form: the representability check below is a blocking
            # host read, which that form promises not to do (and which would
            # abort a hipGraph capture)
            raise NotImplementedError(
                "downsample_method='center' on a float64 cloud in capacity "
                "form (deferred_counts): cast the cloud to float32 first")
        narrow = points.to(torch.float32)
        if not bool((narrow.to(torch.float64) == points).all().item()):
            raise NotImplementedError(
                "downsample_method='center' on a float64 cloud that is not "
                "float32-representable")
        c, i = keypoints_device(narrow, voxel_size, 'center', jitter, seed,
                                fork_kdtree, num_out, k_hint)
        return _lib.tag_count(c.to(torch.float64), _lib.count_of(c)), i
    ws_bytes = lib.pgnn_keypoints_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    kp_idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    kp_xyz = torch.empty((max(n, 1), 3), dtype=points.dtype, device=dev)
    num = num_out if num_out is not None else \
        torch.empty(2, dtype=torch.int32, device=dev)  # always written
    st = _lib.stream_ptr()
    if method == 'center':
        _lib.check(lib.pgnn_voxel_keypoints_center(
            _lib.ptr(points), n, float(voxel_size), _lib.ptr(ws), ws_bytes,
            _lib.ptr(kp_idx), _lib.ptr(kp_xyz), _lib.ptr(num), st,
            # aux stream = NULL: the kd-tree replica runs on this stream.
            # With an aux stream (fork_kdtree: True or a stream) it runs
            # beside the voxel hashing -- 0.15 ms off a single frame's build,
            # nothing for a pipeline of frames, whose other streams keep the
            # chip busy anyway; the overlapped build (overlap_build) passes a
            # stream on a hardware queue of its own.
            ctypes.c_void_p(0 if not fork_kdtree else
                            fork_kdtree.cuda_stream
                            if isinstance(fork_kdtree, torch.cuda.Stream)
                            else _aux_stream(dev).cuda_stream)),
            "pgnn_voxel_keypoints_center")
    elif method == 'random':
        jit = None
        jp = ctypes.c_void_p(0)
        if jitter is not None:
            jit = np.ascontiguousarray(np.asarray(jitter, np.float64).reshape(3))
            jp = ctypes.c_void_p(jit.ctypes.data)
        _lib.check((lib.pgnn_voxel_keypoints_random_f64 if wide else
                    lib.pgnn_voxel_keypoints_random)(
            _lib.ptr(points), n, float(voxel_size), jp,
            int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(ws), ws_bytes,
            _lib.ptr(kp_idx), _lib.ptr(kp_xyz), _lib.ptr(num), st),
            "pgnn_voxel_keypoints_random")
    else:
        raise ValueError("unknown downsample method %r" % (method,))
    if num_out is not None:   # capacity form: K stays on the device
        cnt = _lib.DeviceCount(num[0:1], k_hint)
        return (_lib.tag_count(kp_xyz, cnt),
                _lib.tag_count(kp_idx.reshape(-1, 1), cnt))
    k, status = num.tolist()   # the one host read
    check_kd_status(status)
    return kp_xyz[:k], kp_idx[:k].reshape(k, 1)


def _keypoints_from(cloud, base, voxel_size, method, jitter, seed):
    """keypoints_device for a pooling level above the first (host-sized form
    only: the previous level's size is an argument of the call)."""
    lib = _lib.load()
    if _lib.count_of(base) is not None or _lib.count_of(cloud) is not None:
        raise NotImplementedError(
            "more than one pooling level in capacity form (deferred_counts)")
    n, nb = int(cloud.shape[0]), int(base.shape[0])
    dev = cloud.device
    st = _lib.stream_ptr()
    num = torch.empty(2, dtype=torch.int32, device=dev)
    if method == 'center':
        if cloud.dtype != torch.float32 or base.dtype != torch.float32:
            raise NotImplementedError(
                "downsample_method='center' above the first pooling level "
                "takes float32 coordinates")
        cloud, base = cloud.contiguous(), base.contiguous()
        ws_bytes = lib.pgnn_keypoints_from_workspace_bytes(n, nb)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        kp_idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        kp_xyz = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
        _lib.check(lib.pgnn_voxel_keypoints_center_from(
            _lib.ptr(cloud), n, _lib.ptr(base), nb, float(voxel_size),
            _lib.ptr(ws), ws_bytes, _lib.ptr(kp_idx), _lib.ptr(kp_xyz),
            _lib.ptr(num), st), "pgnn_voxel_keypoints_center_from")
    elif method == 'random':
        if cloud.dtype != base.dtype:
            raise ValueError("keypoints: cloud and base differ in dtype")
        wide = base.dtype == torch.float64
        cloud, base = cloud.contiguous(), base.contiguous()
        ws_bytes = lib.pgnn_keypoints_workspace_bytes(nb)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        kp_idx = torch.empty(max(nb, 1), dtype=torch.int32, device=dev)
        kp_xyz = torch.empty((max(nb, 1), 3), dtype=base.dtype, device=dev)
        jit = None
        jp = ctypes.c_void_p(0)
        if jitter is not None:
            jit = np.ascontiguousarray(np.asarray(jitter, np.float64).reshape(3))
            jp = ctypes.c_void_p(jit.ctypes.data)
        _lib.check((lib.pgnn_voxel_keypoints_random_from_f64 if wide else
                    lib.pgnn_voxel_keypoints_random_from)(
            _lib.ptr(base), nb, _lib.ptr(cloud), n, float(voxel_size), jp,
            int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(ws), ws_bytes,
            _lib.ptr(kp_idx), _lib.ptr(kp_xyz), _lib.ptr(num), st),
            "pgnn_voxel_keypoints_random_from")
    else:
        raise ValueError("unknown downsample method %r" % (method,))
    k, status = num.tolist()   # the one host read
    check_kd_status(status)
    return kp_xyz[:k], kp_idx[:k].reshape(k, 1)


def check_kd_status(status):
    """The tie-order status of the 'center' keypoints' kd-tree replica
    (num_keypoints[1] of pgnn_voxel_keypoints_center): non-zero means the
    frame's exact nearest-neighbour ties may be broken differently from the
    reference's sklearn call."""
    if status != 0:
        msg = ("kd-tree replica: the cloud is outside what the replica "
               "reproduces (status %d); exact nearest-neighbour ties of "
               "'center' keypoints may differ from the reference's sklearn "
               "order" % status)
        if KD_STATUS_POLICY == 'raise':
            raise _lib.PointGnnHipError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning)


def kdtree_replica(points):
    """scikit-learn's KDTree(points, leaf_size=30) arrays rebuilt on the
    device (csrc/kdtree.hip): returns (idx_array int32 [n], node_bounds float64
    [n_nodes, 6] = lo xyz | hi xyz, status int).  Diagnostic / test entry; the
    'center' keypoint kernel builds the same tree internally to break exact
    nearest-neighbour ties the way graph_gen.py:84-88's sklearn call does."""
    lib = _lib.load()
    p, was_np = _to_dev_f32(points)
    n = int(p.shape[0])
    lv = ctypes.c_int32()
    nodes = ctypes.c_int32()
    _lib.check(lib.pgnn_kdtree_shape(n, ctypes.byref(lv), ctypes.byref(nodes)),
               "pgnn_kdtree_shape")
    ws_bytes = lib.pgnn_kdtree_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=p.device)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=p.device)
    bounds = torch.empty((nodes.value, 6), dtype=torch.float64,
                         device=p.device)
    status = torch.zeros(1, dtype=torch.int32, device=p.device)
    _lib.check(lib.pgnn_kdtree_replica(
        _lib.ptr(p), n, _lib.ptr(ws), ws_bytes, _lib.ptr(idx),
        _lib.ptr(bounds), _lib.ptr(status), _lib.stream_ptr()),
        "pgnn_kdtree_replica")
    idx = idx[:n]
    if was_np:
        return idx.cpu().numpy(), bounds.cpu().numpy(), int(status.item())
    return idx, bounds, int(status.item())


def _multi_layer_downsampling(points_xyz, base_voxel_size, levels, add_rnd3d,
                              method, num_out=None, k_hint=0,
                              fork_kdtree=False):
    p, was_np = _to_dev(points_xyz)
    coords = [p]
    kp_list = []
    # 'random': the origin jitters of the pooling levels are drawn from NumPy's
    # global RNG in level order, like the reference's (graph_gen.py:126-128 --
    # its member choice consumes Python's `random`, not NumPy's); the seeds of
    # the device's member choice are drawn behind them.  (One pooling level:
    # jitter, then seed.)
    jitters, seeds = {}, {}
    if method == 'random':
        last_level = 0
        pooling = []
        for li, level in enumerate(levels):
            if not np.isclose(last_level, level):
                pooling.append((li, level))
            last_level = level
        if add_rnd3d:
            for li, level in pooling:
                jitters[li] = base_voxel_size * level * np.random.random(3)
        for li, level in pooling:
            seeds[li] = int(np.random.randint(0, 2 ** 31 - 1))
    last_level = 0
    for li, level in enumerate(levels):
        base = coords[-1]
        if np.isclose(last_level, level):
            # same scale (a GNN level): same vertices, identity keypoints
            coords.append(base)
            kp_list.append(_lib.tag_count(
                _identity_indices(int(base.shape[0]), base.device),
                _lib.count_of(base)))
        else:
            # (a pooling level above the first: the cloud still anchors the
            # voxel grid, graph_gen.py:41-45 / :108-110; the previous level's
            # vertices are searched / voxelised)
            upper = len(coords) != 1
            voxel = base_voxel_size * level
            if np.ndim(voxel) != 0:
                raise NotImplementedError("per-axis voxel sizes")
            if method == 'center':
                if add_rnd3d:
                    raise NotImplementedError(
                        "add_rnd3d with downsample_method='center'")
                if upper:
                    c, i = keypoints_device(p, voxel, 'center', base=base)
                else:
                    c, i = keypoints_device(base, voxel, 'center',
                                            fork_kdtree=fork_kdtree,
                                            num_out=num_out, k_hint=k_hint)
            else:
                jitter = jitters.get(li)  # graph_gen.py:126-128
                seed = seeds[li]
                if upper:
                    c, i = keypoints_device(p, voxel, 'random', jitter, seed,
                                            base=base)
                else:
                    c, i = keypoints_device(base, voxel, 'random', jitter, seed,
                                            num_out=num_out, k_hint=k_hint)
            coords.append(c)
            kp_list.append(i)
        last_level = level
    return coords, kp_list, was_np


def multi_layer_downsampling_select(points_xyz, base_voxel_size, levels=[1],
                                    add_rnd3d=False):
    """graph_gen.py:49-90 ('center' keypoints)."""
    coords, kps, was_np = _multi_layer_downsampling(
        points_xyz, base_voxel_size, levels, add_rnd3d, 'center')
    if was_np:
        return ([c.cpu().numpy() for c in coords],
                [k.cpu().numpy() for k in kps])
    return coords, kps


def multi_layer_downsampling_random(points_xyz, base_voxel_size, levels=[1],
                                    add_rnd3d=False):
    """graph_gen.py:92-153 ('random' keypoints)."""
    coords, kps, was_np = _multi_layer_downsampling(
        points_xyz, base_voxel_size, levels, add_rnd3d, 'random')
    if was_np:
        return ([c.cpu().numpy() for c in coords],
                [k.cpu().numpy() for k in kps])
    return coords, kps


def gen_multi_level_local_graph_v3(points_xyz, base_voxel_size, level_configs,
                                   add_rnd3d=False, downsample_method='center',
                                   deferred_counts=None, overlap_build=False):
    """graph_gen.py:155-195.  Returns (vertex_coord_list,
    keypoint_indices_list, edges_list).

    `deferred_counts` (extension; a CountHints, device tensors only): the
    capacity form.  Nothing is read back while the graph is built -- the
    reference's builder never waits for a size either (NumPy arrays carry
    theirs) -- so the call returns as soon as the kernels are enqueued.  The
    lists then hold capacity-sized tensors tagged with their device-side
    counts (`_lib.count_of`), which the operators of pointgnn_amd.gnn accept
    as they are; `edges_list[0]._pgnn_count.frame` is the FrameCounts record
    the caller reads (once) when it takes the frame's results.
    `overlap_build` (with deferred_counts): issue the parts of the build that
    do not depend on each other on side streams (level-0 cell grid and kd-tree
    replica beside the voxel hash, level-0 queries beside the level-1 graph),
    joined back into the current stream before returning -- the same kernels
    and results, a shorter critical path for a single frame."""
    if isinstance(base_voxel_size, list):
        base_voxel_size = np.array(base_voxel_size)
    scales = [cfg['graph_scale'] for cfg in level_configs]
    if downsample_method not in ('center', 'random'):
        raise ValueError("unknown downsample_method %r" % (downsample_method,))
    if deferred_counts is not None:
        return _multi_level_graph_deferred(
            points_xyz, base_voxel_size, level_configs, scales, add_rnd3d,
            downsample_method, deferred_counts, overlap_build)
    coords, kps, was_np = _multi_layer_downsampling(
        points_xyz, base_voxel_size, scales, add_rnd3d, downsample_method)
    edges_list = []
    batched = not was_np and all(
        cfg['graph_gen_method'] == 'disjointed_rnn_local_graph_v3' and
        cfg['graph_gen_kwargs'].get('num_neighbors', -1) <= 0
        for cfg in level_configs)
    if batched:
        # inference kwargs (no fan-in cap), device tensors: the levels' radius
        # graphs are independent once the keypoints exist
        edges_list = radius_graphs_device([
            (coords[cfg['graph_level']], coords[cfg['graph_level'] + 1],
             cfg['graph_gen_kwargs']['radius'],
             cfg['graph_gen_kwargs'].get('scale'))
            for cfg in level_configs])
    else:
        for cfg in level_configs:
            lvl = cfg['graph_level']
            fn = get_graph_generate_fn(cfg['graph_gen_method'])
            edges_list.append(fn(coords[lvl], coords[lvl + 1],
                                 **cfg['graph_gen_kwargs']))
    if was_np:
        return ([c.cpu().numpy() for c in coords],
                [k.cpu().numpy() for k in kps],
                [e.cpu().numpy() for e in edges_list])
    return coords, kps, edges_list


def _multi_level_graph_deferred(points_xyz, base_voxel_size, level_configs,
                                scales, add_rnd3d, downsample_method, hints,
                                overlap=False):
    if not isinstance(points_xyz, torch.Tensor):
        raise ValueError("deferred_counts needs device tensors (a NumPy "
                         "result has to know its size)")
    for cfg in level_configs:
        if cfg['graph_gen_method'] != 'disjointed_rnn_local_graph_v3':
            raise NotImplementedError(
                "deferred_counts: generator %r has no capacity form"
                % (cfg['graph_gen_method'],))
    n_levels = len(level_configs)
    dev = _device()
    # fan-in caps (training kwargs; any other method name skips the cap like
    # the reference, graph_gen.py:210): seeds drawn here, on the host, in
    # level order -- the draws the host-sized calls would make
    fan_k = [int(cfg['graph_gen_kwargs'].get('num_neighbors', -1) or -1)
             if cfg['graph_gen_kwargs'].get(
                 'neighbors_downsample_method', 'random') == 'random' else -1
             for cfg in level_configs]
    capped = any(k > 0 for k in fan_k)
    seeds = [None] * n_levels
    counts = _zero_counts(2 + (4 if capped else 2) * n_levels, dev)
    caps = [hints.cap(l) for l in range(n_levels)]
    frame = FrameCounts(counts, caps)

    def fan_in(l):
        if fan_k[l] <= 0:
            return None
        if seeds[l] is None:
            seeds[l] = int(np.random.randint(0, 2 ** 31 - 1))
        at = 2 + 2 * n_levels + 2 * l
        return (fan_k[l], seeds[l], hints.raw_cap(l), counts[at:at + 2])

    def level_job(l, points, centers_cap):
        kw = level_configs[l]['graph_gen_kwargs']
        return _RadiusDynJob(points, _lib.count_of(points), centers_cap,
                             kw['radius'], kw.get('scale'), caps[l],
                             counts[2 + 2 * l:4 + 2 * l], hints.edge_hint(l),
                             fan_in(l))

    # Overlapped build (single-frame latency): the level-0 cell grid needs
    # the cloud only, so it runs on a side stream BESIDE the keypoint
    # selection (whose kd-tree replica forks onto a third stream); afterwards
    # level 0's queries (this stream) run beside the whole level-1 graph (side
    # stream).  Same kernels, same buffers, same results; every buffer is
    # allocated on this stream and the side stream is joined back before the
    # function returns, so the caching allocator's stream rule holds.
    p = _to_dev(points_xyz)[0]
    n = int(p.shape[0])
    # (with a cap the levels' seeds are drawn where the host-sized calls draw
    # them -- after the keypoints' draws -- so the jobs cannot be made early)
    overlap = bool(overlap) and not capped and n > 0 and \
        p.dtype == torch.float32 and n_levels >= 1 and level_configs[0]['graph_level'] == 0 and \
        not np.isclose(scales[0], 0)
    job0 = ov = None
    if overlap:
        ov = _overlap_set(dev)
        job0 = level_job(0, p, n)      # keypoint capacity = n rows
        ov.fork.record()
    # (the keypoint call comes first: its kd-tree replica is the longest
    # chain of the build and should be the first thing the device sees)
    coords, kps, _ = _multi_layer_downsampling(
        p, base_voxel_size, scales, add_rnd3d, downsample_method,
        num_out=counts[0:2], k_hint=hints.k,
        fork_kdtree=ov.aux if overlap else False)
    if overlap:
        ov.side.wait_event(ov.fork)
        with torch.cuda.stream(ov.side):
            job0.grid()
            ov.grid.record()
    for t in list(coords) + list(kps):
        c = _lib.count_of(t)
        if c is not None:
            c.frame = frame
    edges_list = [None] * n_levels
    side_jobs = []
    if overlap:
        ov.kp.record()
        for l in range(1, n_levels):   # allocate on this stream ...
            lvl = level_configs[l]['graph_level']
            pts, ctr = coords[lvl], coords[lvl + 1]
            side_jobs.append((l, level_job(l, pts, int(ctr.shape[0])), ctr))
        ov.side.wait_event(ov.kp)
        with torch.cuda.stream(ov.side):   # ... enqueue on the side stream
            for l, job, ctr in side_jobs:
                job.grid()
                edges_list[l] = job.query(ctr, _lib.count_of(ctr))
            ov.join.record()
        torch.cuda.current_stream(dev).wait_event(ov.grid)
        edges_list[0] = job0.query(coords[1], _lib.count_of(coords[1]))
        torch.cuda.current_stream(dev).wait_event(ov.join)
    else:
        for l, cfg in enumerate(level_configs):
            lvl = cfg['graph_level']
            edges_list[l] = radius_graph_dyn_device(
                coords[lvl], coords[lvl + 1], cfg['graph_gen_kwargs']['radius'],
                cfg['graph_gen_kwargs'].get('scale'), caps[l],
                counts[2 + 2 * l:4 + 2 * l], hints.edge_hint(l), fan_in(l))
    for e in edges_list:
        _lib.count_of(e).frame = frame
    # job0 / side_jobs (their workspaces) die here: after the join was enqueued
    return coords, kps, edges_list


def gen_multi_level_local_graph_v3_one_read(points_xyz, hints, **kwargs):
    """gen_multi_level_local_graph_v3 for a caller that needs host-sized
    tensors (the training fetch: label assignment, box encoding and the batch
    concatenation all take sizes) but should not wait three times per frame:
    the graph is built in capacity form -- fan-in cap included -- and ONE read
    of the frame's record (K, edge counts) cuts the views.  `hints` (a
    CountHints) is updated; a frame that overflows a capacity is rebuilt
    host-sized from the same random state, so the result is the one the plain
    call returns for the same NumPy RNG state either way."""
    state = np.random.get_state()
    coords, kps, edges = gen_multi_level_local_graph_v3(
        points_xyz, deferred_counts=hints, **kwargs)
    frame = _lib.count_of(edges[0]).frame
    host = frame.read()
    check_kd_status(frame.kd_status)
    hints.update(frame.k, frame.edges, frame.raw_edges)
    if frame.overflowed:
        after = np.random.get_state()
        np.random.set_state(state)
        out = gen_multi_level_local_graph_v3(points_xyz, **kwargs)
        np.random.set_state(after)
        return out
    base = frame.tensor.storage_offset()

    def cut(t):
        c = _lib.count_of(t)
        if c is None:
            return t
        v = t[:host[c.dev.storage_offset() - base]]
        if getattr(t, '_pgnn_sorted', 0):
            v._pgnn_sorted = 1
        return v
    return [cut(t) for t in coords], [cut(t) for t in kps], \
        [cut(t) for t in edges]


def get_graph_generate_fn(method_name):
    """graph_gen.py:222-227."""
    method_map = {
        'disjointed_rnn_local_graph_v3': gen_disjointed_rnn_local_graph_v3,
        'multi_level_local_graph_v3': gen_multi_level_local_graph_v3,
    }
    return method_map[method_name]
