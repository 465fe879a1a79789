"""Frame-level driver of the hot path: what `run.py`'s per-frame loop does
between "fetch input" and "decode box" (run.py:219-263), device-resident.

    engine = InferenceEngine(config, params)
    logits, boxes = engine.run_frame(xyz_cuda, intensity_cuda)
    outs = engine.run_frames_on_streams(frames, 3)   # steady state: whole
                                                     # frames on three streams

`run_frame` sizes its arrays on the host (two reads per frame in the graph
builder).  `run_frame_deferred`, `run_frames_on_streams` and `capture_frame`
use the capacity form (graph_gen `deferred_counts`): K and the edge counts
stay on the device and are read once, with the frame's results.
Phase names follow run.py's `time_dict` keys ("gen graph", "gnn inference").
`shard_frames` is the multi-GPU decomposition: frames are independent units,
rank r takes frames r, r+W, ... -- inference needs no collective
(SURVEY.md §8e).
"""
import time

import torch

from . import graph_gen, models

__all__ = ["InferenceEngine", "DeferredFrame", "CapturedFrame", "shard_frames",
           "concurrent_streams"]

_CONCURRENT = {}


def _shares_queue(busy, cand, scratch, spin_cycles):
    """True when work queued on `cand` waits behind work on `busy`: the two
    HIP streams sit on the same hardware queue."""
    with torch.cuda.stream(busy):
        torch.cuda._sleep(spin_cycles)
    ev = torch.cuda.Event()
    with torch.cuda.stream(cand):
        scratch.add_(1)
        ev.record()
    t0 = time.perf_counter()
    ev.synchronize()
    waited = time.perf_counter() - t0
    busy.synchronize()
    return waited


def concurrent_streams(n, device=None):
    """`n` side streams that really run beside the current stream and beside
    each other.  HIP multiplexes streams onto a few hardware queues in creation
    order, and two streams on one queue execute strictly one after the other:
    measured here, the 7th torch stream of a process landed on the current
    stream's queue and the frame pipeline / the training step's graph-build
    stream lost all overlap (4.7 -> 6.3 ms per training step).  Candidates are
    probed once (a ~1 ms spin kernel on one stream, a trivial kernel on the
    other: does the trivial one wait?) and the result is cached per (device,
    current stream)."""
    dev = torch.device("cuda", torch.cuda.current_device()) \
        if device is None or device.index is None else device
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    have = _CONCURRENT.setdefault(key, [])
    if len(have) >= n:
        return have[:n]
    scratch = torch.zeros(64, device=dev)
    # calibrate the spin to ~1.5 ms
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    torch.cuda._sleep(1000000)
    e1.record()
    e1.synchronize()
    per_cycle = max(e0.elapsed_time(e1), 1e-3) / 1e6          # ms per cycle
    spin = int(1.5 / per_cycle)
    tried = 0
    while len(have) < n and tried < 24:
        cand = torch.cuda.Stream(device=dev)
        tried += 1
        clash = False
        for busy in [cur] + have:
            if _shares_queue(busy, cand, scratch, spin) > 0.6e-3:
                clash = True
                break
        if not clash:
            have.append(cand)
    while len(have) < n:      # fewer independent queues than asked for
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def shard_frames(num_frames, rank, world_size):
    """Indices of the frames rank `rank` of `world_size` processes."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, num_frames, world_size))


class DeferredFrame(object):
    """A frame enqueued in capacity form (InferenceEngine.run_frame_deferred):
    the graph builder and the model ran without the host reading K or an edge
    count.  `result()` is the frame's one host read: it waits for the frame,
    learns (K, E0, E1, ...), and returns (logits [K, nc], box_encodings
    [K, nc, len]) -- views of the capacity-sized outputs."""

    def __init__(self, engine, xyz, intensity, logits, boxes, counts):
        self.engine = engine
        self.xyz, self.intensity = xyz, intensity
        self.logits, self.boxes = logits, boxes
        self.counts = counts
        self._out = None

    def result(self, host_counts=None):
        if self._out is not None:
            return self._out
        if host_counts is not None:
            self.counts._host = [int(v) for v in host_counts]
        c = self.counts
        eng = self.engine
        graph_gen.check_kd_status(c.kd_status)
        k, edges = c.k, c.edges
        eng._hints = (eng._hints or graph_gen.CountHints()).update(k, edges)
        if c.overflowed:
            # an edge list did not fit the capacity the hints gave it (the
            # hints now hold 2 x this frame's sizes): rebuild the frame
            # with host-read sizes
            eng.deferred_overflows += 1
            self._out = eng.run_frame(self.xyz, self.intensity)
        else:
            eng.frame_shapes.append((k,) + tuple(edges))
            self._out = (self.logits[:k], self.boxes[:k])
        return self._out


class CapturedFrame(object):
    """One frame -- graph build and message passing, ~110 launches through the
    C ABI -- captured into ONE hipGraph (InferenceEngine.capture_frame).  That
    is possible because the capacity form leaves every data-dependent size (K,
    E0, E1) on the device: the launch sequence depends only on the number of
    points N, the capacities and the size hints, so the same graph serves every
    cloud of N points (a frame that overflows a capacity is detected as usual
    and rebuilt eagerly).  `replay(xyz, intensity)` copies the cloud into the
    graph's input buffers, launches the graph and returns a DeferredFrame."""

    def __init__(self, engine, graph, stream, xyz, intensity, frame):
        self.engine, self.graph, self.stream = engine, graph, stream
        self.xyz, self.intensity, self.frame = xyz, intensity, frame

    def replay(self, xyz, intensity):
        if tuple(xyz.shape) != tuple(self.xyz.shape) or \
                tuple(intensity.shape) != tuple(self.intensity.shape):
            raise ValueError(
                "captured for clouds of %d points, got %d: the point count "
                "shapes the kd-tree launches, capture one graph per count"
                % (self.xyz.shape[0], xyz.shape[0]))
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.xyz.copy_(xyz, non_blocking=True)
            self.intensity.copy_(intensity, non_blocking=True)
            self.graph.replay()
        cur.wait_stream(self.stream)
        f = self.frame
        out = DeferredFrame(self.engine, xyz, intensity, f.logits, f.boxes,
                            graph_gen.FrameCounts(f.counts.tensor,
                                                  f.counts.edge_caps))
        return out


class InferenceEngine(object):
    def __init__(self, config, params, box_encoding_len=7, device=None,
                 edge_arith='f32'):
        """`edge_arith`: arithmetic of the per-edge product (gnn.EDGE_ARITHS);
        also settable later through `engine.model.edge_arith`."""
        self.config = config
        self.model = models.get_model(config['model_name'])(
            num_classes=config['num_classes'],
            box_encoding_len=box_encoding_len, mode='test',
            **config['model_kwargs'])
        self.model.edge_arith = edge_arith
        self.model.load_state_dict(params, device)
        self.graph_fn = graph_gen.get_graph_generate_fn(
            config['graph_gen_method'])
        self.graph_kwargs = config['runtime_graph_gen_kwargs']
        self.time_dict = {}
        self.last_graph = None
        # (K, E0, E1, ...) of every frame processed since the caller last
        # cleared it: the sizes are host-known as soon as the graph is built
        self.frame_shapes = []
        # capacity form (run_frame_deferred): what the next frame's sizes
        # are expected to be, learnt from the frames finished so far
        self._hints = None
        self.deferred_overflows = 0

    def _note_shape(self, graph):
        coords, _, edges = graph
        shape = (int(coords[1].shape[0]),) + tuple(int(e.shape[0])
                                                   for e in edges)
        self.frame_shapes.append(shape)
        self._hints = (self._hints or graph_gen.CountHints()).update(
            shape[0], shape[1:])

    def build_graph_deferred(self, xyz, overlap=False):
        """build_graph in capacity form: no size is read back (graph_gen's
        `deferred_counts`).  Needs the sizes of an earlier frame as hints; the
        first frame of an engine therefore goes through run_frame.
        `overlap`: graph_gen's `overlap_build` (independent parts of the build
        on side streams: a shorter critical path for ONE frame; a pipeline of
        frames keeps the chip busy without it)."""
        if self._hints is None:
            raise RuntimeError("no size hints yet: run one frame through "
                               "run_frame() first")
        return self.graph_fn(xyz, deferred_counts=self._hints,
                             overlap_build=overlap, **self.graph_kwargs)

    def run_frame_deferred(self, xyz, intensity, overlap_build=False):
        """run_frame without a host wait: graph build and model are enqueued
        back to back, sizes stay on the device.  Returns a DeferredFrame; its
        .result() gives (logits, box_encodings)."""
        if self._hints is None or int(xyz.shape[0]) == 0:
            # no size hints yet, or an empty cloud (host-known: nothing to
            # defer, and empty tensors have no device pointers to hand over)
            out = self.run_frame(xyz, intensity)
            f = DeferredFrame(self, xyz, intensity, None, None, None)
            f._out = out
            return f
        graph = self.build_graph_deferred(xyz, overlap_build)
        coords, kps, edges = graph
        logits, boxes = self.model.predict(intensity, coords, kps, edges,
                                           is_training=False)
        self.last_graph = graph
        return DeferredFrame(self, xyz, intensity, logits, boxes,
                             edges[0]._pgnn_count.frame)

    def build_graph(self, xyz):
        """(vertex_coord_list, keypoint_indices_list, edges_list) on the
        device -- run.py:219-222."""
        return self.graph_fn(xyz, **self.graph_kwargs)

    def _pipeline_streams(self, graph_cus):
        """(graph stream, [compute streams]).  graph_cus > 0: the graph stream
        (and the aux stream its kd-tree build forks onto) may only use CUs
        [0, graph_cus) -- mask bits are striped over the XCDs, so that is
        graph_cus / 8 CUs of every XCD -- and the compute streams only the
        others (pgnn_stream_create_cu_mask); 0: ordinary streams sharing the
        whole device."""
        dev_index = torch.cuda.current_device()
        key = (dev_index, int(graph_cus))
        cache = self.__dict__.setdefault("_stream_sets", {})
        if key in cache:
            return cache[key]
        graph_cus = int(graph_cus)
        if graph_cus <= 0:
            # one graph stream + two compute streams on hardware queues of
            # their own (concurrent_streams); further compute streams, an
            # option that measured no gain, are ordinary ones
            streams = list(concurrent_streams(3))
            streams += [torch.cuda.Stream() for _ in range(2)]
            cache[key] = (streams[0], streams[1:])
            return cache[key]
        import ctypes
        from . import _lib
        lib = _lib.load()
        dev = torch.device("cuda", dev_index)
        owned = self.__dict__.setdefault("_owned_streams", [])

        def make(complement):
            p = ctypes.c_void_p()
            _lib.check(lib.pgnn_stream_create_cu_mask(
                0, graph_cus, complement, ctypes.byref(p)),
                "pgnn_stream_create_cu_mask")
            owned.append(p.value)     # destroyed by close()
            return torch.cuda.ExternalStream(p.value, device=dev)
        sg, aux = make(0), make(0)
        graph_gen._AUX_STREAMS[(dev.index, sg.cuda_stream)] = aux
        cache[key] = (sg, [make(1) for _ in range(4)])
        return cache[key]

    def close(self):
        """Destroy the CU-masked streams this engine created
        (pgnn_stream_create_cu_mask); ordinary torch streams need nothing.
        Explicit only (never from __del__): torch's caching allocator keeps
        blocks and recorded events keyed by the streams they were used on, so
        the streams may go only after the device is idle and the cache has
        been emptied -- and the caller must hold no tensor produced by a
        pipelined run any more."""
        owned = self.__dict__.pop("_owned_streams", [])
        self.__dict__.pop("_stream_sets", None)
        if owned:
            from . import _lib
            for k in [k for k, s in graph_gen._AUX_STREAMS.items()
                      if s.cuda_stream in owned or k[1] in owned]:
                del graph_gen._AUX_STREAMS[k]
            for k in [k for k in _lib._SCHED_WS if k[1] in owned]:
                del _lib._SCHED_WS[k]
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            lib = _lib.load()
            for h in owned:
                _lib.check(lib.pgnn_stream_destroy(h), "pgnn_stream_destroy")

    def _builder_streams(self, n, compute):
        """`n` streams for graph builds that share no hardware queue with each
        other or with the `compute` streams in use, as far as the device has
        queues (concurrent_streams probes them; HIP gives a process 4 by
        default, GPU_MAX_HW_QUEUES raises that); the first one is the
        pipeline's graph stream."""
        sg, _ = self._pipeline_streams(0)
        out = [sg]
        for s in concurrent_streams(1 + len(compute) + n):
            if len(out) >= n:
                break
            if s is not sg and all(s is not c for c in compute):
                out.append(s)
        return out

    def capture_frame(self, xyz, intensity, overlap_build=False):
        """Capture run_frame_deferred for clouds shaped like `xyz` /
        `intensity` into a hipGraph (see CapturedFrame).  The outputs of a
        replay live in the graph's own buffers: take `.result()` (or copy)
        before the next replay.  Size hints and capacities are those of the
        engine at capture time."""
        if self._hints is None:
            self.run_frame(xyz, intensity)
            self.frame_shapes.pop()
        # ONE capture stream per engine and device: _lib._SCHED_WS, the overlap
        # and concurrent-stream caches are keyed by the current stream, so a
        # fresh stream per capture leaked entries and re-ran the stream probes
        key = ("capture", torch.cuda.current_device())
        cache = self.__dict__.setdefault("_stream_sets", {})
        side = cache.get(key)
        if side is None:
            side = cache[key] = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # every lazily built thing (weight images, LDS attributes, CU
            # counts, counters) must exist before the capture starts
            for _ in range(2):
                self.run_frame_deferred(xyz, intensity,
                                        overlap_build).result()
                self.frame_shapes.pop()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        xs, fs = xyz.clone(), intensity.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            frame = self.run_frame_deferred(xs, fs, overlap_build)
        torch.cuda.synchronize()
        return CapturedFrame(self, graph, side, xs, fs, frame)

    def run_frames_on_streams(self, frames, n_streams=3, gnn_priority=False):
        """Steady-state loop in capacity form: frame i -- graph build AND
        message passing -- runs wholly on stream i % n_streams, and nothing
        is read back until every frame is enqueued.  Frames are independent
        (SURVEY 8e), so the streams need no events between them: while one
        stream's persistent MFMA kernels fill the CUs, the latency-bound
        builder kernels of the other streams' frames run beside them (slowly:
        DESIGN 7) and the next frame's message passing is ready the moment the
        CUs free up.  Measured on car_600k: 1 stream 255, 2 streams 296, 3
        streams 309 frames/s (the builder / compute split of
        run_frames_pipelined with host-read sizes: 288).
        frames: iterable of (xyz, intensity) CUDA tensors.  Returns the list
        of (logits, box_encodings), complete on return (the one host read --
        every frame's sizes in one copy -- waits for the device)."""
        frames = list(frames)
        if not frames:
            return []
        if self._hints is None or not getattr(self, "_warm", False):
            self.run_frame(*frames[0])     # weight images, size hints
            self.frame_shapes.pop()
            self._warm = True
        streams = list(concurrent_streams(max(1, int(n_streams))))
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        pending = []
        t_enq = time.perf_counter()
        if gnn_priority:
            # tested negative (tools/sessions/r03_s33.sh): frame i's
            # graph build on stream i % n as before, its message passing on a
            # HIGH-priority partner stream (gnn_priority < 0: the other way
            # round).  -13 % either way: HIP's priority -1 streams share a
            # smaller pool of hardware queues and run in turn.
            hi = self._priority_streams(len(streams))
            for h in hi:
                h.wait_stream(cur)
            for i, (xyz, intensity) in enumerate(frames):
                b, h = streams[i % len(streams)], hi[i % len(streams)]
                if int(gnn_priority) < 0:    # the BUILD on the high-priority one
                    b, h = h, b
                if self._hints is None or int(xyz.shape[0]) == 0:
                    with torch.cuda.stream(b):
                        pending.append(self.run_frame_deferred(xyz, intensity))
                    continue
                with torch.cuda.stream(b):
                    graph = self.build_graph_deferred(xyz)
                h.wait_stream(b)
                coords, kps, edges = graph
                with torch.cuda.stream(h):
                    for t in list(coords) + list(kps) + list(edges):
                        t.record_stream(h)     # allocated on b, read on h
                    frame = edges[0]._pgnn_count.frame
                    frame.tensor.record_stream(h)
                    logits, boxes = self.model.predict(
                        intensity, coords, kps, edges, is_training=False)
                    # the next build on b may not overtake this frame's GNN by
                    # more than one frame (bounded memory, as before)
                b.wait_stream(h)
                self.last_graph = graph
                pending.append(DeferredFrame(self, xyz, intensity, logits,
                                             boxes, frame))
            for h in hi:
                cur.wait_stream(h)
        else:
            for i, (xyz, intensity) in enumerate(frames):
                with torch.cuda.stream(streams[i % len(streams)]):
                    pending.append(self.run_frame_deferred(xyz, intensity))
        for s in streams:
            cur.wait_stream(s)
        # host time spent enqueuing (no device wait is part of it unless the
        # launch queues filled up): bench.py reports it per frame
        self.last_enqueue_s = time.perf_counter() - t_enq
        live = [f for f in pending if f.counts is not None]
        host = torch.stack([f.counts.tensor for f in live]).tolist() \
            if live else []
        for f, h in zip(live, host):
            f.result(h)
        outs = [f.result() for f in pending]
        for lg, bx in outs:     # allocated on a side stream, used by the caller
            lg.record_stream(cur)
            bx.record_stream(cur)
        if not self._edge_range_clean():
            return self._rerun_batch_f32(
                lambda: self.run_frames_on_streams(frames, n_streams,
                                                   gnn_priority))
        return outs

    def _priority_streams(self, n):
        """n high-priority streams (HIP keeps a pool of hardware queues per
        priority level), made once per device."""
        key = ("hi", torch.cuda.current_device())
        cache = self.__dict__.setdefault("_stream_sets", {})
        have = cache.setdefault(key, [])
        while len(have) < n:
            have.append(torch.cuda.Stream(priority=-1))
        return have[:n]

    def run_frames_pipelined(self, frames, compute_streams=1, graph_cus=0,
                             lookahead=0, deferred=False, graph_streams=1):
        """Steady-state loop over independent frames on HIP streams: while a
        compute stream executes the GNN of frame i, stream G builds the graph
        of frame i+1.  The graph builder needs two host waits per frame (K, then
        E0 and E1, which size its outputs); they synchronise stream G only, so the host
        waits for them while the GPU is busy with frame i's message passing.
        With compute_streams = 2 consecutive frames alternate between two
        compute streams, so the under-filled per-vertex kernels and the tail
        of one frame's edge kernel overlap the next frame's work.
        graph_cus > 0 gives stream G that many CUs of its own and keeps the
        compute streams off them (see _pipeline_streams).
        deferred=True builds the graphs in capacity form (no host wait at
        all while frames are enqueued; sizes are read once, for all frames,
        at the end -- that read waits for the device).  Only then can
        graph_streams > 1 be used: consecutive frames' graphs are built on
        alternating streams, each build running that many frames ahead of its
        GNN.  Beside the persistent MFMA kernels a build takes about as long
        as a frame's message passing (its big-LDS kernels wait for kernel
        boundaries, the others run several times slower than alone: DESIGN
        7), so ONE builder stream is the pipeline's bottleneck on most
        frames; two are not.
        frames: iterable of (xyz, intensity) CUDA tensors.  Returns the list of
        (logits, box_encodings); outputs are complete after
        torch.cuda.synchronize() (or a wait on the compute streams)."""
        frames = list(frames)
        if not frames:
            return []
        if not getattr(self, "_warm", False):
            # first use: the packed weight images are built lazily (host pack +
            # pageable upload) on whichever stream runs first; do that on the
            # CURRENT stream, which every forked stream waits for below, so no
            # compute stream can read a half-uploaded image
            self.run_frame(*frames[0])
            self.frame_shapes.pop()
            self._warm = True
        sg, scs = self._pipeline_streams(graph_cus)
        scs = scs[:max(1, min(4, int(compute_streams)))]
        sgs = [sg]
        if deferred and graph_cus <= 0 and int(graph_streams) > 1:
            sgs = self._builder_streams(int(graph_streams), scs)
        cur = torch.cuda.current_stream()
        for s in tuple(sgs) + tuple(scs):
            s.wait_stream(cur)

        def build(i):
            s_build = sgs[i % len(sgs)]
            with torch.cuda.stream(s_build):
                g = self.build_graph_deferred(frames[i][0]) if deferred \
                    else self.build_graph(frames[i][0])
                ev = torch.cuda.Event()
                ev.record(s_build)
            return g, ev

        # lookahead > 0: graphs come from a builder thread that runs up to
        # that many frames ahead, so the count reads block only that
        # thread.  Measured: no gain over lookahead = 0 (the
        # calling thread builds graph i+1 right after enqueueing frame i; its
        # reads complete long before frame i's message passing does), so 0 is
        # the default.
        ready = None
        worker = None
        if lookahead > 0 and len(frames) > 1:
            import queue
            import threading
            ready = queue.Queue(maxsize=int(lookahead))
            dev_index = torch.cuda.current_device()

            def produce():
                try:
                    torch.cuda.set_device(dev_index)
                    for j in range(len(frames)):
                        ready.put(build(j))
                except BaseException as exc:  # re-raised by the consumer
                    ready.put(exc)
            worker = threading.Thread(target=produce, daemon=True)
            worker.start()

        # builds are enqueued `ahead` frames before their GNN: one per
        # builder stream
        ahead = len(sgs)
        queued = []

        def next_graph(i):
            if ready is None:
                while len(queued) < ahead and i + len(queued) < len(frames):
                    queued.append(build(i + len(queued)))
                return queued.pop(0)
            item = ready.get()
            if isinstance(item, BaseException):
                raise item
            return item

        outs = []
        try:
            graph, ev = next_graph(0)
            for i in range(len(frames)):
                sc = scs[i % len(scs)]
                sc.wait_event(ev)
                with torch.cuda.stream(sc):
                    coords, kps, edges = graph
                    for t in list(coords) + list(kps) + list(edges):
                        t.record_stream(sc)  # allocated on G, consumed on C
                    if deferred:
                        counts = edges[0]._pgnn_count.frame
                        counts.tensor.record_stream(sc)
                    out = self.model.predict(frames[i][1], coords, kps,
                                             edges, is_training=False)
                    outs.append(DeferredFrame(self, frames[i][0], frames[i][1],
                                              out[0], out[1], counts)
                                if deferred else out)
                self.last_graph = graph
                if not deferred:
                    self._note_shape(graph)
                if i + 1 < len(frames):
                    graph, ev = next_graph(i + 1)
        finally:
            if worker is not None:
                # drain so that a failing consumer cannot leave the builder
                # blocked on a full queue
                while worker.is_alive():
                    try:
                        ready.get(timeout=0.05)
                    except Exception:
                        pass
                worker.join()
        for s in scs:
            cur.wait_stream(s)
        if deferred:
            # the one host read, for all frames together (waits for them)
            host = torch.stack([f.counts.tensor for f in outs]).tolist()
            outs = [f.result(h) for f, h in zip(outs, host)]
            for lg, bx in outs:
                lg.record_stream(cur)
                bx.record_stream(cur)
        if not self._edge_range_clean():
            return self._rerun_batch_f32(
                lambda: self.run_frames_pipelined(
                    frames, compute_streams, graph_cus, lookahead, deferred,
                    graph_streams))
        return outs

    def _edge_range_clean(self):
        """edge_arith 'f16x2': True when no frame since the last look left
        fp16's safe range (always True for the other arithmetics)."""
        return self.model.edge_arith != 'f16x2' or self.model.edge_range_ok()

    def _rerun_batch_f32(self, run):
        """The range guard of 'f16x2' tripped somewhere in a batch (the flag is
        the model's, not a frame's): the batch again with the fp32 edge stage
        -- correct results instead of an exception and a discarded batch.
        `f16x2_batch_reruns` counts how often."""
        self.f16x2_batch_reruns = getattr(self, "f16x2_batch_reruns", 0) + 1
        shapes = len(self.frame_shapes)
        self.model.edge_arith = 'f32'
        try:
            # (the first pass noted the batch's shapes already)
            outs = run()
            del self.frame_shapes[shapes:]
            return outs
        finally:
            self.model.edge_arith = 'f16x2'

    def check_edge_range(self):
        """edge_arith 'f16x2' only: raises when an activation of a frame run
        since the last call left fp16's range (the kernel clamps at 65504 and
        flags >= 32768; a trained Point-GNN stays below 100) -- such frames
        must be rerun with edge_arith 'f32'.  One small device-to-host read:
        the batch entry points (run_frames, run_frames_on_streams) call it
        once per batch, callers of run_frame / run_frame_deferred call it when
        they take results."""
        if self.model.edge_arith == 'f16x2' and not self.model.edge_range_ok():
            from . import _lib
            raise _lib.PointGnnHipError(
                "edge_arith 'f16x2': a gathered activation reached 32768 "
                "(fp16 ends at 65504): rerun these frames with edge_arith "
                "'f32'")

    def run_frame(self, xyz, intensity, timed=False):
        """xyz [N,3] float32, intensity [N,F] float32 CUDA tensors ->
        (logits [K,nc], box_encodings [K,nc,7]) CUDA tensors.  With
        timed=True the two phases are wall-clocked (adds two device syncs)."""
        if timed:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        graph = self.build_graph(xyz)
        if timed:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        coords, kps, edges = graph
        out = self.model.predict(intensity, coords, kps, edges,
                                 is_training=False)
        if timed:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            d = self.time_dict
            d['gen graph'] = d.get('gen graph', 0.0) + (t1 - t0)
            d['gnn inference'] = d.get('gnn inference', 0.0) + (t2 - t1)
            d['frames'] = d.get('frames', 0) + 1
        self.last_graph = graph
        self._note_shape(graph)
        return out
