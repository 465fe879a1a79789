"""Frame-level driver of the hot path: what `run.py`'s per-frame loop does
between "fetch input" and "decode box" (run.py:219-263), device-resident.

    engine = InferenceEngine(config, params)
    logits, boxes = engine.run_frame(xyz_cuda, intensity_cuda)

Phase names follow run.py's `time_dict` keys ("gen graph", "gnn inference").
`shard_frames` is the multi-GPU decomposition: frames are independent units,
rank r takes frames r, r+W, ... -- inference needs no collective
(SURVEY.md §8e).
"""
import time

import torch

from . import graph_gen, models

__all__ = ["InferenceEngine", "shard_frames"]


def shard_frames(num_frames, rank, world_size):
    """Indices of the frames rank `rank` of `world_size` processes."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, num_frames, world_size))


class InferenceEngine(object):
    def __init__(self, config, params, box_encoding_len=7, device=None):
        self.config = config
        self.model = models.get_model(config['model_name'])(
            num_classes=config['num_classes'],
            box_encoding_len=box_encoding_len, mode='test',
            **config['model_kwargs'])
        self.model.load_state_dict(params, device)
        self.graph_fn = graph_gen.get_graph_generate_fn(
            config['graph_gen_method'])
        self.graph_kwargs = config['runtime_graph_gen_kwargs']
        self.time_dict = {}
        self.last_graph = None
        # (K, E0, E1, ...) of every frame processed since the caller last
        # cleared it: the sizes are host-known as soon as the graph is built
        self.frame_shapes = []

    def _note_shape(self, graph):
        coords, _, edges = graph
        self.frame_shapes.append(
            (int(coords[1].shape[0]),) + tuple(int(e.shape[0]) for e in edges))

    def build_graph(self, xyz):
        """(vertex_coord_list, keypoint_indices_list, edges_list) on the
        device -- run.py:219-222."""
        return self.graph_fn(xyz, **self.graph_kwargs)

    def run_frames_pipelined(self, frames, compute_streams=1):
        """Steady-state loop over independent frames on HIP streams: while a
        compute stream executes the GNN of frame i, stream G builds the graph
        of frame i+1.  The graph builder needs three host reads per frame (K,
        E0, E1 size its outputs); they synchronise stream G only, so the host
        waits for them while the GPU is busy with frame i's message passing.
        With compute_streams = 2 consecutive frames alternate between two
        compute streams, so the under-filled per-vertex kernels and the tail
        of one frame's edge kernel overlap the next frame's work.
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
        if not hasattr(self, "_streams"):
            self._streams = tuple(torch.cuda.Stream() for _ in range(5))
        sg = self._streams[0]
        scs = self._streams[1:1 + max(1, min(4, int(compute_streams)))]
        cur = torch.cuda.current_stream()
        for s in (sg,) + tuple(scs):
            s.wait_stream(cur)

        def build(i):
            with torch.cuda.stream(sg):
                g = self.build_graph(frames[i][0])
                ev = torch.cuda.Event()
                ev.record(sg)
            return g, ev

        outs = []
        graph, ev = build(0)
        for i in range(len(frames)):
            sc = scs[i % len(scs)]
            sc.wait_event(ev)
            with torch.cuda.stream(sc):
                coords, kps, edges = graph
                for t in list(coords) + list(kps) + list(edges):
                    t.record_stream(sc)  # allocated on G, consumed on C
                outs.append(self.model.predict(frames[i][1], coords, kps,
                                               edges, is_training=False))
            self.last_graph = graph
            self._note_shape(graph)
            if i + 1 < len(frames):
                graph, ev = build(i + 1)
        for s in scs:
            cur.wait_stream(s)
        return outs

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
