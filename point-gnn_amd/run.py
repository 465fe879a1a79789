"""The frame loop of the reference's `run.py` (:203-433) as two functions:
`detect_frame` is one trip through the loop body -- KITTI files -> camera-frame
crop -> graph -> GNN -> softmax -> decode + NMS -> the 16-field KITTI rows --
with every stage on the device, and `run_dataset` is the loop with the
reference's checkpoint restore (:192-201), per-stage `time_dict` (:216-263,
:326, :412) and `<output_dir>/data/<frame>.txt` files (:414-423).

No session, placeholders, visualisation or command line: `config` is the
dictionary `util/config_util.load_config` returns for the reference's config
files, `dataset` a `pointgnn_amd.kitti_dataset.KittiDataset`.
"""
import os
import time

import torch

from . import graph_gen, kitti_output, models, nms, tf_bundle

BOX_ENCODING_LEN = 7


def build_model(config, checkpoint_dir=None, params=None, edge_arith='f32'):
    """run.py:135-141 + :192-201: the model in 'test' mode with the weights of
    the latest checkpoint under `checkpoint_dir` (or a ready name->array
    mapping).  `edge_arith` (extension): gnn.EDGE_ARITHS, default the fp32
    path."""
    model = models.get_model(config['model_name'])(
        num_classes=config['num_classes'], box_encoding_len=BOX_ENCODING_LEN,
        mode='test', **config['model_kwargs'])
    if params is None:
        params = tf_bundle.load_checkpoint(checkpoint_dir)
    params = {k: v for k, v in params.items() if k != 'Variable'}
    model.load_state_dict(params)
    model.edge_arith = edge_arith
    return model


def _input_features(config, points):
    """run.py:224-249: the `input_features` switch of the config."""
    kind = config['input_features']
    attr = points.attr
    if kind == 'irgb':
        return attr
    if kind == 'i':
        return attr[:, :1]
    if kind == '0':
        return torch.zeros_like(attr[:, :1])
    if kind == '0000':
        return torch.zeros_like(attr)
    if kind == '0rgb':
        out = attr.clone()
        out[:, 0] = 0
        return out
    if kind == 'i000':
        out = torch.zeros_like(attr)
        out[:, 0] = attr[:, 0]
        return out
    raise ValueError("input_features %r" % (kind,))


def detect_frame(dataset, frame_idx, model, config, use_box_merge=True,
                 use_box_score=True, time_dict=None, image_reader=None):
    """One iteration of run.py:203-412.  Returns (rows, stages): the KITTI
    rows of the frame and the device tensors of every stage (points, graph,
    logits, box encodings, probs, NMS outputs) for callers that check them."""
    td = time_dict if time_dict is not None else {}

    def lap(key, t_prev):
        torch.cuda.synchronize()
        now = time.time()
        td[key] = td.get(key, 0.0) + now - t_prev
        return now

    t = time.time()
    # run.py:210-211 (colour channels are zero unless `image_reader` supplies
    # the decoded BGR image: PNG decoding is not part of this package)
    image = image_reader(frame_idx) if image_reader is not None else None
    points = dataset.get_cam_points_in_image_with_rgb(
        frame_idx, config['downsample_by_voxel_size'], image=image)
    calib = dataset.get_calib(frame_idx)
    t = lap('fetch input', t)
    fn = graph_gen.get_graph_generate_fn(config['graph_gen_method'])
    coords, kps, edges = fn(points.xyz, **config['runtime_graph_gen_kwargs'])
    t = lap('gen graph', t)
    input_v = _input_features(config, points)
    logits, box_encodings = model.predict(input_v, coords, kps, edges, False)
    if model.edge_arith == 'f16x2' and not model.edge_range_ok():
        # an activation left fp16's safe range (gnn.EDGE_ARITHS): this frame
        # in fp32 (the results are read below anyway: no extra wait)
        td['f16x2 range reruns'] = td.get('f16x2 range reruns', 0) + 1
        model.edge_arith = 'f32'
        try:
            logits, box_encodings = model.predict(input_v, coords, kps, edges,
                                                  False)
        finally:
            model.edge_arith = 'f16x2'
    probs = model.postprocess(logits)
    t = lap('gnn inference', t)
    label_map = kitti_output.LABEL_MAPS[config['label_method']]
    labels, boxes, scores, nms_idx = nms.detect_boxes(
        probs, box_encodings, coords[-1], label_map,
        config['nms_overlapped_thres'],
        box_encoding_method=config['box_encoding_method'],
        use_box_merge=use_box_merge, use_box_score=use_box_score)
    cand_idx, _ = nms.select_candidates(probs)
    cand_xyz = coords[-1][(cand_idx // config['num_classes']).long()]
    t = lap('decode box + nms', t)
    rows = kitti_output.detections_to_kitti_labels(
        labels, boxes, scores, calib, config['label_method'],
        candidate_xyz=cand_xyz, use_box_score=use_box_score)
    lap('kitti rows', t)
    stages = {'points': points, 'calib': calib, 'coords': coords, 'kps': kps,
              'edges': edges, 'logits': logits, 'box_encodings': box_encodings,
              'probs': probs, 'class_labels': labels, 'boxes_3d': boxes,
              'scores': scores, 'nms_indices': nms_idx,
              'candidate_indices': cand_idx, 'candidate_xyz': cand_xyz}
    return rows, stages


def run_dataset(dataset, config, checkpoint_dir, output_dir,
                frame_indices=None, use_box_merge=True, use_box_score=True,
                params=None, log=None, image_reader=None, edge_arith='f32',
                pipelined=True, in_flight=3, prefetch=4):
    """run.py:203-433 over `frame_indices` (default: the whole dataset).
    Writes `<output_dir>/data/<frame name>.txt` in the reference's format and
    returns the accumulated `time_dict` (seconds per stage under run.py's
    names, plus 'frames' and 'wall').

    pipelined (default): frames are kept in flight -- a loader thread reads
    and crops frame i+1.. while the graph build + GNN of `in_flight` frames
    run on as many streams, the decode + NMS of older frames behind them, and
    a writer thread turns finished frames into rows and files.  Same kernels,
    same files, byte for byte (tests/test_gpu_e2e.py); `time_dict` then holds
    DEVICE time between events for the device stages (they overlap: their sum
    exceeds 'wall') and host time for the others.  pipelined=False is the
    reference's strictly sequential loop, a device synchronisation after
    every stage."""
    model = build_model(config, checkpoint_dir, params, edge_arith)
    if frame_indices is None:
        frame_indices = range(dataset.num_files)
    frame_indices = list(frame_indices)
    time_dict = {}
    t_wall = time.time()
    if pipelined and len(frame_indices) > 1 and \
            config['graph_gen_method'] == 'multi_level_local_graph_v3':
        n = FramePipeline(dataset, config, model, output_dir, use_box_merge,
                          use_box_score, log, image_reader, in_flight,
                          prefetch, time_dict).run(frame_indices)
    else:
        n = 0
        for frame_idx in frame_indices:
            _frame_sequential(dataset, frame_idx, model, config, output_dir,
                              use_box_merge, use_box_score, time_dict,
                              image_reader, log)
            n += 1
    time_dict['frames'] = n
    time_dict['wall'] = time.time() - t_wall
    return time_dict


def _frame_sequential(dataset, frame_idx, model, config, output_dir,
                      use_box_merge, use_box_score, time_dict, image_reader,
                      log):
    rows, stages = detect_frame(dataset, frame_idx, model, config,
                                use_box_merge, use_box_score, time_dict,
                                image_reader)
    t = time.time()
    kitti_output.write_kitti_txt(
        os.path.join(output_dir, 'data',
                     dataset.get_filename(frame_idx) + '.txt'), rows)
    time_dict['write txt'] = time_dict.get('write txt', 0.0) + \
        time.time() - t
    if log is not None:
        log("frame %s: %d detections" % (dataset.get_filename(frame_idx),
                                         len(rows)))
    return stages


class _Frame(object):
    """One frame on its way through FramePipeline."""
    __slots__ = ("idx", "points", "calib", "stream", "ev", "graph", "probs",
                 "box_enc", "cand", "host_a", "n_rec", "host_b", "n_cand",
                 "rerun", "t_fetch")


class FramePipeline(object):
    """run.py's frame loop with frames in flight (see run_dataset).

    Per frame, in order, and who waits for what:
      loader thread   files -> upload -> crop kernel on the loader's own
                      stream, `prefetch` frames ahead, the crop's point count
                      on its way to pinned memory (no wait);
      stage A         (main thread, stream i % in_flight) read the point count
                      (it sizes every launch after it), graph build in
                      capacity form, GNN, softmax, candidate selection bounded
                      by the device-side K; the frame's count record + the
                      candidate count go to pinned memory;
      stage B         (`in_flight` frames later: A has long finished) read K,
                      E0, E1, candidates; decode + NMS of the candidates; the
                      results go to pinned memory;
      stage C         (`in_flight` frames later) read the kept boxes; a writer
                      thread builds the KITTI rows on the host and writes the
                      file.
    A frame the capacity form cannot hold (an edge list outgrew its capacity)
    and, under edge_arith 'f16x2', frames the range guard flags run through the
    sequential path instead."""

    def __init__(self, dataset, config, model, output_dir, use_box_merge,
                 use_box_score, log, image_reader, in_flight, prefetch,
                 time_dict):
        self.dataset, self.config, self.model = dataset, config, model
        self.output_dir = output_dir
        self.use_box_merge, self.use_box_score = use_box_merge, use_box_score
        self.log, self.image_reader = log, image_reader
        self.in_flight = max(1, int(in_flight))
        self.prefetch = max(1, int(prefetch))
        self.td = time_dict
        self.label_map = kitti_output.LABEL_MAPS[config['label_method']]
        self.hints = None
        self.fallbacks = 0
        # per-run totals: candidates that entered the NMS, boxes it kept
        self.stats = {'candidates': 0, 'kept': 0, 'rows': 0}

    def _add(self, key, seconds):
        self.td[key] = self.td.get(key, 0.0) + seconds

    # ---- loader thread ----------------------------------------------------------
    def _load(self, frame_idx, stream):
        t0 = time.time()
        with torch.cuda.stream(stream):
            image = self.image_reader(frame_idx) \
                if self.image_reader is not None else None
            calib = self.dataset.get_calib(frame_idx)
            points = self.dataset.get_cam_points_in_image_with_rgb(
                frame_idx, self.config['downsample_by_voxel_size'],
                calib=calib, image=image, deferred=True)
        f = _Frame()
        f.idx, f.points, f.calib, f.ev = frame_idx, points, calib, points.event
        f.rerun = False
        f.t_fetch = time.time() - t0
        return f

    # ---- stage A ------------------------------------------------------------------
    def _stage_a(self, f, stream):
        cfg, model = self.config, self.model
        f.stream = stream
        stream.wait_event(f.ev)
        with torch.cuda.stream(stream):
            f.points.xyz.record_stream(stream)
            f.points.attr.record_stream(stream)
            # the frame's one early read: how many points the crop kept (it
            # sizes every launch after it); enqueued `prefetch` frames ago
            f.points = f.points.result()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            graph = graph_gen.gen_multi_level_local_graph_v3(
                f.points.xyz, deferred_counts=self.hints,
                **cfg['runtime_graph_gen_kwargs'])
            ev[1].record()
            coords, kps, edges = graph
            logits, box_enc = model.predict(_input_features(cfg, f.points),
                                            coords, kps, edges, False)
            probs = model.postprocess(logits)
            ev[2].record()
            frame = edges[0]._pgnn_count.frame
            cand = nms.select_candidates_dyn(probs, frame.tensor[0:1])
            n_rec = int(frame.tensor.numel())
            host = torch.empty(n_rec + 1, dtype=torch.int32, pin_memory=True)
            host[:n_rec].copy_(frame.tensor, non_blocking=True)
            host[n_rec:].copy_(cand[2], non_blocking=True)
            ev[3].record()
        f.graph, f.probs, f.box_enc, f.cand = graph, probs, box_enc, cand
        f.host_a, f.n_rec, f.ev = host, n_rec, ev

    # ---- stage B ------------------------------------------------------------------
    def _stage_b(self, f):
        cfg = self.config
        f.ev[3].synchronize()
        rec = f.host_a.tolist()
        frame = f.graph[2][0]._pgnn_count.frame
        frame._host = rec[:f.n_rec]
        graph_gen.check_kd_status(frame.kd_status)
        k = frame.k
        self.hints = self.hints.update(k, frame.edges)
        self._add('gen graph', f.ev[0].elapsed_time(f.ev[1]) * 1e-3)
        self._add('gnn inference', f.ev[1].elapsed_time(f.ev[2]) * 1e-3)
        if frame.overflowed:
            f.rerun = True
        if self.model.edge_arith == 'f16x2' and not self.model.edge_range_ok():
            # the flag is the model's, not the frame's: every frame whose GNN
            # may have run since the last look goes through the fp32 path
            self.td['f16x2 range reruns'] = \
                self.td.get('f16x2 range reruns', 0) + 1
            f.rerun = True
            for g in self._in_a:
                g.rerun = True
        f.n_cand = rec[f.n_rec]
        if f.rerun:
            return
        coords = f.graph[0]
        with torch.cuda.stream(f.stream):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            n = f.n_cand
            host = None
            if n > 0:
                out, cnt, cand_xyz = nms.detect_candidates_deferred(
                    f.probs[:k], f.box_enc[:k], coords[-1][:k], f.cand[0][:n],
                    f.cand[1][:n], self.label_map, cfg['nms_overlapped_thres'],
                    box_encoding_method=cfg['box_encoding_method'],
                    use_box_merge=self.use_box_merge,
                    use_box_score=self.use_box_score)
                # one pinned block: kept | labels | boxes | scores | cand xyz
                host = torch.empty(1 + n + 7 * n + n + 3 * n,
                                   dtype=torch.float32, pin_memory=True)
                hi = host.view(torch.int32)
                hi[0:1].copy_(cnt, non_blocking=True)
                hi[1:1 + n].copy_(out[0], non_blocking=True)
                host[1 + n:1 + 8 * n].copy_(out[1].reshape(-1),
                                            non_blocking=True)
                host[1 + 8 * n:1 + 9 * n].copy_(out[2], non_blocking=True)
                host[1 + 9 * n:].copy_(cand_xyz.reshape(-1), non_blocking=True)
            e1.record()
        f.host_b = host
        f.ev = [e0, e1]
        # the device tensors of stage A are no longer needed by the host
        f.graph = f.probs = f.box_enc = f.cand = f.points = None

    # ---- stage C ------------------------------------------------------------------
    def _stage_c(self, f, writer):
        if f.rerun:
            self.fallbacks += 1
            edge_arith = self.model.edge_arith
            self.model.edge_arith = 'f32' if edge_arith == 'f16x2' \
                else edge_arith
            try:
                _frame_sequential(self.dataset, f.idx, self.model, self.config,
                                  self.output_dir, self.use_box_merge,
                                  self.use_box_score, self.td,
                                  self.image_reader, self.log)
            finally:
                self.model.edge_arith = edge_arith
            return
        f.ev[1].synchronize()
        self._add('decode box + nms', f.ev[0].elapsed_time(f.ev[1]) * 1e-3)
        self._add('fetch input', f.t_fetch)
        n = f.n_cand
        self.stats['candidates'] += n
        if n > 0:
            h = f.host_b.numpy()
            kept = int(h[:1].view('int32')[0])
            self.stats['kept'] += kept
            res = (h[1:1 + n].view('int32')[:kept].copy(),
                   h[1 + n:1 + 8 * n].reshape(n, 7)[:kept].copy(),
                   h[1 + 8 * n:1 + 9 * n][:kept].copy(),
                   h[1 + 9 * n:].reshape(n, 3).copy())
        else:
            import numpy as np
            res = (np.zeros(0, 'int32'), np.zeros((0, 7), 'float32'),
                   np.zeros(0, 'float32'), np.zeros((0, 3), 'float32'))
        writer.put((f.idx, f.calib, res))

    def _write(self, item):
        idx, calib, (labels, boxes, scores, cand_xyz) = item
        t = time.time()
        rows = kitti_output.detections_to_kitti_labels(
            labels, boxes, scores, calib, self.config['label_method'],
            candidate_xyz=cand_xyz, use_box_score=self.use_box_score,
            host_only=True)
        t1 = time.time()
        kitti_output.write_kitti_txt(
            os.path.join(self.output_dir, 'data',
                         self.dataset.get_filename(idx) + '.txt'), rows)
        t2 = time.time()
        return idx, len(rows), t1 - t, t2 - t1

    # ---- the loop -------------------------------------------------------------------
    def run(self, frame_indices):
        import queue
        import threading
        from collections import deque
        from .engine import concurrent_streams
        dev_index = torch.cuda.current_device()
        dev = torch.device("cuda", dev_index)
        # first frame: the sequential path (weight images, LDS attributes, the
        # size hints the capacity form starts from)
        st = _frame_sequential(self.dataset, frame_indices[0], self.model,
                               self.config, self.output_dir,
                               self.use_box_merge, self.use_box_score, self.td,
                               self.image_reader, self.log)
        self.hints = graph_gen.CountHints().update(
            int(st['coords'][1].shape[0]),
            [int(e.shape[0]) for e in st['edges']])
        del st
        rest = frame_indices[1:]
        streams = list(concurrent_streams(self.in_flight + 1, dev))
        compute, s_load = streams[:self.in_flight], streams[-1]
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        loaded = queue.Queue(maxsize=self.prefetch)
        to_write = queue.Queue()
        failure = []
        stop = threading.Event()

        def loader():
            try:
                torch.cuda.set_device(dev_index)
                for idx in rest:
                    if stop.is_set():
                        break
                    loaded.put(self._load(idx, s_load))
            except BaseException as exc:
                failure.append(exc)
                loaded.put(exc)

        def writer():
            try:
                while True:
                    item = to_write.get()
                    if item is None:
                        return
                    idx, n_rows, t_rows, t_txt = self._write(item)
                    written.append((n_rows, t_rows, t_txt))
                    if self.log is not None:
                        self.log("frame %s: %d detections" % (
                            self.dataset.get_filename(idx), n_rows))
            except BaseException as exc:
                failure.append(exc)

        written = []
        threads = [threading.Thread(target=loader, daemon=True),
                   threading.Thread(target=writer, daemon=True)]
        for t in threads:
            t.start()
        self._in_a = deque()
        in_b = deque()
        try:
            for i in range(len(rest)):
                f = loaded.get()
                if isinstance(f, BaseException):
                    raise f
                self._stage_a(f, compute[i % len(compute)])
                self._in_a.append(f)
                if len(self._in_a) > self.in_flight:
                    g = self._in_a.popleft()
                    self._stage_b(g)
                    in_b.append(g)
                if len(in_b) > self.in_flight:
                    self._stage_c(in_b.popleft(), to_write)
                if failure:
                    raise failure[0]
            while self._in_a:
                g = self._in_a.popleft()
                self._stage_b(g)
                in_b.append(g)
            while in_b:
                self._stage_c(in_b.popleft(), to_write)
        finally:
            stop.set()
            to_write.put(None)
            # unblock a loader waiting on a full queue
            while threads[0].is_alive():
                try:
                    loaded.get(timeout=0.05)
                except queue.Empty:
                    pass
            for t in threads:
                t.join()
            for s in streams:
                cur.wait_stream(s)
        if failure:
            raise failure[0]
        for n_rows, t_rows, t_txt in written:
            self._add('kitti rows', t_rows)
            self._add('write txt', t_txt)
            self.stats['rows'] += n_rows
        self.td['sequential fallbacks'] = self.fallbacks
        return 1 + len(rest)
