"""Training step of the hot path (BASELINE config 4; SURVEY.md §8 rows a11, a12
and (e)): forward with saved activations, loss, backward, one flat gradient
all-reduce, SGD -- the device-side equivalent of what `train.py` builds with
TF towers:

  * frame merging by index offset            train.py:135-171  -> batch_data()
  * MultiLayerFastLocalGraphModelV2.loss     models.py:170-311 -> pgnn_loss_fwd_bwd
  * unify_copies re-weighting                train.py:264-288  -> global 1/N, 1/N_valid
  * average_gradients over towers            util/tf_util.py:3-43 -> ONE all-reduce(sum)
    of the flat fp32 gradient buffer (RCCL over xGMI; 5.96 MB for car_auto_T3)
  * GradientDescentOptimizer (and momentum / rmsprop / adam) + staircase decay  train.py:375-405 -> pgnn_sgd_step / pgnn_optimizer_step

Every arithmetic step runs in a HIP kernel behind the C ABI; torch is used for
allocation, views/concatenation and the three residual-style tensor adds of the
backward pass.  The inference path never materialises E x C activations; the
training forward does (the backward needs them).

Weights live in ONE flat device buffer (reference variable order, [in,out]
layouts); gradients in a second one, so data-parallel synchronisation is a
single collective.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .gnn import padded_width
from .weights import init_params, mlp_names, variable_specs

__all__ = ["Trainer", "batch_data", "learning_rate", "fetch_data",
           "train_epochs", "BatchPrefetcher", "allreduce_endpoint_counts",
           "allreduce_endpoint_counts_device", "allreduce_gradients"]


def learning_rate(train_config, step):
    """tf.train.exponential_decay(..., staircase=True), train.py:375-378."""
    p = step // train_config['decay_step'] \
        if train_config.get('is_staircase', True) \
        else step / float(train_config['decay_step'])
    return train_config['initial_lr'] * train_config['decay_factor'] ** p


def batch_data(batch_list):
    """train.py:135-171: merge frames into one disjoint graph by offsetting
    point / centre indices.  Accepts NumPy arrays or torch tensors (same kind
    for all frames); returns the same kind."""
    (n_input_v, n_coords, n_kps, n_edges, n_labels, n_boxes,
     n_valid) = zip(*batch_list)
    is_torch = isinstance(n_input_v[0], torch.Tensor)
    if is_torch and n_input_v[0].is_cuda:
        merged = _batch_data_device(n_input_v, n_coords, n_kps, n_edges,
                                    n_labels, n_boxes, n_valid)
        if merged is not None:
            return merged
    cat = (lambda xs: torch.cat(list(xs), dim=0)) if is_torch else \
        (lambda xs: np.concatenate(list(xs), axis=0))
    level_num = len(n_coords[0])
    kp_out, edge_out = [], []
    for lvl in range(level_num - 1):
        kps, eds = [], []
        point_counter = 0
        center_counter = 0
        for b in range(len(batch_list)):
            kps.append(n_kps[b][lvl] + point_counter)
            e = n_edges[b][lvl]
            if is_torch:
                off = torch.tensor([point_counter, center_counter],
                                   dtype=e.dtype, device=e.device)
            else:
                off = np.array([point_counter, center_counter], dtype=e.dtype)
            eds.append(e + off)
            point_counter += n_coords[b][lvl].shape[0]
            center_counter += n_kps[b][lvl].shape[0]
        kp_out.append(cat(kps))
        merged = cat(eds)
        # frames keep their order and every frame's centres move up by the
        # centres before it: sorted frames merge into a sorted list
        if is_torch and all(getattr(n_edges[b][lvl], "_pgnn_sorted", 0) == 1
                            for b in range(len(batch_list))):
            merged._pgnn_sorted = 1
        edge_out.append(merged)
    coords = [cat([n_coords[b][lvl] for b in range(len(batch_list))])
              for lvl in range(level_num)]
    return (cat(n_input_v), coords, kp_out, edge_out, cat(n_labels),
            cat(n_boxes), cat(n_valid))


def _batch_data_device(n_input_v, n_coords, n_kps, n_edges, n_labels, n_boxes,
                       n_valid):
    """batch_data for CUDA tensors of 4-byte elements as ONE launch
    (pgnn_merge_rows): the merged arrays are allocated, every frame's piece is
    a copy job, index arrays carry their offsets in the job.  None when an
    array is not 4-byte / contiguous-izable that way (the caller concatenates
    with torch then)."""
    ok4 = (torch.float32, torch.int32)
    frames = len(n_input_v)
    levels = len(n_coords[0])
    groups = []       # (list of per-frame tensors, add0s, add1s)
    zeros = [0] * frames
    groups.append((list(n_input_v), zeros, zeros))
    for lvl in range(levels):
        groups.append(([n_coords[b][lvl] for b in range(frames)], zeros, zeros))
    for lvl in range(levels - 1):
        pts = np.cumsum([0] + [int(n_coords[b][lvl].shape[0])
                               for b in range(frames)])[:-1].tolist()
        ctr = np.cumsum([0] + [int(n_kps[b][lvl].shape[0])
                               for b in range(frames)])[:-1].tolist()
        if int(np.max(pts + ctr + [0])) >= 2 ** 31:
            return None
        groups.append(([n_kps[b][lvl] for b in range(frames)], pts, pts))
        groups.append(([n_edges[b][lvl] for b in range(frames)], pts, ctr))
    for arrs in (n_labels, n_boxes, n_valid):
        groups.append((list(arrs), zeros, zeros))
    outs, jobs, keep = [], [], []
    for tensors, a0, a1 in groups:
        t0 = tensors[0]
        if any((not t.is_cuda) or t.dtype not in ok4 or t.dtype != t0.dtype or
               t.shape[1:] != t0.shape[1:] for t in tensors):
            return None
        is_idx = any(a0) or any(a1)
        if is_idx and (t0.dtype != torch.int32 or
                       int(np.prod(t0.shape[1:])) not in (1, 2)):
            return None
        rows = sum(int(t.shape[0]) for t in tensors)
        out = torch.empty((rows,) + tuple(t0.shape[1:]), dtype=t0.dtype,
                          device=t0.device)
        at = 0
        width = int(np.prod(t0.shape[1:])) if t0.dim() > 1 else 1
        for t, x0, x1 in zip(tensors, a0, a1):
            t = t.contiguous()
            keep.append(t)
            n = int(t.numel())
            if n:
                jobs.append((t.data_ptr(), out.data_ptr() + 4 * at, n,
                             int(x0), int(x1) if width == 2 else int(x0)))
            at += n
        outs.append(out)
    arr = (_lib.MergeJob * max(1, len(jobs)))()
    for i, (src, dst, n, x0, x1) in enumerate(jobs):
        arr[i].src, arr[i].dst, arr[i].n_words = src, dst, n
        arr[i].add0, arr[i].add1 = x0, x1
    _lib.check(_lib.load().pgnn_merge_rows(arr, len(jobs), _lib.stream_ptr()),
               "pgnn_merge_rows")
    del keep
    it = iter(outs)
    input_v = next(it)
    coords = [next(it) for _ in range(levels)]
    kp_out, edge_out = [], []
    for lvl in range(levels - 1):
        kp_out.append(next(it))
        merged = next(it)
        if all(getattr(n_edges[b][lvl], "_pgnn_sorted", 0) == 1
               for b in range(frames)):
            merged._pgnn_sorted = 1
        edge_out.append(merged)
    return (input_v, coords, kp_out, edge_out, next(it), next(it), next(it))


def _world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def allreduce_endpoint_counts(n_local, nv_local, device, group=None,
                              comm=None):
    """Global (num_endpoint, num_valid_endpoint) over all ranks: the
    `unify_copies` normalisers of train.py:268-284, as Python floats.  One
    rank: no device work at all.  Several: one tiny all-reduce and its host
    read (`allreduce_endpoint_counts_device` is the form without the read)."""
    if (comm.world if comm is not None else _world(group)) <= 1:
        return float(n_local), float(nv_local)
    c = allreduce_endpoint_counts_device(n_local, nv_local, device, group,
                                         comm=comm)
    c = c.tolist()
    return float(c[0]), float(c[1])


def allreduce_endpoint_counts_device(n_local, nv_local, device, group=None,
                                     force=False, comm=None, out=None):
    """The same, left ON THE DEVICE: a float64 [2] tensor the loss kernel
    reads (pgnn_loss_fwd_bwd_counts), so a multi-rank step has no host wait
    between its forward and its backward pass.  nv_local may be a Python
    number (the data loader knows it) or a 0-d device tensor.  `comm` (a
    comm.Communicator): the reduction is pgnn_allreduce_sum_f64 on the current
    stream (RCCL behind the C ABI) instead of torch.distributed's; `force`:
    issue the collective for a world of one rank too (its fixed cost, and the
    proof that the path runs, on a 1-GPU box); `out`: a float64 [2] device
    buffer to use (the step's result record)."""
    import torch.distributed as dist
    counts = out if out is not None else torch.empty(
        2, dtype=torch.float64, device=device)
    # (fills, not an upload: nothing of the host's pageable memory on the
    # step's stream)
    counts[0:1].fill_(float(n_local))
    if isinstance(nv_local, torch.Tensor):
        counts[1:2].copy_(nv_local.reshape(1))
    else:
        counts[1:2].fill_(float(nv_local))
    if comm is not None:
        if comm.world > 1 or force:
            comm.allreduce_sum(counts)
    elif _world(group) > 1 or (force and dist.is_available() and
                               dist.is_initialized()):
        dist.all_reduce(counts, group=group)
    return counts


def allreduce_gradients(flat_grad, sums=None, group=None, force=False,
                        comm=None):
    """The one data-path collective of a training step: SUM of the flat fp32
    gradient buffer over ranks.  Because every rank already scaled its loss by
    the GLOBAL 1/N and 1/N_valid, the sum equals
    util/tf_util.py:average_gradients of the re-weighted towers.  With `comm`
    (a comm.Communicator) it is pgnn_allreduce_step -- gradient and loss sums
    as ONE RCCL group on the current stream, behind the C ABI; without, a
    torch.distributed all_reduce (`nccl` = RCCL on GPUs, `gloo` in the CPU
    tests).  `force`: also for a world of one rank."""
    if comm is not None:
        if comm.world > 1 or force:
            comm.allreduce_step(flat_grad, sums)
        return flat_grad
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(flat_grad, group=group)
        if sums is not None:
            dist.all_reduce(sums, group=group)
    return flat_grad


_ASSIGN = {'yaw': 'assign_classaware_label_to_points',
           'Car': 'assign_classaware_car_label_to_points',
           'Pedestrian_and_Cyclist':
               'assign_classaware_ped_and_cyc_label_to_points'}


def fetch_data(dataset, frame_idx, config, train_config, aug_fn=None,
               graph_hints=None):
    """train.py:78-133 with every per-point step on the device: crop to the
    image, augmentations, training-mode graph, label assignment, box
    encoding.  Returns the 7-tuple `batch_data` / `Trainer.train_step` take
    (device tensors).

    Order of casts = the reference's: the augmented cloud is float64 from its
    first `xyz.dot(R.T)` on, and graph generation (voxel keys, radius
    predicate), label assignment and box encoding all see those float64
    coordinates (train.py:88-122); float32 / int32 happen last
    (train.py:123-130).

    `graph_hints` (a graph_gen.CountHints kept by the caller across frames):
    the graph -- fan-in cap included -- is built in capacity form and its sizes
    are read ONCE per frame instead of once per keypoint / count / cap stage
    (same tensors for the same NumPy RNG state)."""
    from . import box_encoding, graph_gen, preprocess
    from .run import _input_features
    points = dataset.get_cam_points_in_image_with_rgb(
        frame_idx, config['downsample_by_voxel_size'])
    labels = dataset.get_label(frame_idx)
    if 'crop_aug' in train_config:
        raise NotImplementedError("crop_aug: no shipped train config uses it")
    if aug_fn is None:
        aug_fn = preprocess.get_data_aug(
            train_config.get('data_aug_configs', []))
    points, labels = aug_fn(points, labels)
    if graph_hints is not None and isinstance(points.xyz, torch.Tensor) and \
            config['graph_gen_method'] == 'multi_level_local_graph_v3':
        coords, kps, edges = graph_gen.gen_multi_level_local_graph_v3_one_read(
            points.xyz, graph_hints, **config['graph_gen_kwargs'])
    else:
        fn = graph_gen.get_graph_generate_fn(config['graph_gen_method'])
        coords, kps, edges = fn(points.xyz, **config['graph_gen_kwargs'])
    input_v = _input_features(config, points).contiguous()
    level = config['model_kwargs']['layer_configs'][-1]['graph_level']
    last_xyz = coords[level + 1]
    cls_labels, boxes_3d, valid, label_map = getattr(
        dataset, _ASSIGN[config['label_method']])(
        labels, last_xyz,
        expend_factor=train_config.get('expend_factor', (1.0, 1.0, 1.0)))
    encoded = box_encoding.get_box_encoding_fn(config['box_encoding_method'])(
        cls_labels, last_xyz, boxes_3d, label_map)
    # train.py:123-130
    coords = [c.to(torch.float32) for c in coords]
    return (input_v.to(torch.float32), coords, kps, edges, cls_labels, encoded,
            valid)


def train_epochs(dataset, config, train_config, trainer=None, max_epoch=None,
                 log=None, process_group=None):
    """The epoch loop of train.py:500-650 for THIS rank (one process per GPU;
    the reference drives NUM_GPU towers from one process): resume from
    `train_dir`, a fresh random frame order per epoch (drawn by rank 0), each
    rank fetching its `batch_size / world` frames of every batch, train_step,
    the streaming metrics and the reference's report lines, a checkpoint +
    config files every `save_every_epoch` epochs, at `max_steps` and at the
    end.  Returns (trainer, last results dict)."""
    import os
    import time
    import torch.distributed as dist
    from . import configs as configs_mod, metrics as metrics_mod, preprocess
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(process_group), \
            dist.get_rank(process_group)
    if trainer is None:
        # several ranks on GPUs of their own (`nccl` group): the step's
        # collectives go through the C ABI's RCCL communicator, its id handed
        # out over the group; any other group (gloo) carries them itself
        comm = None
        if world > 1 and dist.get_backend(process_group) == 'nccl':
            from .comm import Communicator
            comm = Communicator.from_torch(process_group)
        trainer = Trainer(config, train_config, process_group=process_group,
                          comm=comm)
        if comm is not None:
            # the reference's towers share ONE set of variables
            # (train.py:225-227): rank 0's initial weights on every rank
            comm.broadcast(trainer.flat, 0)
            trainer.repack()
    train_dir = train_config['train_dir']
    if os.path.isdir(train_dir) and any(
            f.endswith('.index') for f in os.listdir(train_dir)):
        trainer.load_checkpoint(train_dir)
    aug_fn = preprocess.get_data_aug(train_config.get('data_aug_configs', []))
    n_samples = train_config.get('NUM_TEST_SAMPLE', -1)
    if n_samples is None or n_samples < 0:
        n_samples = dataset.num_files
    batch_size = int(train_config.get('batch_size', 1))
    per_rank = batch_size // world
    assert per_rank >= 1, "batch_size smaller than the number of ranks"
    meter = metrics_mod.StreamingMetrics(config['num_classes'],
                                         device=trainer.device)
    from . import graph_gen
    graph_hints = graph_gen.CountHints()
    if max_epoch is None:
        max_epoch = train_config['max_epoch']

    def save():
        if rank != 0:
            return
        trainer.save_checkpoint(train_dir)
        configs_mod.save_config(os.path.join(
            train_dir, train_config.get('config_path', 'config')), config)
        configs_mod.save_train_config(os.path.join(train_dir, 'train_config'),
                                      train_config)

    results = {}
    first = (trainer.global_step * batch_size) // max(n_samples, 1)
    for epoch_idx in range(first, max_epoch):
        meter.reset()
        start = time.time()
        order = torch.from_numpy(np.random.permutation(n_samples))
        if world > 1:
            order = order.to(trainer.device)
            dist.broadcast(order, 0, group=process_group)
            order = order.cpu()
        order = order.numpy()
        def make_batch(b0):
            mine = order[b0 + rank * per_rank:b0 + (rank + 1) * per_rank]
            return batch_data([fetch_data(dataset, int(i), config,
                                          train_config, aug_fn, graph_hints)
                               for i in mine])
        # the epoch's batches come from a loader thread, two ahead (the
        # reference: a 16-process pool, train.py:430-440); the epoch's frame
        # order was drawn above, every later draw happens in that thread
        loader = BatchPrefetcher(
            make_batch, range(0, n_samples - batch_size + 1, batch_size),
            depth=int(train_config.get('prefetch_batches', 2)),
            device=trainer.device)
        for batch in loader:
            if train_config.get('is_pseudo_batch', False):
                results, _ = trainer.pseudo_batch_step(batch)
            else:
                results = trainer.train_step(batch)
            results['total_loss'] = results['cls_loss'] + \
                results['loc_loss'] + results['reg_loss']
            probs = torch.softmax(
                trainer.last_logits[:, :config['num_classes']], dim=1)
            results.update(meter.update(probs, batch[4], results))
            results['step'] = trainer.global_step
            max_steps = train_config.get('max_steps', -1)
            if max_steps and max_steps > 0 and \
                    trainer.global_step >= max_steps:
                loader.close()
                save()
                return trainer, results
        if log is not None and results:
            log('STEP: %d, epoch_idx: %d, lr: %f, time cost: %f' % (
                results['step'], epoch_idx, results['learning_rate'],
                time.time() - start))
            log('cls:%f, loc:%f, reg:%f, loss: %f' % (
                results['cls_loss'], results['loc_loss'], results['reg_loss'],
                results['total_loss']))
            log(meter.format(results))
        if (epoch_idx + 1) % train_config.get('save_every_epoch', 1) == 0:
            save()
    save()
    return trainer, results


class BatchPrefetcher(object):
    """The data side of the training loop off the stepping thread: what the
    reference does with a 16-process loader pool (train.py:430-440,
    `…train_train_config` NUM_LOADERS) is ONE thread here, because the
    per-point work of a batch -- crop, augmentation, training graph, labels,
    box encoding, frame merge -- runs on the device: the thread only enqueues
    it, on a stream of its own, and takes the batch's few size reads there,
    while the stepping thread enqueues forward / backward / update of the
    batch before.  (With the build on the stepping thread the 2-frame step
    was bound by the HOST: 3.07 ms against 2.75 ms for the device work.)

        for batch in BatchPrefetcher(make_batch, range(n_steps)):
            trainer.train_step(batch)

    make(i) -> a batch (any object; CUDA tensors inside tuples / lists are
    found) is called in the loader thread with the loader stream current, in
    the order of `items`, `depth` batches ahead.  Every random draw of the
    loader (NumPy's global state: graph seeds, augmentations) therefore
    happens in one thread and in order.  Iteration makes the consumer's
    current stream wait for the batch and hands the tensors' ownership over
    (record_stream)."""

    def __init__(self, make, items, depth=2, device=None, stream=None):
        import queue
        import threading
        from .engine import concurrent_streams
        self.dev = device or torch.device("cuda", torch.cuda.current_device())
        self.stream = stream or concurrent_streams(1, self.dev)[0]
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._items = list(items)
        self._make = make
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.dev)
            for it in self._items:
                if self._stop.is_set():
                    return
                with torch.cuda.stream(self.stream):
                    batch = self._make(it)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self._q.put((batch, ev))
            self._q.put(None)
        except BaseException as exc:      # re-raised in the consumer
            self._q.put(exc)

    @staticmethod
    def _tensors(obj):
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                yield obj
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                for t in BatchPrefetcher._tensors(o):
                    yield t

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        batch, ev = item
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(ev)
        for t in self._tensors(batch):
            t.record_stream(cur)
        return batch

    def close(self):
        """Stop early (the loader finishes the batch it is building)."""
        import queue
        self._stop.set()
        while self._thread.is_alive():
            try:
                self._q.get(timeout=0.05)
            except queue.Empty:
                pass
        self._thread.join()


class StepResult(object):
    """The numbers of one training step ([sum CE, sum loc, -, -, L1 of the
    weights, (global n, n_valid)] float64), on their way to pinned host memory
    on the step's stream.  get() waits for that copy -- not for anything queued
    after it -- and returns the loss dict of models.py:308-311."""

    def __init__(self, trainer, dev_values, counts, lr):
        self.cls_w, self.loc_w = trainer.cls_w, trainer.loc_w
        self.l1_scale = trainer.l1_scale
        self.counts, self.lr = counts, lr
        self.host = torch.empty(dev_values.shape, dtype=dev_values.dtype,
                                pin_memory=True)
        self.host.copy_(dev_values, non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record()
        self._dev = dev_values      # alive until the copy has run
        self._out = None

    def get(self):
        if self._out is None:
            self.done.synchronize()
            self._dev = None
            v = self.host.tolist()
            n_total, nv_total = self.counts if self.counts is not None \
                else (v[5], v[6])
            self._out = {
                'cls_loss': self.cls_w * float(v[0]) / max(n_total, 1.0),
                'loc_loss': (self.loc_w * float(v[1]) / nv_total)
                if nv_total > 0 else 0.0,
                'reg_loss': self.l1_scale * float(v[4]),
                'num_endpoint': int(n_total), 'num_valid_endpoint': nv_total,
                'learning_rate': self.lr,
            }
        return self._out


class _Fc(object):
    """One fully connected layer of the flat parameter buffer."""

    def __init__(self, name, k_in, n_out, w, b, gw, gb):
        self.name, self.k_in, self.n_out = name, k_in, n_out
        self.w, self.b, self.gw, self.gb = w, b, gw, gb
        self.packed = None    # forward image  (k_in -> n_out, bias)
        self.packed_t = None  # backward image (n_out -> k_in, no bias)
        self.wt = None        # plain W^T [n_out, pad16(k_in)] (sparse adjoint)


# train.py:380-391: name -> (pgnn_optimizer_step kind, TF 1.x default kwargs,
# TF slot names in a checkpoint: '<variable>/<slot>')
_OPTIMIZERS = {
    'sgd': (0, {}, ()),
    'momentum': (1, {'momentum': 0.9, 'use_nesterov': False}, ('Momentum',)),
    'rmsprop': (2, {'momentum': 0.9, 'decay': 0.9, 'epsilon': 1.0,
                    'centered': False}, ('RMSProp', 'RMSProp_1')),
    'adam': (3, {'beta1': 0.9, 'beta2': 0.999, 'epsilon': 1e-8},
             ('Adam', 'Adam_1')),
}


def check_trainable_kinds(config):
    """gnn.py:17-32: the training step differentiates act = ReLU, normalizer =
    NONE (every shipped config).  A config that asks for batch statistics or
    another activation is refused -- it would otherwise be trained as if they
    were absent."""
    for lc in config['model_kwargs']['layer_configs']:
        for key, val in (lc.get('kwargs') or {}).items():
            if key.endswith('normalization_type') and \
                    val not in ('NONE', None):
                raise NotImplementedError(
                    "%s: %s = %r has no training path (batch statistics are "
                    "not differentiated)" % (lc['scope'], key, val))
            if key.endswith('activation_type') and val != 'ReLU':
                raise NotImplementedError(
                    "%s: %s = %r has no training path" % (lc['scope'], key, val))


class Trainer(object):
    def __init__(self, config, train_config=None, params=None, seed=0,
                 device=None, box_encoding_len=7, process_group=None,
                 comm=None, force_collective=False):
        """process_group: a torch.distributed group whose all_reduce carries the
        step's collectives; comm: a comm.Communicator instead -- RCCL behind
        the C ABI (pgnn_allreduce_*), enqueued by the native step itself
        (pgnn_trainer_backward_sync); force_collective: issue the collectives
        for a world of ONE rank too (they are skipped otherwise)."""
        self.config = config
        self.train_config = train_config or {
            'initial_lr': 0.125, 'decay_step': 400000, 'decay_factor': 0.1,
            'optimizer': 'sgd', 'unify_copies': True}
        # train.py:380-391: optimizer class + default kwargs, overridden by
        # train_config['optimizer_kwargs']
        opt = self.train_config.get('optimizer', 'sgd')
        if opt not in _OPTIMIZERS:
            raise ValueError("optimizer %r (train.py:380-385 knows %s)"
                             % (opt, sorted(_OPTIMIZERS)))
        self.optimizer = opt
        self.opt_kwargs = dict(_OPTIMIZERS[opt][1])
        self.opt_kwargs.update(self.train_config.get('optimizer_kwargs') or {})
        unknown = set(self.opt_kwargs) - set(_OPTIMIZERS[opt][1])
        if unknown or self.opt_kwargs.get('use_nesterov') or \
                self.opt_kwargs.get('centered'):
            raise NotImplementedError(
                "optimizer_kwargs %s of %r" % (sorted(self.opt_kwargs), opt))
        if not self.train_config.get('unify_copies', True):
            # train.py:264-288: without it every tower normalises by its OWN
            # endpoint counts; the step below always uses the global ones
            raise NotImplementedError(
                "unify_copies=False (every shipped train config sets it True)")
        # train.py:559-575 (is_pseudo_batch): gradients of `pseudo_batch_factor`
        # batches are summed before one optimizer step -- pseudo_batch_step().
        # train.py:174-182 (COPY_PER_GPU): several towers per GPU.  With
        # unify_copies (required above) every tower's loss is re-weighted to the
        # GLOBAL per-vertex means (train.py:264-288), so how a batch's frames
        # are grouped into towers does not change its gradient: one process per
        # GPU takes its batch_size / world frames as one merged batch whatever
        # COPY_PER_GPU says.
        self._pseudo_sum = None
        self._pseudo_n = 0
        self._pseudo_ctr = 0
        if int(self.train_config.get('COPY_PER_GPU', 1)) < 1:
            raise ValueError("COPY_PER_GPU must be >= 1")
        if self.train_config.get('is_pseudo_batch', False) and \
                int(self.train_config.get('pseudo_batch_factor', 0)) < 1:
            raise ValueError("is_pseudo_batch needs pseudo_batch_factor >= 1")
        if not torch.cuda.is_available():
            raise _lib.PointGnnHipError("Trainer needs a GPU (no CPU fallback)")
        self.device = device or torch.device("cuda",
                                             torch.cuda.current_device())
        self.lib = _lib.load()
        # the last per-edge layer + scatter-max of every stage is
        # differentiated sparsely (pgnn_segmax_fc_bwd_f32); False = the dense
        # adjoint primitives (tests compare the two)
        self.sparse_adjoint = True
        self.box_len = box_encoding_len
        self.nc = config['num_classes']
        self.pg = process_group
        self.comm = comm
        self.force_collective = bool(force_collective)
        self.global_step = 0
        # optimizer steps taken with THIS optimizer's slots (Adam's bias
        # correction counts them, not global_step: resuming another
        # optimizer's checkpoint starts fresh slots and fresh beta powers,
        # as TF's beta1_power / beta2_power variables would)
        self.opt_step = 0
        mk = config['model_kwargs']
        if mk.get('regularizer_type') not in (None, 'l1'):
            raise NotImplementedError("regularizer %r" % mk['regularizer_type'])
        self.l1_scale = float(mk['regularizer_kwargs']['scale']) \
            if mk.get('regularizer_type') == 'l1' else 0.0
        check_trainable_kinds(config)
        from .models import cls_loss_kind, loss_top_k
        # models.py:198-208: the loss entries may be dicts keyed by mode
        lcfg = {key: (val['train'] if isinstance(val, dict) and 'train' in val
                      and key in ('cls_loss_type', 'cls_loss_kwargs',
                                  'loc_loss_type', 'loc_loss_kwargs',
                                  'cls_loss_weight', 'loc_loss_weight')
                      else val)
                for key, val in config['loss'].items()}
        self.cls_w = float(lcfg['cls_loss_weight'])
        self.loc_w = float(lcfg['loc_loss_weight'])
        self.cls_kind = cls_loss_kind(lcfg.get('cls_loss_type', 'softmax'),
                                      lcfg.get('cls_loss_kwargs'))
        # 'top_k_softmax' / 'top_k_huber_loss' (models.py:222-228, 266-291):
        # the loss of the k worst vertices of THIS rank's batch (a tower's, in
        # the reference)
        self.cls_topk, self.loc_topk = loss_top_k(
            lcfg.get('cls_loss_type', 'softmax'), lcfg.get('cls_loss_kwargs'),
            lcfg.get('loc_loss_type', 'huber_loss'),
            lcfg.get('loc_loss_kwargs'))
        self._class_loc_w_host = (lcfg.get('loc_loss_kwargs') or {}
                                  ).get('classwise_loc_loss_weight')
        self._class_loc_w = None
        # ---- flat parameter / gradient buffers --------------------------
        self.specs = variable_specs(config, box_encoding_len=box_encoding_len)
        if params is None:
            params = init_params(config, seed=seed)
        total = int(sum(int(np.prod(s)) for _, s in self.specs))
        flat = np.empty(total, np.float32)
        mask = np.zeros(total, np.float32)
        self.offsets = {}
        off = 0
        for name, shape in self.specs:
            n = int(np.prod(shape))
            flat[off:off + n] = np.asarray(params[name], np.float32).reshape(-1)
            if name.endswith('/weights'):
                mask[off:off + n] = 1.0
            self.offsets[name] = (off, shape)
            off += n
        self.flat = torch.from_numpy(flat).to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self.is_weight = torch.from_numpy(mask).to(self.device)
        # optimizer slots (TF: zeros, RMSProp's mean square ones), parameter layout
        self.slots = [torch.zeros_like(self.flat)
                      for _ in _OPTIMIZERS[self.optimizer][2]]
        if self.optimizer == 'rmsprop':
            self.slots[0].fill_(1.0)
        self.fc = {}
        for name, shape in self.specs:
            if not name.endswith('/weights'):
                continue
            base = name[:-len('/weights')]
            k, n = shape
            self.fc[base] = _Fc(base, k, n, self._view(self.flat, name),
                                self._view(self.flat, base + '/biases'),
                                self._view(self.grad, name),
                                self._view(self.grad, base + '/biases'))
        self._ws = None
        self._ws2 = None
        self._pack_jobs = None
        # The step runs natively by default (csrc/trainer.hip: forward and
        # backward are one C call each, all launches issued from C++);
        # native = False drives the same primitives from Python (the readable
        # composition below; tests hold the two to the same gradients).
        self.native = True
        n_pool = sum(lc['type'] == 'scatter_max_point_set_pooling'
                     for lc in config['model_kwargs']['layer_configs'])
        if n_pool > 1:
            # csrc/trainer.hip orchestrates ONE pooling stage (every shipped
            # config); a second one (models.py:119-149 is generic) trains
            # through the Python-driven composition of the same primitives
            self.native = False
        self._native = None
        self._native_ws = None
        self._py_images_stale = True
        self._native_images_stale = False
        self.repack()

    # ---- plumbing -----------------------------------------------------------
    def _view(self, flat, name):
        off, shape = self.offsets[name]
        n = int(np.prod(shape))
        return flat[off:off + n].view(*shape)

    def state_dict(self):
        return {name: self._view(self.flat, name).cpu().numpy()
                for name, _ in self.specs}

    def pseudo_batch_step(self, batch, **kw):
        """train.py:559-575 (`is_pseudo_batch`): this batch's gradients are
        computed (all-reduced over ranks) and ADDED to the pending sum; when the
        reference's counter says so -- `batch_ctr % pseudo_batch_factor == 0`,
        counting from 0: the very first batch alone, then every `factor`
        batches -- the sum is applied in one optimizer step.  Every batch's
        gradient includes the regulariser's (it is part of each tower loss,
        train.py:286-288), so the sum carries it once per summed batch.
        Returns (loss dict of this batch, whether a step was applied)."""
        out = self.train_step(batch, apply=False, **kw)
        if self._pseudo_sum is None:
            self._pseudo_sum = torch.zeros_like(self.grad)
        self._pseudo_sum += self.grad
        self._pseudo_n += 1
        factor = int(self.train_config['pseudo_batch_factor'])
        applied = self._pseudo_ctr % factor == 0
        self._pseudo_ctr += 1
        if applied:
            self.grad.copy_(self._pseudo_sum)
            self._apply_gradients(
                learning_rate(self.train_config, self.global_step),
                l1_mult=self._pseudo_n)
            self.repack()
            self.global_step += 1
            self.opt_step += 1
            self._pseudo_sum.zero_()
            self._pseudo_n = 0
        return out, applied

    def _apply_gradients(self, lr, l1_mult=1):
        """optimizer.apply_gradients (train.py:392-405) on the flat buffers;
        `lr` is the decayed learning rate of this step; `l1_mult`: how many
        batches' regulariser gradients the buffer stands for."""
        kind = _OPTIMIZERS[self.optimizer][0]
        kw = self.opt_kwargs
        l1 = self.l1_scale * l1_mult
        if kind == 0:
            _lib.check(self.lib.pgnn_sgd_step(
                _lib.ptr(self.flat), _lib.ptr(self.grad),
                _lib.ptr(self.is_weight), self.flat.numel(),
                ctypes.c_float(lr), ctypes.c_float(1.0),
                ctypes.c_float(l1), self._st()), "pgnn_sgd_step")
            return
        if kind == 1:
            h = (kw['momentum'], 0.0, 0.0)
        elif kind == 2:
            h = (kw['momentum'], kw['decay'], kw['epsilon'])
        else:
            # AdamOptimizer: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), t
            # counting this optimizer's own steps from 1 (its beta*_power
            # variables: opt_step, restored from a checkpoint's beta1_power)
            t = self.opt_step + 1
            lr = lr * np.sqrt(1.0 - kw['beta2'] ** t) / (1.0 - kw['beta1'] ** t)
            h = (kw['beta1'], kw['beta2'], kw['epsilon'])
        _lib.check(self.lib.pgnn_optimizer_step(
            kind, _lib.ptr(self.flat), _lib.ptr(self.grad),
            _lib.ptr(self.is_weight), _lib.ptr(self.slots[0]),
            _lib.ptr(self.slots[1]) if len(self.slots) > 1 else None,
            self.flat.numel(), ctypes.c_float(lr), ctypes.c_float(1.0),
            ctypes.c_float(l1), ctypes.c_float(h[0]),
            ctypes.c_float(h[1]), ctypes.c_float(h[2]), self._st()),
            "pgnn_optimizer_step")

    def optimizer_state_dict(self):
        """The optimizer's slot variables under TF's names ('<variable>/<slot>',
        plus Adam's beta1_power / beta2_power): what tf.train.Saver writes next
        to the weights."""
        out = {}
        for slot, buf in zip(_OPTIMIZERS[self.optimizer][2], self.slots):
            for name, _ in self.specs:
                out[name + '/' + slot] = self._view(buf, name).cpu().numpy()
        if self.optimizer == 'adam':
            t = self.opt_step + 1         # TF keeps beta^(steps taken + 1)
            out['beta1_power'] = np.float32(self.opt_kwargs['beta1'] ** t)
            out['beta2_power'] = np.float32(self.opt_kwargs['beta2'] ** t)
        return out

    def save_checkpoint(self, train_dir):
        """train.py:625-638: TF-bundle checkpoint `model-<global_step>` (weights
        under their TF names + the int32 step `Variable`) that the reference's
        run.py / train.py restore unchanged."""
        from . import tf_bundle
        state = self.state_dict()
        state.update(self.optimizer_state_dict())
        return tf_bundle.save_checkpoint(
            train_dir, state, global_step=self.global_step,
            name=self.train_config.get('checkpoint_path', 'model'))

    def load_checkpoint(self, train_dir):
        """train.py:512-516: resume weights and step from `train_dir`."""
        from . import tf_bundle
        ck = tf_bundle.load_checkpoint(train_dir)
        if 'Variable' in ck:
            self.global_step = int(ck['Variable'])
        self.opt_step = 0
        if self.optimizer == 'adam' and 'beta1_power' in ck:
            # beta1_power = beta1 ^ (steps taken + 1)
            b1 = float(self.opt_kwargs['beta1'])
            p = float(np.asarray(ck['beta1_power']).reshape(-1)[0])
            if 0.0 < p < 1.0 and 0.0 < b1 < 1.0:
                self.opt_step = max(0, int(round(np.log(p) / np.log(b1))) - 1)
        elif self.optimizer != 'adam' and any(
                (name + '/' + slot) in ck for name, _ in self.specs[:1]
                for slot in _OPTIMIZERS[self.optimizer][2]):
            self.opt_step = self.global_step
        for name, _ in self.specs:
            self._view(self.flat, name).copy_(
                torch.from_numpy(np.ascontiguousarray(ck[name], np.float32)))
            # optimizer slots, when the checkpoint has them (one written by
            # another optimizer restores the weights only, like a Saver built
            # for these variables would fail -- here: fresh slots)
            for slot, buf in zip(_OPTIMIZERS[self.optimizer][2], self.slots):
                key = name + '/' + slot
                if key in ck:
                    self._view(buf, name).copy_(torch.from_numpy(
                        np.ascontiguousarray(ck[key], np.float32)))
        self.repack()
        return self

    def grad_dict(self):
        return {name: self._view(self.grad, name).cpu().numpy()
                for name, _ in self.specs}

    def _st(self):
        return _lib.stream_ptr()

    def _sparse_layers(self):
        """Names of the layers whose output feeds a scatter-max (the last
        per-edge layer of every pooling / GNN stage): their adjoint runs
        sparse (pgnn_segmax_fc_bwd_f32) and needs a plain W^T image."""
        out = []
        for lc in self.config['model_kwargs']['layer_configs'][:-1]:
            kw = lc['kwargs']
            if lc['type'] == 'scatter_max_point_set_pooling':
                n = len(kw['point_MLP_depth_list'])
            elif lc['type'] == 'scatter_max_graph_auto_center_net':
                n = len(kw['edge_MLP_depth_list'])
            else:
                continue
            if n >= 2:   # the layer needs a materialised input activation
                out.append(mlp_names(
                    lc['scope'] + '/extract_vertex_features', n)[-1])
        return out

    def _build_pack_jobs(self):
        """Device table of pgnn_pack_fc_many: the fragment images (forward,
        transposed) of every layer, the W^T rows of the Wx block of each first
        edge layer (dx' = dQ Wx^T), plain W^T of the sparse-adjoint layers."""
        lib = self.lib
        sparse = set(self._sparse_layers())
        jobs = []

        def add(w_ptr, b_ptr, dst, k_in, n_out, kind):
            if kind == 2:
                elems = n_out * padded_width(k_in)
            elif kind == 1:
                elems = lib.pgnn_packed_fc_floats(n_out, k_in)
            else:
                elems = lib.pgnn_packed_fc_floats(k_in, n_out)
            jobs.append((w_ptr, b_ptr, dst.data_ptr(), k_in, n_out, kind,
                         (int(elems) + 255) // 256))
        for fc in self.fc.values():
            fc.packed = torch.empty(
                lib.pgnn_packed_fc_floats(fc.k_in, fc.n_out),
                dtype=torch.float32, device=self.device)
            fc.packed_t = torch.empty(
                lib.pgnn_packed_fc_floats(fc.n_out, fc.k_in),
                dtype=torch.float32, device=self.device)
            add(fc.w.data_ptr(), fc.b.data_ptr(), fc.packed, fc.k_in,
                fc.n_out, 0)
            add(fc.w.data_ptr(), 0, fc.packed_t, fc.k_in, fc.n_out, 1)
            if fc.name in sparse:
                fc.wt = torch.empty((fc.n_out, padded_width(fc.k_in)),
                                    dtype=torch.float32, device=self.device)
                add(fc.w.data_ptr(), 0, fc.wt, fc.k_in, fc.n_out, 2)
        # Wx = rows c..c+2 of every first edge layer, transposed image
        self.wx_packed_t = {}
        for lc in self.config['model_kwargs']['layer_configs'][:-1]:
            if lc['type'] != 'scatter_max_graph_auto_center_net':
                continue
            name = mlp_names(lc['scope'] + '/extract_vertex_features', 1)[0]
            w1 = self.fc[name]
            c = w1.k_in - 3
            pk = torch.empty(lib.pgnn_packed_fc_floats(w1.n_out, 3),
                             dtype=torch.float32, device=self.device)
            add(w1.w.data_ptr() + 4 * c * w1.n_out, 0, pk, 3, w1.n_out, 1)
            self.wx_packed_t[name] = pk
        arr = (_lib.PackJob * len(jobs))()
        first = 0
        for i, (w, b, dst, k, n, kind, blocks) in enumerate(jobs):
            arr[i].w, arr[i].b, arr[i].dst = w, b or None, dst
            arr[i].k_in, arr[i].n_out, arr[i].kind = int(k), int(n), int(kind)
            arr[i].first_block = first
            first += blocks
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self._pack_jobs = torch.from_numpy(raw).to(self.device)
        self._pack_n, self._pack_blocks = len(jobs), first

    def repack(self):
        """Refresh every device image derived from the flat buffer (MFMA
        fragment images forward / transposed, plain W^T of the sparse-adjoint
        layers) -- after init and after every SGD step: ONE launch.  Only the
        images of the path in use are refreshed (native step: the handle's;
        Python-driven step: this object's, lazily)."""
        if self.native and self.sparse_adjoint:
            self._py_images_stale = True
            if self._native is not None:
                _lib.check(self.lib.pgnn_trainer_repack(self._native,
                                                        self._st()),
                           "pgnn_trainer_repack")
            # (a handle created later packs in pgnn_trainer_bind)
            self._native_images_stale = False
            return
        # Python-driven step: the native handle's images (if one exists) are
        # now behind the flat buffer; _native_forward repacks before reuse
        self._native_images_stale = self._native is not None
        self._repack_py()

    def _repack_py(self):
        if getattr(self, '_pack_jobs', None) is None:
            self._build_pack_jobs()
        _lib.check(self.lib.pgnn_pack_fc_many(
            _lib.ptr(self._pack_jobs), self._pack_n, self._pack_blocks,
            self._st()), "pgnn_pack_fc_many")
        self._py_images_stale = False

    def _layer_array(self, packed, k_in, n_out, relu):
        arr = (_lib.FcLayer * 1)()
        arr[0].packed = packed.data_ptr()
        arr[0].k_in, arr[0].n_out = int(k_in), int(n_out)
        arr[0].relu_from = 0 if relu else int(n_out)
        return arr

    def _mlp1(self, packed, k_in, n_out, relu, x, nx, x2=None, nx2=0,
              residual=None):
        rows = int(x.shape[0])
        y = torch.empty((rows, padded_width(n_out)), dtype=torch.float32,
                        device=self.device)
        arr = self._layer_array(packed, k_in, n_out, relu)
        _lib.check(self.lib.pgnn_mlp_fwd(
            _lib.ptr(x), x.stride(0), int(nx), _lib.ptr(x2),
            x2.stride(0) if x2 is not None else 0, int(nx2), rows, arr, 1,
            _lib.ptr(residual),
            residual.stride(0) if residual is not None else 0, _lib.ptr(y),
            y.stride(0), self._st()), "pgnn_mlp_fwd")
        return y

    def fc_fwd(self, name, x, relu, x2=None, nx2=0, residual=None):
        fc = self.fc[name]
        return self._mlp1(fc.packed, fc.k_in, fc.n_out, relu, x,
                          fc.k_in - nx2, x2, nx2, residual)

    def _weight_grad(self, x, ld_x, k_in, dz, n_out, gw_ptr, gb_ptr):
        rows = int(dz.shape[0])
        need = self.lib.pgnn_weight_grad_workspace_bytes(k_in, n_out, rows)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8,
                                   device=self.device)
        _lib.check(self.lib.pgnn_weight_grad_f32(
            _lib.ptr(x), ld_x, k_in, _lib.ptr(dz), dz.stride(0), n_out, rows,
            gw_ptr, gb_ptr, 1, _lib.ptr(self._ws), self._ws.numel(),
            self._st()), "pgnn_weight_grad_f32")

    def fc_bwd(self, name, x, y, dy, relu, need_dx=True):
        """dy is consumed (masked in place when relu).  x: the saved input
        [rows, >= k_in] (already concatenated for two-input layers)."""
        fc = self.fc[name]
        if relu:
            _lib.check(self.lib.pgnn_relu_mask_mul(
                _lib.ptr(dy), _lib.ptr(y), dy.numel(), self._st()),
                "pgnn_relu_mask_mul")
        self._weight_grad(x, x.stride(0), fc.k_in, dy, fc.n_out,
                          _lib.ptr(fc.gw), _lib.ptr(fc.gb))
        if not need_dx:
            return None
        return self._mlp1(fc.packed_t, fc.n_out, fc.k_in, False, dy, fc.n_out)

    def _scatter_max(self, data, dst, k):
        out = torch.empty((k, data.shape[1]), dtype=torch.float32,
                          device=self.device)
        _lib.check(self.lib.pgnn_scatter_max_f32(
            _lib.ptr(data), data.stride(0), _lib.ptr(dst), int(data.shape[0]),
            int(data.shape[1]), k, _lib.ptr(out), out.stride(0),
            int(getattr(dst, "_pgnn_sorted", 0)), self._st()),
            "pgnn_scatter_max_f32")
        return out

    def _scatter_max_bwd(self, data, dst, out, gout):
        k, cols = int(out.shape[0]), int(data.shape[1])
        ties = torch.empty(k * cols, dtype=torch.int32, device=self.device)
        gdata = torch.empty_like(data)
        _lib.check(self.lib.pgnn_scatter_max_bwd_f32(
            _lib.ptr(data), data.stride(0), _lib.ptr(dst), int(data.shape[0]),
            cols, k, _lib.ptr(out), out.stride(0), _lib.ptr(gout),
            gout.stride(0), _lib.ptr(ties), _lib.ptr(gdata), gdata.stride(0),
            int(getattr(dst, "_pgnn_sorted", 0)), self._st()),
            "pgnn_scatter_max_bwd_f32")
        return gdata

    def _segmax_fc_bwd(self, name, y, dst, out, gout, x, need_dx=True,
                       mask_x=True):
        """Adjoint of out = scatter_max(y), y = ReLU(x W + b) for layer `name`
        (pgnn_segmax_fc_bwd_f32): accumulates dW / db, returns dX (masked by
        x > 0) or None."""
        fc = self.fc[name]
        rows, k = int(y.shape[0]), int(out.shape[0])
        need = self.lib.pgnn_segmax_fc_bwd_workspace_bytes(rows, fc.n_out, k,
                                                           fc.k_in)
        if self._ws2 is None or self._ws2.numel() < need:
            self._ws2 = torch.empty(int(need), dtype=torch.uint8,
                                    device=self.device)
        dx = None
        if need_dx:
            dx = torch.empty((rows, int(x.shape[1])), dtype=torch.float32,
                             device=self.device)
        _lib.check(self.lib.pgnn_segmax_fc_bwd_f32(
            _lib.ptr(y), y.stride(0), _lib.ptr(dst), rows, fc.n_out, k,
            _lib.ptr(out), out.stride(0), _lib.ptr(gout), gout.stride(0),
            _lib.ptr(x), x.stride(0), fc.k_in, _lib.ptr(fc.wt),
            fc.wt.stride(0), _lib.ptr(dx),
            dx.stride(0) if dx is not None else 0,
            int(x.shape[1]) if dx is not None else 0, 1 if mask_x else 0,
            _lib.ptr(fc.gw), _lib.ptr(fc.gb), _lib.ptr(self._ws2),
            self._ws2.numel(), self._st()), "pgnn_segmax_fc_bwd_f32")
        return dx

    @staticmethod
    def _sorted_dst(edges):
        """dst column, tagged with whether the list is grouped by ascending
        dst: graph_gen / batch_data lists are (tagged there), a foreign list is
        checked once (gnn._edges_sorted_flag) and takes the all-atomic path of
        the scatter-max kernels when it is not."""
        from . import gnn
        d = edges[:, 1].contiguous()
        d._pgnn_sorted = int(gnn._edges_sorted_flag(edges))
        return d

    # ---- native step (csrc/trainer.hip) -----------------------------------------------
    def _fc_ref(self, dst, name):
        off_w, (k, n) = self.offsets[name + '/weights']
        off_b, _ = self.offsets[name + '/biases']
        dst.w_off, dst.b_off, dst.k_in, dst.n_out = off_w, off_b, int(k), int(n)

    def _native_handle(self):
        """pgnn_trainer_create + bind for this model (once)."""
        if self._native is not None:
            return self._native
        lib = self.lib
        m = _lib.TrainModel()
        lcs = self.config['model_kwargs']['layer_configs']
        m.n_stages = len(lcs) - 1
        m.num_classes, m.box_len = self.nc, self.box_len
        m.n_params = int(self.flat.numel())
        for si, lc in enumerate(lcs[:-1]):
            st, kw, scope = m.stages[si], lc['kwargs'], lc['scope']
            st.graph_level = int(lc['graph_level'])
            if lc['type'] == 'scatter_max_point_set_pooling':
                st.kind = 0
                a = mlp_names(scope + '/extract_vertex_features',
                              len(kw['point_MLP_depth_list']))
                b = mlp_names(scope + '/combined_features',
                              len(kw['output_MLP_depth_list']))
                c = []
            elif lc['type'] == 'scatter_max_graph_auto_center_net':
                st.kind = 1
                a = mlp_names(scope + '/extract_vertex_features',
                              len(kw['edge_MLP_depth_list']))
                b = mlp_names(scope + '/combined_features',
                              len(kw['update_MLP_depth_list']))
                c = mlp_names(scope, len(kw['auto_offset_MLP_depth_list'])) \
                    if kw['auto_offset'] else []
            else:
                raise NotImplementedError(lc['type'])
            st.n_a, st.n_b, st.n_c = len(a), len(b), len(c)
            for arr, names in ((st.a, a), (st.b, b), (st.c, c)):
                for i, n in enumerate(names):
                    self._fc_ref(arr[i], n)
        pc = lcs[-1]
        if pc['type'] != 'classaware_predictor':
            raise NotImplementedError(pc['type'])
        ps = pc['scope'] + '/predictor'
        for i, n in enumerate(mlp_names(ps + '/cls', 2)):
            self._fc_ref(m.cls[i], n)
        for j in range(self.nc):
            for i, n in enumerate(mlp_names(ps + '/loc/cls_%d' % j, 3)):
                self._fc_ref(m.loc[j][i], n)
        h = ctypes.c_void_p()
        rc = lib.pgnn_trainer_create(ctypes.byref(m), ctypes.byref(h))
        if rc == _lib.E_UNSUPPORTED:
            # layer shapes outside what csrc/trainer.hip orchestrates: the
            # Python-driven composition of the same primitives takes over
            import warnings
            warnings.warn("native training step unavailable for this model "
                          "(%s); using the Python-driven step" % (
                              (lib.pgnn_last_error() or b'?').decode(),),
                          RuntimeWarning)
            self.native = False
            return None
        _lib.check(rc, "pgnn_trainer_create")
        self._native_images = torch.empty(
            int(lib.pgnn_trainer_images_bytes(h)), dtype=torch.uint8,
            device=self.device)
        _lib.check(lib.pgnn_trainer_bind(
            h, _lib.ptr(self.flat), _lib.ptr(self.grad),
            _lib.ptr(self._native_images), self._native_images.numel(),
            self._st()), "pgnn_trainer_bind")
        self._native = h
        return h

    def __del__(self):
        h, self._native = getattr(self, '_native', None), None
        if h is not None:
            try:
                self.lib.pgnn_trainer_destroy(h)
            except Exception:
                pass

    def _native_batch(self, input_v, coords, kps, edges):
        """pgnn_train_batch for these tensors (kept alive by the caller)."""
        dev = self.device
        f32 = lambda t: torch.as_tensor(t).to(
            device=dev, dtype=torch.float32).contiguous()
        i32 = lambda t: torch.as_tensor(t).to(
            device=dev, dtype=torch.int32).contiguous()
        input_v = f32(input_v)
        coords = [f32(c) for c in coords]
        kps = [i32(torch.as_tensor(k).reshape(-1)) for k in kps]
        from . import gnn
        flags = [int(gnn._edges_sorted_flag(torch.as_tensor(e)))
                 for e in edges]
        edges = [i32(e) for e in edges]
        b = _lib.TrainBatch()
        b.input_v, b.n_feat = input_v.data_ptr(), int(input_v.shape[1])
        b.n_levels = len(edges)
        for l, c in enumerate(coords):
            b.n_vertices[l], b.coords[l] = int(c.shape[0]), c.data_ptr()
        for l in range(len(edges)):
            b.keypoints[l] = kps[l].data_ptr()
            b.edges[l], b.n_edges[l] = edges[l].data_ptr(), int(edges[l].shape[0])
            b.edges_sorted[l] = flags[l]
        return b, (input_v, coords, kps, edges)

    def _native_forward(self, input_v, coords, kps, edges):
        lib = self.lib
        h = self._native_handle()
        if h is None:     # unsupported shapes: Trainer.native was switched off
            self._py_images_stale = True
            return self.forward(input_v, coords, kps, edges)
        if self._native_images_stale:
            # weights were updated while native = False (repack() refreshed
            # only the Python-side images then)
            _lib.check(lib.pgnn_trainer_repack(h, self._st()),
                       "pgnn_trainer_repack")
            self._native_images_stale = False
        batch, keep = self._native_batch(input_v, coords, kps, edges)
        need = int(lib.pgnn_trainer_workspace_bytes(h, ctypes.byref(batch)))
        if need == 0:
            raise _lib.PointGnnHipError("pgnn_trainer_workspace_bytes: %s" % (
                (lib.pgnn_last_error() or b'?').decode()))
        if self._native_ws is None or self._native_ws.numel() < need:
            self._native_ws = torch.empty(
                (int(need * 1.25) + 255) // 256 * 256, dtype=torch.uint8,
                device=self.device)
        lg, ld, pb = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_void_p()
        _lib.check(lib.pgnn_trainer_forward(
            h, ctypes.byref(batch), _lib.ptr(self._native_ws),
            self._native_ws.numel(), ctypes.byref(lg), ctypes.byref(ld),
            ctypes.byref(pb), self._st()), "pgnn_trainer_forward")
        k = int(batch.n_vertices[batch.n_levels])
        # views INTO the workspace (valid until the next forward)
        base = self._native_ws.data_ptr()
        flat = self._native_ws.view(torch.float32)
        o_lg = (lg.value - base) // 4
        o_pb = (pb.value - base) // 4
        logits = flat[o_lg:o_lg + k * ld.value].view(k, ld.value)
        pred = flat[o_pb:o_pb + k * self.nc * self.box_len].view(
            k, self.nc, self.box_len)
        self._saved = ('native', batch, keep)
        self._h_width = padded_width(self.fc[mlp_names(
            self.config['model_kwargs']['layer_configs'][-1]['scope'] +
            '/predictor/cls', 2)[0]].k_in)
        return logits[:, :self.nc], pred

    def _native_backward(self, dlogits, dpred, sync_sums=None):
        _, batch, keep = self._saved
        dl = dlogits.contiguous()
        dp = dpred.contiguous()
        sync = sync_sums is not None and self.comm is not None
        _lib.check(self.lib.pgnn_trainer_backward_sync(
            self._native_handle(), ctypes.byref(batch),
            _lib.ptr(self._native_ws), self._native_ws.numel(), _lib.ptr(dl),
            _lib.ptr(dp), self.comm.handle if sync else None,
            _lib.ptr(sync_sums) if sync else None,
            sync_sums.numel() if sync else 0, self._st()),
            "pgnn_trainer_backward_sync")
        self._saved = None
        return sync

    # ---- forward with saved activations ------------------------------------------
    def forward(self, input_v, coords, kps, edges):
        """Returns (logits [K,nc], pred_box [K,nc,L]) and keeps what backward
        needs in self._saved."""
        if self.native and self.sparse_adjoint:
            return self._native_forward(input_v, coords, kps, edges)
        if getattr(self, '_py_images_stale', True):
            self._repack_py()
        lib, st = self.lib, self._st()
        dev = self.device
        f32 = lambda t: torch.as_tensor(t).to(
            device=dev, dtype=torch.float32).contiguous()
        i32 = lambda t: torch.as_tensor(t).to(
            device=dev, dtype=torch.int32).contiguous()
        input_v = f32(input_v)
        coords = [f32(c) for c in coords]
        kps = [i32(k.reshape(-1)) for k in kps]
        edges = [i32(e) for e in edges]
        saved = []
        h = None
        for lc in self.config['model_kwargs']['layer_configs'][:-1]:
            scope, kw, lvl = lc['scope'], lc['kwargs'], lc['graph_level']
            if lc['type'] == 'scatter_max_point_set_pooling':
                e = edges[lvl]
                k = int(kps[lvl].shape[0])
                names = mlp_names(scope + '/extract_vertex_features',
                                  len(kw['point_MLP_depth_list']))
                upper = None
                if h is None:
                    feat = torch.empty((int(e.shape[0]), 16),
                                       dtype=torch.float32, device=dev)
                    _lib.check(lib.pgnn_pool_features_fwd(
                        _lib.ptr(input_v), int(input_v.shape[1]),
                        _lib.ptr(coords[lvl]), _lib.ptr(kps[lvl]), _lib.ptr(e),
                        int(e.shape[0]), _lib.ptr(feat), st),
                        "pgnn_pool_features_fwd")
                else:
                    # a pooling level above the first (models.py:119-149 is
                    # generic; no shipped config has one): the edge rows
                    # [h[src], x[src] - x[kp[dst]]] of the previous level's
                    # features are materialised, as gnn.PointSetPooling's
                    # wide-feature path does
                    n_feat = self.fc[names[0]].k_in - 3
                    feat = torch.empty((int(e.shape[0]),
                                        padded_width(n_feat + 3)),
                                       dtype=torch.float32, device=dev)
                    _lib.check(lib.pgnn_pool_features_wide_fwd(
                        _lib.ptr(h), h.stride(0), n_feat,
                        _lib.ptr(coords[lvl]), _lib.ptr(kps[lvl]), _lib.ptr(e),
                        int(e.shape[0]), _lib.ptr(feat), feat.stride(0), st),
                        "pgnn_pool_features_wide_fwd")
                    upper = (e, n_feat, int(h.shape[0]), int(h.shape[1]))
                acts = [feat]
                for n in names:
                    acts.append(self.fc_fwd(n, acts[-1], True))
                dst = self._sorted_dst(e)
                agg = self._scatter_max(acts[-1], dst, k)
                onames = mlp_names(scope + '/combined_features',
                                   len(kw['output_MLP_depth_list']))
                oacts = [agg]
                for n in onames:
                    oacts.append(self.fc_fwd(n, oacts[-1], True))
                h = oacts[-1]
                saved.append(('pool', names, acts, dst, agg, onames, oacts,
                              upper))
            elif lc['type'] == 'scatter_max_graph_auto_center_net':
                e = edges[lvl]
                x = coords[lvl]
                k = int(h.shape[0])
                enames = mlp_names(scope + '/extract_vertex_features',
                                   len(kw['edge_MLP_depth_list']))
                w1 = self.fc[enames[0]]
                c = w1.k_in - 3
                wq = padded_width(w1.n_out)
                off_names, off_acts, delta = None, None, None
                if kw['auto_offset']:
                    off_names = mlp_names(
                        scope, len(kw['auto_offset_MLP_depth_list']))
                    off_acts = [h]
                    for i, n in enumerate(off_names):
                        off_acts.append(self.fc_fwd(
                            n, off_acts[-1], i + 1 < len(off_names)))
                    delta = off_acts[-1]
                wx = torch.zeros((3, wq), dtype=torch.float32, device=dev)
                wx[:, :w1.n_out] = w1.w[c:c + 3]
                xo = torch.empty_like(x)
                q = torch.empty((k, wq), dtype=torch.float32, device=dev)
                _lib.check(lib.pgnn_offset_apply(
                    _lib.ptr(x), _lib.ptr(delta),
                    delta.stride(0) if delta is not None else 0, k,
                    _lib.ptr(wx), _lib.ptr(xo), _lib.ptr(q), wq, st),
                    "pgnn_offset_apply")
                hx = torch.zeros((k, padded_width(c + 3)),
                                 dtype=torch.float32, device=dev)
                hx[:, :c] = h[:, :c]
                hx[:, c:c + 3] = x
                p = self.fc_fwd(enames[0], hx, False)
                h1 = torch.empty((int(e.shape[0]), wq), dtype=torch.float32,
                                 device=dev)
                _lib.check(lib.pgnn_edge_hidden_fwd(
                    _lib.ptr(p), _lib.ptr(q), wq, _lib.ptr(e),
                    int(e.shape[0]), _lib.ptr(h1), st), "pgnn_edge_hidden_fwd")
                eacts = [h1]
                for n in enames[1:]:
                    eacts.append(self.fc_fwd(n, eacts[-1], True))
                dst = self._sorted_dst(e)
                agg = self._scatter_max(eacts[-1], dst, k)
                unames = mlp_names(scope + '/combined_features',
                                   len(kw['update_MLP_depth_list']))
                uacts = [agg]
                for i, n in enumerate(unames):
                    last = i + 1 == len(unames)
                    uacts.append(self.fc_fwd(n, uacts[-1], not last,
                                             residual=h if last else None))
                saved.append(('gnn', enames, hx, xo, e, eacts, dst, agg,
                              unames, uacts, off_names, off_acts, c))
                h = uacts[-1]
            else:
                raise NotImplementedError(lc['type'])
        pc = self.config['model_kwargs']['layer_configs'][-1]
        if pc['type'] != 'classaware_predictor':
            raise NotImplementedError(pc['type'])
        ps = pc['scope'] + '/predictor'
        cls_names = mlp_names(ps + '/cls', 2)
        cacts = [h, self.fc_fwd(cls_names[0], h, True)]
        cacts.append(self.fc_fwd(cls_names[1], cacts[-1], False))
        logits = cacts[-1]
        loc = []
        k = int(h.shape[0])
        pred = torch.empty((k, self.nc, self.box_len), dtype=torch.float32,
                           device=dev)
        for j in range(self.nc):
            names = mlp_names(ps + '/loc/cls_%d' % j, 3)
            a = [h]
            for i, n in enumerate(names):
                a.append(self.fc_fwd(n, a[-1], i < 2))
            pred[:, j, :] = a[-1][:, :self.box_len]
            loc.append((names, a))
        saved.append(('heads', cls_names, cacts, loc))
        self._saved = saved
        self._h_width = int(h.shape[1])
        return logits[:, :self.nc], pred

    # ---- loss ---------------------------------------------------------------------
    def _loss_inputs(self, logits, labels, gt_box, valid):
        dev = self.device
        k = int(logits.shape[0])
        lg = logits if logits.stride(1) == 1 else logits.contiguous()
        labels = labels.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        gt = gt_box.to(device=dev, dtype=torch.float32).reshape(
            k, self.box_len).contiguous()
        va = valid.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if self._class_loc_w is None and self._class_loc_w_host is not None:
            self._class_loc_w = torch.tensor(
                self._class_loc_w_host, dtype=torch.float32, device=dev)
        return lg, labels, gt, va

    def top_k_selection(self, logits, pred, labels, gt_box, valid):
        """(sel_cls, sel_loc): float masks [K] of the vertices the top-k losses
        keep (None where the loss is not a top-k one)."""
        from .models import top_k_selection
        lg, labels, gt, va = self._loss_inputs(logits, labels, gt_box, valid)
        return top_k_selection(lg, labels, pred.contiguous(), self.box_len, gt,
                               va, self.cls_kind[0], self.cls_kind[1],
                               self.cls_kind[2], self._class_loc_w,
                               self.cls_topk, self.loc_topk)

    def loss_and_grads(self, logits, pred, labels, gt_box, valid, n_total,
                       nv_total, want_grads=True, counts_dev=None, sel=None,
                       sums_out=None):
        """sums4 = [sum CE, sum loc, n, n_valid] (device double tensor),
        dlogits [K,nc], dpred [K,nc,L] for the globally normalised loss.
        counts_dev: the global (n_total, nv_total) as a device float64 [2]
        tensor instead of the two host numbers (no host read needed).
        sel: top_k_selection()'s masks (computed here when omitted and the
        loss is a top-k one); with 'top_k_huber_loss' nv_total / counts_dev[1]
        must count the valid vertices INSIDE the selection (models.py:283)."""
        if self.cls_topk or self.loc_topk:
            return self._loss_and_grads_top_k(
                logits, pred, labels, gt_box, valid, n_total, nv_total,
                want_grads, counts_dev, sel, sums_out)
        dev = self.device
        k = int(logits.shape[0])
        lg = logits if logits.stride(1) == 1 else logits.contiguous()
        labels = labels.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        gt = gt_box.to(device=dev, dtype=torch.float32).reshape(
            k, self.box_len).contiguous()
        va = valid.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        sums = sums_out if sums_out is not None else torch.empty(
            4, dtype=torch.float64, device=dev)   # (zeroed by the entries)
        dlog = torch.empty((k, self.nc), dtype=torch.float32, device=dev) \
            if want_grads else None
        dpred = torch.empty((k, self.nc, self.box_len), dtype=torch.float32,
                            device=dev) if want_grads else None
        if self.cls_kind[0] != 0 or self._class_loc_w_host is not None:
            if self._class_loc_w is None and self._class_loc_w_host is not None:
                self._class_loc_w = torch.tensor(
                    self._class_loc_w_host, dtype=torch.float32, device=dev)
            cls_scale = self.cls_w / n_total \
                if counts_dev is None and n_total > 0 else 0.0
            loc_scale = self.loc_w / nv_total \
                if counts_dev is None and nv_total > 0 else 0.0
            _lib.check(self.lib.pgnn_loss_fwd_bwd_ex(
                _lib.ptr(lg), lg.stride(0), _lib.ptr(labels), _lib.ptr(pred),
                self.box_len, _lib.ptr(gt), _lib.ptr(va), k, self.nc,
                ctypes.c_float(cls_scale), ctypes.c_float(loc_scale),
                _lib.ptr(counts_dev), ctypes.c_double(self.cls_w),
                ctypes.c_double(self.loc_w), self.cls_kind[0],
                ctypes.c_float(self.cls_kind[1]),
                ctypes.c_float(self.cls_kind[2]), _lib.ptr(self._class_loc_w),
                _lib.ptr(sums), _lib.ptr(dlog), _lib.ptr(dpred), self._st()),
                "pgnn_loss_fwd_bwd_ex")
            return sums, dlog, dpred
        if counts_dev is not None:
            _lib.check(self.lib.pgnn_loss_fwd_bwd_counts(
                _lib.ptr(lg), lg.stride(0), _lib.ptr(labels), _lib.ptr(pred),
                self.box_len, _lib.ptr(gt), _lib.ptr(va), k, self.nc,
                ctypes.c_double(self.cls_w), ctypes.c_double(self.loc_w),
                _lib.ptr(counts_dev), _lib.ptr(sums), _lib.ptr(dlog),
                _lib.ptr(dpred), self._st()), "pgnn_loss_fwd_bwd_counts")
            return sums, dlog, dpred
        cls_scale = self.cls_w / n_total if n_total > 0 else 0.0
        loc_scale = self.loc_w / nv_total if nv_total > 0 else 0.0  # div_no_nan
        _lib.check(self.lib.pgnn_loss_fwd_bwd(
            _lib.ptr(lg), lg.stride(0), _lib.ptr(labels), _lib.ptr(pred),
            self.box_len, _lib.ptr(gt), _lib.ptr(va), k, self.nc,
            ctypes.c_float(cls_scale), ctypes.c_float(loc_scale),
            _lib.ptr(sums), _lib.ptr(dlog), _lib.ptr(dpred), self._st()),
            "pgnn_loss_fwd_bwd")
        return sums, dlog, dpred

    def _loss_and_grads_top_k(self, logits, pred, labels, gt_box, valid,
                              n_total, nv_total, want_grads, counts_dev, sel,
                              sums_out=None):
        dev = self.device
        k = int(logits.shape[0])
        lg, labels, gt, va = self._loss_inputs(logits, labels, gt_box, valid)
        pred = pred.contiguous()
        if sel is None:
            sel = self.top_k_selection(lg, pred, labels, gt, va)
        sel_c, sel_l = sel
        sums = sums_out if sums_out is not None else torch.empty(
            4, dtype=torch.float64, device=dev)   # (zeroed by the entry)
        dlog = torch.empty((k, self.nc), dtype=torch.float32, device=dev) \
            if want_grads else None
        dpred = torch.empty((k, self.nc, self.box_len), dtype=torch.float32,
                            device=dev) if want_grads else None
        # the reference's mean over the k selected values, written as the
        # callers' sum / n_total: every selected vertex weighs n_rank / k
        ce_w = float(k) / self.cls_topk if self.cls_topk else 1.0
        cls_scale = self.cls_w * ce_w / n_total \
            if counts_dev is None and n_total > 0 else 0.0
        loc_scale = self.loc_w / nv_total \
            if counts_dev is None and nv_total > 0 else 0.0
        _lib.check(self.lib.pgnn_loss_fwd_bwd_sel(
            _lib.ptr(lg), lg.stride(0), _lib.ptr(labels), _lib.ptr(pred),
            self.box_len, _lib.ptr(gt), _lib.ptr(va), k, self.nc,
            ctypes.c_float(cls_scale), ctypes.c_float(loc_scale),
            _lib.ptr(counts_dev), ctypes.c_double(self.cls_w),
            ctypes.c_double(self.loc_w), self.cls_kind[0],
            ctypes.c_float(self.cls_kind[1]), ctypes.c_float(self.cls_kind[2]),
            _lib.ptr(self._class_loc_w), _lib.ptr(sel_c), _lib.ptr(sel_l),
            ctypes.c_float(ce_w), None, None, _lib.ptr(sums), _lib.ptr(dlog),
            _lib.ptr(dpred), self._st()), "pgnn_loss_fwd_bwd_sel")
        return sums, dlog, dpred

    def reg_loss(self):
        out = torch.zeros(1, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.pgnn_l1_norm(
            _lib.ptr(self.flat), _lib.ptr(self.is_weight), self.flat.numel(),
            _lib.ptr(out), self._st()), "pgnn_l1_norm")
        return self.l1_scale * float(out.item())

    # ---- backward -----------------------------------------------------------------
    def _multi(self):
        """Does a step issue its collectives?  More than one rank, or one rank
        with force_collective."""
        if self.comm is not None:
            return self.comm.world > 1 or self.force_collective
        if _world(self.pg) > 1:
            return True
        import torch.distributed as dist
        return self.force_collective and dist.is_available() and \
            dist.is_initialized()

    def backward(self, dlogits, dpred, sync_sums=None):
        """Gradients of the last forward into self.grad.  sync_sums (the loss
        sums of a multi-rank step): with a Communicator and the native step the
        all-reduce of (self.grad, sync_sums) is enqueued by the same C call;
        returns True when that happened (the caller reduces otherwise)."""
        if self._saved and self._saved[0] == 'native':
            return self._native_backward(dlogits, dpred, sync_sums)
        lib, st, dev = self.lib, self._st(), self.device
        saved = list(self._saved)
        kind, cls_names, cacts, loc = saved.pop()
        k = int(cacts[0].shape[0])
        hw = self._h_width
        dh = torch.zeros((k, hw), dtype=torch.float32, device=dev)

        def add_dh(dx, width):
            dh[:, :width] += dx[:, :width]

        # heads
        dy = torch.zeros((k, padded_width(self.nc)), dtype=torch.float32,
                         device=dev)
        dy[:, :self.nc] = dlogits
        d1 = self.fc_bwd(cls_names[1], cacts[1], cacts[2], dy, False)
        dx = self.fc_bwd(cls_names[0], cacts[0], cacts[1], d1, True)
        cwidth = self.fc[cls_names[0]].k_in
        add_dh(dx, cwidth)
        for j, (names, a) in enumerate(loc):
            dy = torch.zeros((k, padded_width(self.box_len)),
                             dtype=torch.float32, device=dev)
            dy[:, :self.box_len] = dpred[:, j, :]
            d = self.fc_bwd(names[2], a[2], a[3], dy, False)
            d = self.fc_bwd(names[1], a[1], a[2], d, True)
            d = self.fc_bwd(names[0], a[0], a[1], d, True)
            add_dh(d, cwidth)
        # layers in reverse
        while saved:
            item = saved.pop()
            if item[0] == 'gnn':
                (_, enames, hx, xo, e, eacts, dst, agg, unames, uacts,
                 off_names, off_acts, c) = item
                d = dh            # gradient w.r.t. this layer's output
                dh_in = dh.clone()  # residual branch (gnn.py:372)
                for i in range(len(unames) - 1, -1, -1):
                    last = i + 1 == len(unames)
                    dcur = d if not last else d.clone()
                    d = self.fc_bwd(unames[i], uacts[i], uacts[i + 1], dcur,
                                    not last)
                dagg = d
                if self.sparse_adjoint and self.fc[enames[-1]].wt is not None:
                    # last edge layer + scatter-max: sparse adjoint (returns
                    # the gradient w.r.t. its input, already ReLU-masked)
                    g = self._segmax_fc_bwd(enames[-1], eacts[-1], dst, agg,
                                            dagg, eacts[-2])
                    dense_from = len(enames) - 2
                else:
                    g = self._scatter_max_bwd(eacts[-1], dst, agg, dagg)
                    dense_from = len(enames) - 1
                for i in range(dense_from, 0, -1):
                    # g is already masked by the ReLU of layer i (relu_mask)
                    g = self.fc_bwd(enames[i], eacts[i - 1], eacts[i], g,
                                    relu=False)
                    _lib.check(lib.pgnn_relu_mask_mul(
                        _lib.ptr(g), _lib.ptr(eacts[i - 1]), g.numel(), st),
                        "pgnn_relu_mask_mul")
                kk = int(hx.shape[0])
                wq = int(g.shape[1])
                dp = torch.empty((kk, wq), dtype=torch.float32, device=dev)
                dq = torch.empty((kk, wq), dtype=torch.float32, device=dev)
                _lib.check(lib.pgnn_edge_hidden_bwd(
                    _lib.ptr(g), wq, _lib.ptr(e), int(e.shape[0]), kk,
                    _lib.ptr(dp), _lib.ptr(dq), st), "pgnn_edge_hidden_bwd")
                w1 = self.fc[enames[0]]
                if getattr(self, 'debug', None) is not None:
                    self.debug[enames[0]] = dict(
                        g1=g.clone(), dp=dp.clone(), dq=dq.clone(), dagg=dagg.clone(),
                        dh_out=dh.clone(), hx=hx.clone(), agg=agg.clone())
                # P = [h, x] W1 + b1
                dhx = self.fc_bwd(enames[0], hx, None, dp, False)
                dh_in[:, :c] += dhx[:, :c]
                # Q = x' Wx, Wx = rows c..c+2 of W1 (pre-activation is P - Q; the
                # minus sign is already in dq)
                gwx_ptr = ctypes.c_void_p(w1.gw.data_ptr() + 4 * c * w1.n_out)
                self._weight_grad(xo, 3, 3, dq, w1.n_out, gwx_ptr, None)
                if off_names is not None:
                    # dx' = dQ Wx^T (image refreshed by repack())
                    d = self._mlp1(self.wx_packed_t[enames[0]], w1.n_out, 3,
                                   False, dq, w1.n_out)
                    for i in range(len(off_names) - 1, -1, -1):
                        d = self.fc_bwd(off_names[i], off_acts[i],
                                        off_acts[i + 1], d,
                                        i + 1 < len(off_names))
                    dh_in[:, :c] += d[:, :c]
                dh = dh_in
            else:
                _, names, acts, dst, agg, onames, oacts, upper = item
                d = dh
                for i in range(len(onames) - 1, -1, -1):
                    d = self.fc_bwd(onames[i], oacts[i], oacts[i + 1], d, True)
                # (the gradient w.r.t. the stage's input rows is wanted only
                # above the first level; acts[0] is a gathered input, not a
                # ReLU output: no mask there)
                if self.sparse_adjoint and self.fc[names[-1]].wt is not None:
                    first = len(names) == 1
                    g = self._segmax_fc_bwd(
                        names[-1], acts[-1], dst, agg, d, acts[-2],
                        need_dx=not first or upper is not None,
                        mask_x=not first)
                    dense_from = len(names) - 2
                else:
                    g = self._scatter_max_bwd(acts[-1], dst, agg, d)
                    dense_from = len(names) - 1
                for i in range(dense_from, -1, -1):
                    g = self.fc_bwd(names[i], acts[i], acts[i + 1], g,
                                    relu=False,
                                    need_dx=i > 0 or upper is not None)
                    if i > 0:
                        _lib.check(lib.pgnn_relu_mask_mul(
                            _lib.ptr(g), _lib.ptr(acts[i]), g.numel(), st),
                            "pgnn_relu_mask_mul")
                if upper is not None:
                    # adjoint of the gather h[src]: dh_prev[src] += g[:, :n]
                    # (the coordinate columns behind them have no parameters
                    # upstream)
                    e, n_feat, k_prev, w_prev = upper
                    src = e[:, 0].contiguous()
                    dh = torch.zeros((k_prev, w_prev), dtype=torch.float32,
                                     device=dev)
                    _lib.check(lib.pgnn_scatter_sum_f32(
                        _lib.ptr(g), g.stride(0), _lib.ptr(src),
                        int(src.shape[0]), n_feat, k_prev, _lib.ptr(dh),
                        dh.stride(0), 0, None, st), "pgnn_scatter_sum_f32")
        self._saved = None

    # ---- one training step ----------------------------------------------------------
    def train_step(self, batch, apply=True, num_valid=None,
                   after_enqueue=None, deferred=False):
        """batch = (input_v, vertex_coord_list, keypoint_indices_list,
        edges_list, cls_labels [K,1], encoded_boxes [K,1,L], valid_boxes
        [K,1,1]) -- what train.py's batch_data returns for this rank.
        Returns the loss dict of models.py:308-311 (global values).

        num_valid: sum(valid_boxes) when the caller already knows it on the
        host (the data loader does): saves the one device read that otherwise
        sits between forward and backward.  after_enqueue: called once forward,
        loss and backward are all queued and before the host waits for the loss
        sums (the gradient all-reduce, SGD and the image refresh are queued by
        then too) -- the place to build the NEXT batch's graph on another stream.
        deferred (extension): return a StepResult instead of the dict; the
        step's numbers are copied to pinned host memory behind the step and
        `.get()` waits for that copy alone, so the caller can queue the next
        step before it looks at this one's loss (sess.run hands the values
        back at once, train.py:563-575; the values are the same)."""
        (input_v, coords, kps, edges, labels, boxes, valid) = batch
        self.grad.zero_()
        # the step's numbers, written in place by their kernels: [sum CE, sum
        # loc, n, n_valid | L1 of the weights | global n, n_valid] -- no
        # concatenate at the end
        rec = torch.empty(7, dtype=torch.float64, device=self.device)
        va = torch.as_tensor(valid).to(self.device, torch.float32).reshape(-1)
        k = int(va.shape[0])
        multi = self._multi()
        counts = counts_dev = None
        if self.loc_topk:
            # 'top_k_huber_loss': num_valid_endpoint counts the valid vertices
            # among the k worst (models.py:283-285) -- known after the forward
            # only, and left on the device
            num_valid = None
        if multi:
            # several ranks: the global counts are all-reduced and STAY on the
            # device (the loss kernel divides by them): nothing is read back
            # between forward and backward.  Queued before the forward when the
            # loader knows this rank's count, so the tiny collective overlaps it.
            if num_valid is not None:
                counts_dev = allreduce_endpoint_counts_device(
                    k, float(num_valid), self.device, self.pg,
                    self.force_collective, self.comm, out=rec[5:7])
        elif num_valid is not None:
            counts = (float(k), float(num_valid))
        logits, pred = self.forward(input_v, coords, kps, edges)
        self.last_logits = logits   # for the streaming metrics (train.py:299)
        assert int(logits.shape[0]) == k, "labels do not match the vertices"
        sel = None
        if self.cls_topk or self.loc_topk:
            sel = self.top_k_selection(logits, pred, torch.as_tensor(labels),
                                       torch.as_tensor(boxes), va)
        if self.loc_topk:
            counts_dev = allreduce_endpoint_counts_device(
                k, (va * sel[1]).sum(), self.device, self.pg,
                self.force_collective and multi, self.comm if multi else None,
                out=rec[5:7])
        elif multi and counts_dev is None:
            counts_dev = allreduce_endpoint_counts_device(
                k, va.sum(), self.device, self.pg, self.force_collective,
                self.comm, out=rec[5:7])
        elif not multi and counts is None:
            counts = (float(k), float(va.sum().item()))
        # unify_copies: global endpoint counts (train.py:268-284)
        n_total, nv_total = counts if counts is not None else (None, None)
        sums, dlog, dpred = self.loss_and_grads(
            logits, pred, torch.as_tensor(labels), torch.as_tensor(boxes), va,
            n_total, nv_total, counts_dev=counts_dev, sel=sel,
            sums_out=rec[0:4])
        ev = getattr(self, 'allreduce_events', None)
        # with a Communicator the native step enqueues the all-reduce itself,
        # right behind its last gradient kernel (pgnn_trainer_backward_sync);
        # bench.py's timing keeps the two apart to see the collective alone
        synced = self.backward(
            dlog, dpred, sync_sums=sums if multi and ev is None else None)
        if ev is not None:  # bench.py: device time of the collective
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if multi and not synced:
            allreduce_gradients(self.grad, sums, self.pg,
                                self.force_collective, self.comm)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        lr = learning_rate(self.train_config, self.global_step)
        # Everything the device still has to do is queued BEFORE the host reads
        # anything: the L1 term of the weights this step used, SGD, the image
        # refresh.  The step's numbers then come back in one copy.
        l1 = rec[4:5]
        _lib.check(self.lib.pgnn_l1_norm(         # (the call zeroes `l1` first)
            _lib.ptr(self.flat), _lib.ptr(self.is_weight), self.flat.numel(),
            _lib.ptr(l1), self._st()), "pgnn_l1_norm")
        if apply:
            self._apply_gradients(lr)
            self.repack()
            self.global_step += 1
            self.opt_step += 1
        res = StepResult(self, rec[:5] if counts_dev is None else rec, counts,
                         lr)
        if after_enqueue is not None:
            # (behind the collective and the update: the host reads of a graph
            # build must not hold back the all-reduce's launch)
            after_enqueue()
        return res if deferred else res.get()
