"""ORACLE (test infrastructure only) -- the same GNN arithmetic as
oracle/gnn_oracle.py (models/gnn.py:222-283, 298-373, 133-163;
models/models.py:79-163), written for SPEED on the host: torch-CPU, every
per-edge stage evaluated in row chunks that stay in the last-level cache
(gather -> GEMM chain -> segment max per chunk; nothing of size E x C is ever
materialised).  Two ways to use the cores: torch's intra-op threads inside
every op (WORKERS = 0), or -- what scales on a many-core host -- WORKERS
Python threads that each take whole chunks through gather -> GEMM chain ->
segment max with single-threaded ops (torch releases the GIL inside an op; the
chunks are independent up to the final maximum(), which is taken under a lock).

Only `bench.py`'s `cpu_baseline` leg times it (the "CPU port of the reference
path" of the bench contract: the NumPy oracle spends its time in concatenate /
argsort / reduceat and reaches ~4 % of the host's sgemm rate, this one is
GEMM-bound) and `tests/test_oracle_cpu.py` holds it to gnn_oracle.predict.
Edge lists must be grouped by ascending dst (what the reference's graph
generator emits and `graph_oracle` reproduces); the product path never imports
this module.
"""
import numpy as np
import torch

CHUNK_ROWS = 1 << 15
WORKERS = 0          # > 0: chunk-parallel worker threads (see module docstring)
_POOL = {}


def _pool(n):
    from concurrent.futures import ThreadPoolExecutor
    p = _POOL.get(n)
    if p is None:
        p = _POOL[n] = ThreadPoolExecutor(max_workers=n)
    return p


def _map_chunks(fn, n_rows):
    """fn(lo, hi) over row chunks: in order on this thread, or on WORKERS
    threads."""
    spans = [(lo, min(n_rows, lo + CHUNK_ROWS))
             for lo in range(0, n_rows, CHUNK_ROWS)]
    if WORKERS > 0 and len(spans) > 1:
        return list(_pool(WORKERS).map(lambda s: fn(*s), spans))
    return [fn(lo, hi) for lo, hi in spans]


def _layers(params, scope):
    out = []
    i = 0
    while True:
        name = scope + '/fully_connected' + ('' if i == 0 else '_%d' % i)
        if name + '/weights' not in params:
            break
        out.append((torch.from_numpy(np.ascontiguousarray(
                        params[name + '/weights'], dtype=np.float32)),
                    torch.from_numpy(np.ascontiguousarray(
                        params[name + '/biases'], dtype=np.float32))))
        i += 1
    return out


def _mlp(x, layers, is_logits):
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        x = torch.addmm(b, x, w)
        if not (is_logits and i == n - 1):
            x = torch.relu_(x)
    return x


def _edge_mlp_segment_max(gather, layers, dst, num_segments):
    """max over the runs of equal dst of MLP(gather(rows)), chunk by chunk.
    dst ascending; a run cut by a chunk boundary is merged with maximum()."""
    import threading
    lowest = float(np.finfo(np.float32).min)
    width = layers[-1][0].shape[1]
    out = torch.full((num_segments, width), lowest, dtype=torch.float32)
    n = int(dst.shape[0])
    lock = threading.Lock()

    def chunk(lo, hi):
        rows = _mlp(gather(lo, hi), layers, is_logits=False)
        seg, counts = torch.unique_consecutive(dst[lo:hi], return_counts=True)
        red = torch.segment_reduce(rows, 'max', lengths=counts)
        with lock:   # a run cut by a chunk boundary meets its other half here
            out[seg] = torch.maximum(out[seg], red)
    _map_chunks(chunk, n)
    return out


def _mlp_rows(x, layers, is_logits):
    """_mlp over row chunks (the per-vertex stages on the worker threads)."""
    if WORKERS <= 0 or x.shape[0] <= CHUNK_ROWS:
        return _mlp(x, layers, is_logits)
    parts = _map_chunks(lambda lo, hi: _mlp(x[lo:hi], layers, is_logits),
                        int(x.shape[0]))
    return torch.cat(parts, dim=0)


def predict(params, config, initial_vertex_features, vertex_coord_list,
            keypoint_indices_list, edges_list):
    """MultiLayerFastLocalGraphModelV2.predict in float32 -> (logits, boxes)
    as NumPy arrays."""
    t = lambda a, dt=torch.float32: torch.from_numpy(  # noqa: E731
        np.ascontiguousarray(a)).to(dt)
    feats = t(initial_vertex_features)
    coords = [t(c) for c in vertex_coord_list]
    layer_configs = config['model_kwargs']['layer_configs']
    with torch.no_grad():
        for lc in layer_configs[:-1]:
            lvl = lc['graph_level']
            e = t(edges_list[lvl], torch.int64)
            src, dst = e[:, 0].contiguous(), e[:, 1].contiguous()
            if dst.numel() and bool((dst[1:] < dst[:-1]).any()):
                raise ValueError("edge list is not grouped by ascending dst")
            scope = lc['scope']
            if lc['type'] == 'scatter_max_point_set_pooling':
                pc = coords[lvl]
                kp = t(keypoint_indices_list[lvl], torch.int64).reshape(-1)
                pf = feats

                def gather(lo, hi, pf=pf, pc=pc, kp=kp, src=src, dst=dst):
                    s = src[lo:hi]
                    return torch.cat([pf[s], pc[s] - pc[kp[dst[lo:hi]]]], dim=1)
                agg = _edge_mlp_segment_max(
                    gather, _layers(params, scope + '/extract_vertex_features'),
                    dst, int(kp.shape[0]))
                feats = _mlp_rows(agg, _layers(params,
                                               scope + '/combined_features'),
                                  is_logits=False)
            elif lc['type'] == 'scatter_max_graph_auto_center_net':
                h, x = feats, coords[lvl]
                x_dst = x
                if lc['kwargs']['auto_offset']:
                    x_dst = x + _mlp_rows(h, _layers(params, scope),
                                          is_logits=True)

                def gather(lo, hi, h=h, x=x, x_dst=x_dst, src=src, dst=dst):
                    s = src[lo:hi]
                    return torch.cat([h[s], x[s] - x_dst[dst[lo:hi]]], dim=1)
                agg = _edge_mlp_segment_max(
                    gather, _layers(params, scope + '/extract_vertex_features'),
                    dst, int(h.shape[0]))
                feats = _mlp_rows(agg, _layers(params,
                                               scope + '/combined_features'),
                                  is_logits=True) + h
            else:
                raise NotImplementedError(lc['type'])
        pc_ = layer_configs[-1]
        assert pc_['type'] == 'classaware_predictor'
        scope = pc_['scope'] + '/predictor'
        logits = _mlp(feats, _layers(params, scope + '/cls'), is_logits=True)
        boxes = [_mlp(feats, _layers(params, scope + '/loc/cls_%d' % c),
                      is_logits=True)[:, None, :]
                 for c in range(config['num_classes'])]
    return logits.numpy(), torch.cat(boxes, dim=1).numpy()
