"""ORACLE (test infrastructure only) -- a NumPy interpreter for the TensorFlow
graphs the reference itself serialized.

Only `tests/`, `tests/golden/make_golden_tfgraph.py`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; the product path
never does.

The reference cannot be executed here (tensorflow-gpu==1.15.0 is not
installable, README.md:20-23), but it ships the graphs `train.py` built:
`checkpoints/*/model-*.meta` are `MetaGraphDef` protobufs holding every node
`train.py:178-405` created -- the per-tower placeholders, the
GatherV2 / ConcatV2 / MatMul / BiasAdd / Relu / UnsortedSegmentMax wiring of
`models/gnn.py:222-373`, the loss of `models/models.py:170-311`, the
`unify_copies` re-weighting (`train.py:264-288`), `tf.gradients` for every
tower, `average_gradients` (`util/tf_util.py:3-43`), the
`ApplyGradientDescent` update and the `tf.metrics.*` update ops
(`train.py:301-368`).  Evaluating THAT graph is evaluating the reference: the
wiring (which tensor feeds which op, in which order, with which axis, mask and
attribute) is the reference's own, byte for byte; only the arithmetic of each
primitive op (MatMul, Relu, UnsortedSegmentMax, ...) is restated from TF 1.15's
documented kernel semantics, in float32.

This file contains
  * a protobuf wire-format reader for MetaGraphDef / GraphDef / NodeDef /
    AttrValue / TensorProto (no `tensorflow`, no generated _pb2 modules);
  * `Graph.run(fetches, feed_dict)`: demand-driven evaluation with TF's
    control-dependency, Switch/Merge (dead-tensor) and variable semantics;
  * one NumPy kernel per op type present in the shipped graphs.

Summation order inside MatMul / Sum / Mean is NumPy's (BLAS / pairwise), not
Eigen's: results equal TF's to float32 rounding, not bit for bit.
"""
import struct
import sys

import numpy as np

# --------------------------------------------------------------- wire format


def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one protobuf message."""
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed(v):
    """varint -> int64 (two's complement)."""
    return v - (1 << 64) if v >= (1 << 63) else v


# tensorflow/core/framework/types.proto
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16,
       6: np.int8, 7: object, 9: np.int64, 10: np.bool_}


def _parse_shape(buf):
    """TensorShapeProto -> tuple (None for unknown rank, -1 unknown dim)."""
    dims = []
    for f, _, v in _fields(buf):
        if f == 2:
            size = 0
            for g, _, w in _fields(v):
                if g == 1:
                    size = _signed(w)
            dims.append(size)
        elif f == 3 and v:
            return None
    return tuple(dims)


def _packed(v, wt, fmt, size):
    if wt == 2:
        return list(struct.unpack("<%d%s" % (len(v) // size, fmt), bytes(v)))
    return [struct.unpack("<" + fmt, v)[0]]


def _parse_tensor(buf):
    """TensorProto -> ndarray."""
    dtype = None
    shape = ()
    content = None
    vals = []
    for f, wt, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = _parse_shape(v)
        elif f == 4:
            content = bytes(v)
        elif f == 5:
            vals += _packed(v, wt, "f", 4)
        elif f == 6:
            vals += _packed(v, wt, "d", 8)
        elif f in (7, 10, 11):
            if wt == 2:
                pos = 0
                while pos < len(v):
                    x, pos = _varint(v, pos)
                    vals.append(_signed(x))
            else:
                vals.append(_signed(v))
        elif f == 8:
            vals.append(bytes(v))
    np_dt = _DT[dtype]
    n = int(np.prod(shape)) if shape else 1
    if content is not None and np_dt is not object:
        return np.frombuffer(content, dtype=np_dt).reshape(shape).copy()
    if np_dt is object:
        arr = np.empty(n, dtype=object)
        for i in range(n):
            arr[i] = vals[i] if i < len(vals) else (vals[-1] if vals else b"")
        return arr.reshape(shape)
    if not vals:
        return np.zeros(shape, dtype=np_dt)
    if len(vals) < n:                      # last value repeats (TF convention)
        vals = vals + [vals[-1]] * (n - len(vals))
    return np.array(vals[:n], dtype=np_dt).reshape(shape)


def _parse_attr(buf):
    """AttrValue -> python value."""
    for f, wt, v in _fields(buf):
        if f == 2:
            return bytes(v)
        if f == 3:
            return _signed(v)
        if f == 4:
            return struct.unpack("<f", v)[0]
        if f == 5:
            return bool(v)
        if f == 6:
            return ("dtype", v)
        if f == 7:
            return ("shape", _parse_shape(v))
        if f == 8:
            return _parse_tensor(v)
        if f == 1:
            out = []
            for g, wt2, w in _fields(v):
                if g == 2:
                    out.append(bytes(w))
                elif g == 3:
                    if wt2 == 2:
                        pos = 0
                        while pos < len(w):
                            x, pos = _varint(w, pos)
                            out.append(_signed(x))
                    else:
                        out.append(_signed(w))
                elif g == 4:
                    out += _packed(w, wt2, "f", 4)
                elif g == 5:
                    out.append(bool(w))
                elif g == 6:
                    if wt2 == 2:
                        pos = 0
                        while pos < len(w):
                            x, pos = _varint(w, pos)
                            out.append(("dtype", x))
                    else:
                        out.append(("dtype", w))
                elif g == 7:
                    out.append(("shape", _parse_shape(w)))
            return out
    return None


class Node(object):
    __slots__ = ("name", "op", "inputs", "device", "_attr_raw", "_attr")

    def __init__(self, buf):
        self.name = self.op = self.device = ""
        self.inputs = []
        self._attr_raw = {}
        self._attr = {}
        for f, _, v in _fields(buf):
            if f == 1:
                self.name = bytes(v).decode()
            elif f == 2:
                self.op = bytes(v).decode()
            elif f == 3:
                self.inputs.append(bytes(v).decode())
            elif f == 4:
                self.device = bytes(v).decode()
            elif f == 5:
                key = val = None
                for g, _, w in _fields(v):
                    if g == 1:
                        key = bytes(w).decode()
                    elif g == 2:
                        val = w
                self._attr_raw[key] = val

    def attr(self, key, default=None):
        if key not in self._attr:
            raw = self._attr_raw.get(key)
            self._attr[key] = default if raw is None else _parse_attr(raw)
            if self._attr[key] is None:
                self._attr[key] = default
        return self._attr[key]

    def dtype(self, key):
        v = self.attr(key)
        return _DT[v[1]]

    def __repr__(self):
        return "Node(%s, %s, %s)" % (self.name, self.op, self.inputs)


class _Dead(object):
    """The untaken branch of a Switch (tensorflow/core/common_runtime:
    a dead tensor makes every consumer dead, Merge forwards the live input)."""
    def __repr__(self):
        return "DEAD"


DEAD = _Dead()


def _split_input(s):
    """'^name' -> (name, None); 'name:k' -> (name, k); 'name' -> (name, 0)."""
    if s.startswith("^"):
        return s[1:], None
    if ":" in s:
        a, b = s.rsplit(":", 1)
        return a, int(b)
    return s, 0


# ---------------------------------------------------------------- op kernels

_KERNELS = {}


def _op(*names):
    def deco(fn):
        for n in names:
            _KERNELS[n] = fn
        return fn
    return deco


def _axes(a, ndim):
    a = np.asarray(a).reshape(-1).astype(np.int64)
    return tuple(int(x) % ndim if ndim else 0 for x in a)


@_op("Identity", "StopGradient", "PreventGradient", "Snapshot")
def _identity(g, n, x):
    return (x,)


@_op("IdentityN")
def _identity_n(g, n, *xs):
    return tuple(xs)


@_op("Const")
def _const(g, n):
    return (n.attr("value"),)


@_op("NoOp", "Assert")
def _noop(g, n, *xs):
    if n.op == "Assert" and not bool(np.all(xs[0])):
        raise AssertionError("tf.Assert failed at %s: %r" % (n.name, xs[1:]))
    return ()


@_op("Add", "AddV2")
def _add(g, n, a, b):
    return (a + b,)


@_op("Sub")
def _sub(g, n, a, b):
    return (a - b,)


@_op("Mul")
def _mul(g, n, a, b):
    return (a * b,)


@_op("RealDiv")
def _realdiv(g, n, a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return (np.asarray(a / b, dtype=np.result_type(a, b)),)


@_op("DivNoNan")
def _divnonan(g, n, a, b):
    a, b = np.broadcast_arrays(np.asarray(a), np.asarray(b))
    out = np.zeros(a.shape, dtype=a.dtype)
    np.divide(a, b, out=out, where=(b != 0))
    return (out,)


@_op("FloorDiv")
def _floordiv(g, n, a, b):
    return (np.floor_divide(a, b),)


@_op("FloorMod")
def _floormod(g, n, a, b):
    return (np.mod(a, b),)


@_op("Maximum")
def _maximum(g, n, a, b):
    return (np.maximum(a, b),)


@_op("Minimum")
def _minimum(g, n, a, b):
    return (np.minimum(a, b),)


@_op("Pow")
def _pow(g, n, a, b):
    return (np.power(a, b).astype(np.asarray(a).dtype),)


@_op("Neg")
def _neg(g, n, a):
    return (-a,)


@_op("Abs")
def _abs(g, n, a):
    return (np.abs(a),)


@_op("Sign")
def _sign(g, n, a):
    return (np.sign(a),)


@_op("Log")
def _log(g, n, a):
    with np.errstate(divide="ignore", invalid="ignore"):
        return (np.log(a),)


@_op("Exp")
def _exp(g, n, a):
    return (np.exp(a),)


@_op("Sqrt")
def _sqrt(g, n, a):
    return (np.sqrt(a),)


@_op("Square")
def _square(g, n, a):
    return (a * a,)


@_op("Floor")
def _floor(g, n, a):
    return (np.floor(a),)


@_op("Reciprocal", "Inv")
def _recip(g, n, a):
    return ((1 / a).astype(np.asarray(a).dtype),)


@_op("Equal")
def _equal(g, n, a, b):
    return (np.equal(a, b),)


@_op("NotEqual")
def _nequal(g, n, a, b):
    return (np.not_equal(a, b),)


@_op("Greater")
def _greater(g, n, a, b):
    return (np.greater(a, b),)


@_op("GreaterEqual")
def _ge(g, n, a, b):
    return (np.greater_equal(a, b),)


@_op("Less")
def _less(g, n, a, b):
    return (np.less(a, b),)


@_op("LessEqual")
def _le(g, n, a, b):
    return (np.less_equal(a, b),)


@_op("LogicalAnd")
def _land(g, n, a, b):
    return (np.logical_and(a, b),)


@_op("LogicalOr")
def _lor(g, n, a, b):
    return (np.logical_or(a, b),)


@_op("LogicalNot")
def _lnot(g, n, a):
    return (np.logical_not(a),)


@_op("IsNan")
def _isnan(g, n, a):
    return (np.isnan(a),)


@_op("Select")
def _select(g, n, c, a, b):
    c = np.asarray(c)
    a = np.asarray(a)
    if c.ndim == 1 and a.ndim > 1:       # tf.where row-select form
        c = c.reshape((-1,) + (1,) * (a.ndim - 1))
    return (np.where(c, a, b).astype(a.dtype),)


@_op("ZerosLike")
def _zeros_like(g, n, a):
    return (np.zeros_like(a),)


@_op("OnesLike")
def _ones_like(g, n, a):
    return (np.ones_like(a),)


@_op("Fill")
def _fill(g, n, dims, v):
    v = np.asarray(v)
    return (np.full(tuple(int(d) for d in np.asarray(dims).reshape(-1)), v,
                    dtype=v.dtype),)


@_op("Cast")
def _cast(g, n, a):
    dst = n.dtype("DstT")
    a = np.asarray(a)
    if np.issubdtype(dst, np.integer) and np.issubdtype(a.dtype, np.floating):
        a = np.trunc(a)
    return (a.astype(dst),)


@_op("Shape")
def _shape(g, n, a):
    return (np.array(np.shape(a), dtype=n.dtype("out_type")
                     if n.attr("out_type") else np.int32),)


@_op("ShapeN")
def _shape_n(g, n, *xs):
    dt = n.dtype("out_type") if n.attr("out_type") else np.int32
    return tuple(np.array(np.shape(x), dtype=dt) for x in xs)


@_op("Size")
def _size(g, n, a):
    return (np.array(np.size(a), dtype=n.dtype("out_type")
                     if n.attr("out_type") else np.int32),)


@_op("Rank")
def _rank(g, n, a):
    return (np.array(np.ndim(a), dtype=np.int32),)


@_op("Reshape")
def _reshape(g, n, a, shape):
    return (np.reshape(a, tuple(int(s) for s in np.asarray(shape).reshape(-1))),)


@_op("ExpandDims")
def _expand_dims(g, n, a, axis):
    a = np.asarray(a)
    ax = int(np.asarray(axis).reshape(-1)[0])
    if ax < 0:
        ax += a.ndim + 1
    return (np.expand_dims(a, ax),)


@_op("Squeeze")
def _squeeze(g, n, a):
    dims = n.attr("squeeze_dims", [])
    a = np.asarray(a)
    if dims:
        return (np.squeeze(a, axis=tuple(int(d) % a.ndim for d in dims)),)
    return (np.squeeze(a),)


@_op("Transpose")
def _transpose(g, n, a, perm):
    return (np.transpose(a, tuple(int(p) for p in np.asarray(perm))),)


@_op("Tile")
def _tile(g, n, a, mult):
    return (np.tile(a, tuple(int(m) for m in np.asarray(mult).reshape(-1))),)


@_op("Pack")
def _pack(g, n, *xs):
    axis = n.attr("axis", 0)
    return (np.stack([np.asarray(x) for x in xs], axis=axis),)


@_op("Unpack")
def _unpack(g, n, a):
    axis = n.attr("axis", 0)
    a = np.asarray(a)
    return tuple(np.take(a, i, axis=axis) for i in range(a.shape[axis]))


@_op("ConcatV2")
def _concat(g, n, *xs):
    axis = int(np.asarray(xs[-1]))
    return (np.concatenate([np.asarray(x) for x in xs[:-1]], axis=axis),)


@_op("ConcatOffset")
def _concat_offset(g, n, dim, *shapes):
    dim = int(np.asarray(dim))
    out = []
    off = 0
    for s in shapes:
        s = np.asarray(s)
        o = np.zeros_like(s)
        o[dim] = off
        off += int(s[dim])
        out.append(o)
    return tuple(out)


@_op("Slice")
def _slice(g, n, a, begin, size):
    a = np.asarray(a)
    begin = np.asarray(begin).reshape(-1)
    size = np.asarray(size).reshape(-1)
    idx = tuple(slice(int(b), a.shape[i] if int(s) == -1 else int(b) + int(s))
                for i, (b, s) in enumerate(zip(begin, size)))
    return (a[idx],)


@_op("StridedSlice")
def _strided_slice(g, n, a, begin, end, strides):
    """tensorflow/core/util/strided_slice_op.cc semantics."""
    a = np.asarray(a)
    begin = [int(x) for x in np.asarray(begin).reshape(-1)]
    end = [int(x) for x in np.asarray(end).reshape(-1)]
    strides = [int(x) for x in np.asarray(strides).reshape(-1)]
    bm, em = n.attr("begin_mask", 0), n.attr("end_mask", 0)
    elm, nam = n.attr("ellipsis_mask", 0), n.attr("new_axis_mask", 0)
    sam = n.attr("shrink_axis_mask", 0)
    idx = []
    for i in range(len(begin)):
        bit = 1 << i
        if elm & bit:
            idx.append(Ellipsis)
        elif nam & bit:
            idx.append(np.newaxis)
        elif sam & bit:
            idx.append(begin[i])
        else:
            b = None if bm & bit else begin[i]
            e = None if em & bit else end[i]
            idx.append(slice(b, e, strides[i]))
    return (a[tuple(idx)],)


@_op("StridedSliceGrad")
def _strided_slice_grad(g, n, shape, begin, end, strides, dy):
    out = np.zeros(tuple(int(s) for s in np.asarray(shape)), dtype=dy.dtype)
    begin = [int(x) for x in np.asarray(begin).reshape(-1)]
    end = [int(x) for x in np.asarray(end).reshape(-1)]
    strides = [int(x) for x in np.asarray(strides).reshape(-1)]
    bm, em = n.attr("begin_mask", 0), n.attr("end_mask", 0)
    elm, nam = n.attr("ellipsis_mask", 0), n.attr("new_axis_mask", 0)
    sam = n.attr("shrink_axis_mask", 0)
    idx = []
    for i in range(len(begin)):
        bit = 1 << i
        if elm & bit:
            idx.append(Ellipsis)
        elif nam & bit:
            idx.append(np.newaxis)
        elif sam & bit:
            idx.append(begin[i])
        else:
            idx.append(slice(None if bm & bit else begin[i],
                             None if em & bit else end[i], strides[i]))
    out[tuple(idx)] = np.reshape(dy, out[tuple(idx)].shape)
    return (out,)


@_op("Range")
def _range(g, n, start, limit, delta):
    dt = np.asarray(start).dtype
    return (np.arange(np.asarray(start).item(), np.asarray(limit).item(),
                      np.asarray(delta).item()).astype(dt),)


@_op("GatherV2")
def _gather(g, n, params, indices, axis):
    return (np.take(params, np.asarray(indices),
                    axis=int(np.asarray(axis))),)


@_op("GatherNd")
def _gather_nd(g, n, params, indices):
    indices = np.asarray(indices)
    return (np.asarray(params)[tuple(np.moveaxis(indices, -1, 0))],)


@_op("ScatterNd")
def _scatter_nd(g, n, indices, updates, shape):
    updates = np.asarray(updates)
    out = np.zeros(tuple(int(s) for s in np.asarray(shape)),
                   dtype=updates.dtype)
    indices = np.asarray(indices)
    np.add.at(out, tuple(np.moveaxis(indices, -1, 0)), updates)
    return (out,)


@_op("Where")
def _where(g, n, c):
    return (np.argwhere(np.asarray(c)).astype(np.int64),)


@_op("MatMul")
def _matmul(g, n, a, b):
    if n.attr("transpose_a", False):
        a = a.T
    if n.attr("transpose_b", False):
        b = b.T
    return (np.matmul(a, b),)


@_op("BiasAdd")
def _bias_add(g, n, a, b):
    return (a + b,)


@_op("BiasAddGrad")
def _bias_add_grad(g, n, dy):
    dy = np.asarray(dy)
    return (dy.reshape(-1, dy.shape[-1]).sum(axis=0, dtype=dy.dtype),)


@_op("Relu")
def _relu(g, n, a):
    return (np.maximum(a, np.zeros((), dtype=np.asarray(a).dtype)),)


@_op("ReluGrad")
def _relu_grad(g, n, dy, features):
    return (np.where(np.asarray(features) > 0, dy,
                     np.zeros((), dtype=np.asarray(dy).dtype)),)


@_op("Softmax")
def _softmax(g, n, a):
    z = a - a.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return ((e / e.sum(axis=-1, keepdims=True)).astype(a.dtype),)


@_op("SparseSoftmaxCrossEntropyWithLogits")
def _sparse_xent(g, n, logits, labels):
    """tensorflow/core/kernels/sparse_xent_op.h: loss = log(sum exp(z-max))
    - (z[label]-max); backprop = softmax - onehot."""
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    s = e.sum(axis=-1, keepdims=True)
    rows = np.arange(logits.shape[0])
    lab = np.asarray(labels).astype(np.int64)
    loss = np.log(s[:, 0]) - z[rows, lab]
    bp = e / s
    bp[rows, lab] -= 1
    return (loss.astype(logits.dtype), bp.astype(logits.dtype))


def _reduce(fn):
    def kernel(g, n, a, axes):
        a = np.asarray(a)
        ax = _axes(axes, a.ndim)
        keep = n.attr("keep_dims", False)
        if a.ndim == 0:
            return (a.copy(),)
        if fn in (np.sum, np.mean, np.prod):
            return (np.asarray(fn(a, axis=ax, keepdims=keep, dtype=a.dtype)),)
        return (np.asarray(fn(a, axis=ax, keepdims=keep)),)
    return kernel


_KERNELS["Sum"] = _reduce(np.sum)
_KERNELS["Mean"] = _reduce(np.mean)
_KERNELS["Prod"] = _reduce(np.prod)
_KERNELS["Max"] = _reduce(np.max)
_KERNELS["Min"] = _reduce(np.min)
_KERNELS["All"] = _reduce(np.all)
_KERNELS["Any"] = _reduce(np.any)


@_op("ArgMax")
def _argmax(g, n, a, axis):
    return (np.argmax(a, axis=int(np.asarray(axis))).astype(
        n.dtype("output_type") if n.attr("output_type") else np.int64),)


@_op("AddN")
def _add_n(g, n, *xs):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return (out,)


@_op("Cumsum", "Cumprod")
def _cum(g, n, a, axis):
    a = np.asarray(a)
    ax = int(np.asarray(axis))
    fn = np.cumsum if n.op == "Cumsum" else np.cumprod
    ident = 0 if n.op == "Cumsum" else 1
    if n.attr("reverse", False):
        a = np.flip(a, ax)
    out = fn(a, axis=ax, dtype=a.dtype)
    if n.attr("exclusive", False):
        out = np.roll(out, 1, axis=ax)
        sl = [slice(None)] * a.ndim
        sl[ax] = 0
        out[tuple(sl)] = ident
    if n.attr("reverse", False):
        out = np.flip(out, ax)
    return (out,)


def _segment_reduce(ufunc, flat, ids, out):
    """out[ids[i]] = ufunc(out[ids[i]], flat[i]) for ids[i] >= 0, by a stable
    sort + reduceat (ufunc.at is two orders of magnitude slower)."""
    ok = ids >= 0
    if not ok.all():
        flat, ids = flat[ok], ids[ok]
    if ids.size == 0:
        return out
    order = np.argsort(ids, kind="stable")
    s = ids[order]
    starts = np.flatnonzero(np.concatenate([[True], s[1:] != s[:-1]]))
    red = ufunc.reduceat(flat[order], starts, axis=0)
    out[s[starts]] = ufunc(out[s[starts]], red)
    return out


@_op("UnsortedSegmentMax")
def _unsorted_segment_max(g, n, data, ids, num):
    """tensorflow/core/kernels/segment_reduction_ops.cc: output initialised to
    numeric_limits<T>::lowest(), negative ids dropped."""
    data = np.asarray(data)
    ids = np.asarray(ids).astype(np.int64).reshape(-1)
    num = int(np.asarray(num))
    flat = data.reshape((ids.shape[0], -1))
    out = np.full((num, flat.shape[1]), np.finfo(data.dtype).min
                  if np.issubdtype(data.dtype, np.floating)
                  else np.iinfo(data.dtype).min, dtype=data.dtype)
    _segment_reduce(np.maximum, flat, ids, out)
    return (out.reshape((num,) + data.shape[1:]),)


@_op("UnsortedSegmentSum")
def _unsorted_segment_sum(g, n, data, ids, num):
    data = np.asarray(data)
    ids = np.asarray(ids).astype(np.int64)
    num = int(np.asarray(num))
    flat = data.reshape((ids.size, -1))
    out = np.zeros((num, flat.shape[1]), dtype=data.dtype)
    _segment_reduce(np.add, flat, ids.reshape(-1), out)
    return (out.reshape((num,) + data.shape[ids.ndim:]),)


@_op("BroadcastGradientArgs")
def _bcast_grad_args(g, n, s0, s1):
    s0 = [int(x) for x in np.asarray(s0)]
    s1 = [int(x) for x in np.asarray(s1)]
    r = max(len(s0), len(s1))
    p0 = [1] * (r - len(s0)) + s0
    p1 = [1] * (r - len(s1)) + s1
    r0 = [i for i in range(r) if p0[i] == 1 and not (p1[i] == 1 and
                                                     i >= r - len(s0))
          or i < r - len(s0)]
    r1 = [i for i in range(r) if p1[i] == 1 and not (p0[i] == 1 and
                                                     i >= r - len(s1))
          or i < r - len(s1)]
    # TF reduces along every axis where the operand has size 1 (even if the
    # other also has 1): summing a size-1 axis is the identity, so both
    # conventions give the same gradient.
    r0 = sorted(set(r0) | {i for i in range(r) if p0[i] == 1})
    r1 = sorted(set(r1) | {i for i in range(r) if p1[i] == 1})
    return (np.array(r0, dtype=np.int32), np.array(r1, dtype=np.int32))


@_op("DynamicStitch")
def _dynamic_stitch(g, n, *xs):
    k = len(xs) // 2
    idx, data = xs[:k], xs[k:]
    size = max(int(np.max(i)) for i in idx if np.size(i)) + 1
    first = np.asarray(data[0])
    tail = first.shape[np.asarray(idx[0]).ndim:]
    out = np.zeros((size,) + tail, dtype=first.dtype)
    for i, d in zip(idx, data):
        i = np.asarray(i).reshape(-1)
        out[i] = np.asarray(d).reshape((i.size,) + tail)
    return (out,)


@_op("TopKV2")
def _topk(g, n, a, k):
    k = int(np.asarray(k))
    a = np.asarray(a)
    order = np.argsort(-a, axis=-1, kind="stable")[..., :k]
    return (np.take_along_axis(a, order, axis=-1), order.astype(np.int32))


@_op("UniqueWithCounts")
def _unique_counts(g, n, a):
    a = np.asarray(a)
    vals, first, inv, cnt = np.unique(a, return_index=True,
                                      return_inverse=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    dt = n.dtype("out_idx") if n.attr("out_idx") else np.int32
    return (vals[order], rank[inv].astype(dt), cnt[order].astype(dt))


@_op("RandomUniform")
def _random_uniform(g, n, shape):
    # variable initialisers only (never on a fetched path with fed variables)
    rng = np.random.RandomState(n.attr("seed2", 0) or 0)
    return (rng.uniform(size=tuple(int(s) for s in np.asarray(shape))
                        ).astype(n.dtype("dtype")),)


@_op("DenseToDenseSetOperation")
def _set_op(g, n, a, b):
    """tf.sets.* on the last axis -> SparseTensor (indices, values, shape)."""
    a = np.asarray(a)
    b = np.asarray(b)
    kind = n.attr("set_operation").decode()
    lead = a.shape[:-1]
    rows_a = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(1, -1)
    rows_b = b.reshape(-1, b.shape[-1]) if b.ndim > 1 else b.reshape(1, -1)
    idx = []
    vals = []
    width = 0
    for r in range(rows_a.shape[0]):
        sa, sb = set(rows_a[r].tolist()), set(rows_b[r].tolist())
        if kind == "a-b":
            s = sa - sb
        elif kind == "b-a":
            s = sb - sa
        elif kind == "intersection":
            s = sa & sb
        else:
            s = sa | sb
        s = sorted(s)
        width = max(width, len(s))
        lead_idx = np.unravel_index(r, lead) if lead else ()
        for j, v in enumerate(s):
            idx.append(tuple(lead_idx) + (j,))
            vals.append(v)
    idx = np.array(idx, dtype=np.int64).reshape(-1, len(lead) + 1)
    return (idx, np.array(vals, dtype=a.dtype),
            np.array(tuple(lead) + (width,), dtype=np.int64))


# ------------------------------------------------------------------- graph

_STATEFUL = ("Assign", "AssignAdd", "AssignSub", "ApplyGradientDescent")


class Graph(object):
    """One `GraphDef` plus variable state.  `variables` maps variable node name
    (e.g. 'layer1/extract_vertex_features/fully_connected/weights') to ndarray;
    reading an unset variable raises KeyError (no silent initialisers)."""

    def __init__(self, nodes, collections=None):
        self.nodes = {n.name: n for n in nodes}
        self.order = [n.name for n in nodes]
        self.collections = collections or {}
        self.variables = {}

    # -- variables ---------------------------------------------------------
    def variable_nodes(self):
        return [n for n in (self.nodes[k] for k in self.order)
                if n.op in ("VariableV2", "Variable", "VarHandleOp")]

    def variable_shape(self, name):
        return self.nodes[name].attr("shape")[1]

    def trainable_variables(self):
        """Names in `tf.trainable_variables()` order (the
        'trainable_variables' collection of the MetaGraphDef)."""
        return list(self.collections.get("trainable_variables", []))

    def set_variables(self, values):
        for k, v in values.items():
            if k in self.nodes and self.nodes[k].op == "VariableV2":
                want = self.variable_shape(k)
                v = np.asarray(v, dtype=self.nodes[k].dtype("dtype"))
                if tuple(v.shape) != tuple(want):
                    raise ValueError("%s: shape %s, graph has %s" % (
                        k, v.shape, want))
                self.variables[k] = v.copy()

    # -- evaluation --------------------------------------------------------
    def run(self, fetches, feed_dict=None):
        """Evaluate tensors / ops by name ('node', 'node:1').  Each call is one
        `session.run`: stateful ops execute at most once, in dependency order."""
        feed = {}
        for k, v in (feed_dict or {}).items():
            feed[_split_input(k)[0]] = v
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        cache = {}
        for nm in names:
            self._eval(_split_input(nm)[0], cache, feed)
        out = []
        for nm in names:
            node, k = _split_input(nm)
            vals = cache[node]
            out.append(None if (k is None or not vals) and
                       self.nodes[node].op in ("NoOp", "Assert")
                       else (vals if vals is DEAD else vals[k or 0]))
        return out[0] if single else out

    def _eval(self, root, cache, feed):
        """Iterative post-order evaluation (graphs are ~10^4 nodes deep)."""
        stack = [(root, False)]
        while stack:
            name, ready = stack.pop()
            if name in cache:
                continue
            node = self.nodes[name]
            if name in feed:
                v = np.asarray(feed[name])
                if node.op in ("Placeholder", "PlaceholderWithDefault"):
                    v = v.astype(node.dtype("dtype"))
                cache[name] = (v,)
                continue
            deps = [_split_input(s) for s in node.inputs]
            if node.op in ("VariableV2", "Variable"):
                cache[name] = (self._read_var(name),)
                continue
            if node.op == "Placeholder":
                raise KeyError("placeholder %s is not fed" % name)
            if not ready:
                stack.append((name, True))
                # reversed: inputs are evaluated left to right
                for d, _ in reversed(deps):
                    if d not in cache:
                        stack.append((d, False))
                continue
            cache[name] = self._exec(node, deps, cache)

    def _read_var(self, name):
        if name not in self.variables:
            raise KeyError("variable %s has no value (set_variables first)"
                           % name)
        return self.variables[name]

    def _exec(self, node, deps, cache):
        args = []
        dead = False
        for d, k in deps:
            vals = cache[d]
            if vals is DEAD:
                dead = True
                continue
            if k is None:
                continue
            v = vals[k]
            if v is DEAD:
                dead = True
            args.append(v)
        op = node.op
        if op == "Merge":
            live = [a for a in args if a is not DEAD]
            if not live:
                return DEAD
            idx = [i for i, a in enumerate(args) if a is not DEAD][0]
            return (live[0], np.array(idx, dtype=np.int32))
        if dead:
            return DEAD
        if op == "Switch":
            data, pred = args
            return (DEAD, data) if bool(pred) else (data, DEAD)
        if op == "PlaceholderWithDefault":
            return (args[0],)
        if op in _STATEFUL:
            return self._exec_stateful(node, deps, args)
        if op in ("SaveV2", "RestoreV2"):
            raise NotImplementedError("checkpoint IO ops are not evaluated")
        kern = _KERNELS.get(op)
        if kern is None:
            raise NotImplementedError("op %s (node %s)" % (op, node.name))
        return kern(self, node, *args)

    def _exec_stateful(self, node, deps, args):
        var = deps[0][0]
        # the ref input of Assign* may pass through Identity-free ref edges only
        while self.nodes[var].op not in ("VariableV2", "Variable"):
            var = _split_input(self.nodes[var].inputs[0])[0]
        if node.op == "Assign":
            new = np.asarray(args[1]).astype(self.nodes[var].dtype("dtype"))
        elif node.op == "AssignAdd":
            new = self._read_var(var) + args[1]
        elif node.op == "AssignSub":
            new = self._read_var(var) - args[1]
        else:  # ApplyGradientDescent: var -= alpha * delta
            alpha, delta = args[1], args[2]
            new = self._read_var(var) - alpha * delta
        new = np.asarray(new, dtype=self.nodes[var].dtype("dtype"))
        self.variables[var] = new
        return (new,)


def load_meta(path):
    """Parse a `tf.train.Saver` .meta file (MetaGraphDef) -> Graph."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    graph_def = None
    collections = {}
    for f, _, v in _fields(buf):
        if f == 2:
            graph_def = v
        elif f == 4:                         # map<string, CollectionDef>
            key = val = None
            for g, _, w in _fields(v):
                if g == 1:
                    key = bytes(w).decode()
                elif g == 2:
                    val = w
            collections[key] = _parse_collection(val)
    if graph_def is None:
        raise ValueError("%s holds no GraphDef" % path)
    nodes = [Node(v) for f, _, v in _fields(graph_def) if f == 1]
    return Graph(nodes, collections)


def _parse_collection(buf):
    """CollectionDef.  Variables are stored as bytes_list of serialized
    VariableDef {1: variable_name 'x:0', ...}; ops as node_list."""
    out = []
    for f, _, v in _fields(buf):
        if f == 1:                           # node_list
            out += [bytes(w).decode() for g, _, w in _fields(v) if g == 1]
        elif f == 2:                         # bytes_list
            for g, _, w in _fields(v):
                if g != 1:
                    continue
                try:
                    name = None
                    for h, wt, x in _fields(w):
                        if h == 1 and wt == 2:
                            name = bytes(x).decode()
                            break
                    out.append(name.rsplit(":", 1)[0] if name else bytes(w))
                except Exception:            # not a VariableDef
                    out.append(bytes(w))
    return out


if __name__ == "__main__":
    gr = load_meta(sys.argv[1])
    import collections as _c
    print(len(gr.nodes), "nodes")
    for op, c in _c.Counter(n.op for n in gr.nodes.values()).most_common():
        print("%6d %s%s" % (c, op, "" if op in _KERNELS or op in _STATEFUL or
                            op in ("Switch", "Merge", "Placeholder",
                                   "VariableV2", "PlaceholderWithDefault")
                            else "   <-- no kernel"))
