"""Pure-Python reader for TensorFlow "bundle" checkpoints (no TensorFlow).

The reference saves/restores its weights with `tf.train.Saver`
(run.py:192-201, train.py:496,634-636).  A checkpoint `model-N` is
  * `model-N.index`  -- a LevelDB-style SSTable: 48-byte footer (metaindex and
    index block handles as varint64 pairs, zero padding, 8-byte magic
    0xdb4775248b80fb57), prefix-compressed key/value blocks each followed by a
    5-byte trailer (compression type + crc), values = `BundleEntryProto`
    {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c};
    the empty key holds the `BundleHeaderProto`;
  * `model-N.data-00000-of-00001` -- raw little-endian tensor bytes.
Only uncompressed blocks (what TF writes) are supported.
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


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


def _block_entries(buf, offset, size):
    """Yield (key, value) from one SSTable block at [offset, offset+size)."""
    if buf[offset + size] != 0:
        raise ValueError("compressed SSTable blocks are not supported")
    block = buf[offset:offset + size]
    n_restarts = struct.unpack_from("<I", block, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _parse_proto(buf):
    """Minimal protobuf wire parser -> {field: [values]} (varint / len / fixed)."""
    out = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _shape(shape_bytes):
    dims = []
    for d in _parse_proto(shape_bytes).get(2, []):
        dims.append(_parse_proto(d).get(1, [0])[0])
    return tuple(dims)


def list_variables(prefix):
    """[(name, dtype, shape, offset, size)] in key order for checkpoint `prefix`
    (e.g. '.../model-1400000')."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError("not a TF bundle index: bad magic")
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _varint(footer, pos)       # metaindex offset
    _, pos = _varint(footer, pos)       # metaindex size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(buf, idx_off, idx_size):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for key, val in _block_entries(buf, boff, bsize):
            if key == b"":
                continue                # BundleHeaderProto
            e = _parse_proto(val)
            dtype = _DTYPES[e.get(1, [0])[0]]
            shape = _shape(e[2][0]) if 2 in e else ()
            out.append((key.decode(), dtype, shape, e.get(4, [0])[0],
                        e.get(5, [0])[0]))
    return out


def load_checkpoint(ckpt_dir_or_prefix):
    """{variable name: ndarray}.  Accepts a checkpoint directory (uses its
    `checkpoint` file's `model_checkpoint_path`, like
    tf.train.latest_checkpoint at run.py:199) or an explicit prefix."""
    prefix = ckpt_dir_or_prefix
    if os.path.isdir(prefix):
        name = None
        with open(os.path.join(prefix, "checkpoint")) as f:
            for line in f:
                if line.startswith("model_checkpoint_path:"):
                    name = line.split(":", 1)[1].strip().strip('"')
        if name is None:
            raise ValueError("no model_checkpoint_path in %s" % prefix)
        prefix = os.path.join(prefix, os.path.basename(name))
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        data = f.read()
    out = {}
    for name, dtype, shape, off, size in list_variables(prefix):
        arr = np.frombuffer(data, dtype=dtype, count=size // np.dtype(dtype).itemsize,
                            offset=off)
        out[name] = arr.reshape(shape).copy()
    return out
