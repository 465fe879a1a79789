"""Pure-Python reader and writer for TensorFlow "bundle" checkpoints (no
TensorFlow).  `save_checkpoint` re-encodes the reference's shipped checkpoints
bit for bit (tests/test_host_cpu.py), so files written here restore in TF.

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


# ---------------------------------------------------------------------------
# writer (train.py:634-636 `saver.save`): same on-disk format, so a checkpoint
# written here restores in the reference's run.py / train.py unchanged.
# ---------------------------------------------------------------------------
_DTYPE_ENUM = {np.dtype(np.float32): 1, np.dtype(np.float64): 2,
               np.dtype(np.int32): 3, np.dtype(np.int64): 9}
_BLOCK_SIZE = 262144       # tensorflow/core/lib/io/table_options.h
_RESTART_INTERVAL = 16


def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of TF's bundle / table files."""
    tab = _CRC_TAB
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(crc):
    """crc32c::Mask of tensorflow/core/lib/hash/crc32c.h."""
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _entry_proto(dtype, shape, offset, size, crc_masked):
    """BundleEntryProto; proto3 default values (offset 0, shard 0) are omitted
    exactly like the C++ serializer does."""
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in
                    [b"\x08" + _put_varint(int(s)) for s in shape])
    out = b"\x08" + _put_varint(_DTYPE_ENUM[np.dtype(dtype)])
    out += b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size)
    out += b"\x35" + struct.pack("<I", crc_masked)
    return out


class _BlockBuilder(object):
    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""

    def add(self, key, value):
        shared = 0
        if self.counter < _RESTART_INTERVAL:
            n = min(len(self.last_key), len(key))
            while shared < n and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + \
            _put_varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        out = bytes(self.buf)
        for r in self.restarts:
            out += struct.pack("<I", r)
        return out + struct.pack("<I", len(self.restarts))


def _short_separator(a, b):
    """BytewiseComparator::FindShortestSeparator."""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _short_successor(a):
    """BytewiseComparator::FindShortSuccessor."""
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def _read_state(ckpt_dir):
    """all_model_checkpoint_paths of an existing `checkpoint` state file."""
    paths = []
    try:
        with open(os.path.join(ckpt_dir, "checkpoint")) as f:
            for line in f:
                if line.startswith("all_model_checkpoint_paths:"):
                    paths.append(line.split('"')[1])
    except (IOError, OSError, IndexError):
        pass
    return paths


def save_checkpoint(ckpt_dir, variables, global_step=None, name="model",
                    max_to_keep=5):
    """Write `<ckpt_dir>/<name>-<global_step>.{index,data-00000-of-00001}` and
    the `checkpoint` state file (tf.train.Saver.save with one shard).
    variables: {TF variable name: ndarray}; when `global_step` is given it is
    also stored as the int32 scalar `Variable`, as the reference's graph does
    (train.py:375).  Like tf.train.Saver (default max_to_keep = 5, which
    train.py:496 uses) the state file lists the checkpoints kept so far, oldest
    first, and the ones beyond `max_to_keep` are deleted (None / 0 keeps
    all), so `saver.recover_last_checkpoints` of the reference (train.py:516)
    sees them.  Returns the checkpoint prefix."""
    os.makedirs(ckpt_dir, exist_ok=True)
    items = {k: np.ascontiguousarray(v) for k, v in variables.items()}
    if global_step is not None and "Variable" not in items:
        items["Variable"] = np.array(global_step, dtype=np.int32)
    base = name if global_step is None else "%s-%d" % (name, int(global_step))
    prefix = os.path.join(ckpt_dir, base)
    keys = sorted(items, key=lambda s: s.encode())
    data = bytearray()
    entries = [(b"", b"\x08\x01\x1a\x02\x08\x01")]   # BundleHeaderProto
    for k in keys:
        arr = items[k]
        if arr.dtype.byteorder == '>':
            arr = arr.astype(arr.dtype.newbyteorder('<'))
        raw = arr.tobytes()
        entries.append((k.encode(), _entry_proto(
            arr.dtype, arr.shape, len(data), len(raw),
            _mask_crc(crc32c(raw)))))
        data += raw
    # SSTable: data blocks, empty metaindex block, index block, footer
    out = bytearray()
    index = _BlockBuilder()

    def write_block(contents):
        handle = _put_varint(len(out)) + _put_varint(len(contents))
        trailer_crc = _mask_crc(crc32c(b"\x00", crc32c(contents)))
        out.extend(contents + b"\x00" + struct.pack("<I", trailer_crc))
        return handle

    block = _BlockBuilder()
    pending = None      # (last key of the finished block, its handle)
    for key, val in entries:
        if pending is not None:
            index.add(_short_separator(pending[0], key), pending[1])
            pending = None
        block.add(key, val)
        if block.size_estimate() >= _BLOCK_SIZE:
            pending = (block.last_key, write_block(block.finish()))
            block = _BlockBuilder()
    if block.buf:
        pending = (block.last_key, write_block(block.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = write_block(_BlockBuilder().finish())
    index_handle = write_block(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out += footer
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    kept = [p for p in _read_state(ckpt_dir) if p != base] + [base]
    if max_to_keep:
        for old in kept[:-max_to_keep]:
            for suffix in (".index", ".data-00000-of-00001"):
                try:
                    os.remove(os.path.join(ckpt_dir, old + suffix))
                except OSError:
                    pass
        kept = kept[-max_to_keep:]
    with open(os.path.join(ckpt_dir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % base)
        for p in kept:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    return prefix
