"""Test-only writer of TensorFlow object-graph checkpoints (TensorBundle .index/.data + 'checkpoint' state file).

Written independently of human-pose-estimation_amd/tf_checkpoint.py from the same public format descriptions
(LevelDB table_format.md, tensor_bundle.proto, trackable_object_graph.proto) so that the reader is exercised on byte
streams it did not produce itself: prefix-compressed keys with restarts every 16 entries, several data blocks, block
trailers with masked CRC-32C, BundleEntryProto with fixed32 crc, string tensors (length varints + length checksum).
It is NOT TensorFlow: parity of the reader with real checkpoints stays unpinned (see the reader's header).
"""
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
_DT = {"float32": 1, "float64": 2, "int32": 3, "int64": 9, "bool": 10}


def _crc32c(data):
    # bitwise (different code path from the reader's table version)
    crc = 0xFFFFFFFF
    for b in bytes(data):
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def _fast_crc32c(data):
    # the bitwise loop above is the independent implementation (the reader's crc32c is checked against it on random
    # buffers in test_tf_checkpoint.py); megabyte tensors borrow the reader's vectorised one to keep the suite short
    if len(data) <= 4096:
        return _crc32c(data)
    from hpe_amd.tf_checkpoint import crc32c

    return crc32c(data)


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(fn, wt, payload):
    tag = _vi((fn << 3) | wt)
    if wt == 0:
        return tag + _vi(payload)
    if wt == 2:
        return tag + _vi(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.n, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.n % self.ri == 0 and self.n:
            self.restarts.append(len(self.buf))
        elif self.n:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _vi(shared) + _vi(len(key) - shared) + _vi(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def finish(self):
        out = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        return out


def _write_table(path, items, block_size=1024):
    """items: sorted [(key bytes, value bytes)]"""
    f = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def emit(block_bytes):
        off = len(f)
        f.extend(block_bytes)
        f.append(0)  # no compression
        f.extend(struct.pack("<I", _mask(_crc32c(block_bytes + b"\0"))))
        return off, len(block_bytes)

    bb, last_key = _BlockBuilder(), None
    for k, v in items:
        bb.add(k, v)
        last_key = k
        if len(bb.buf) >= block_size:
            off, size = emit(bb.finish())
            index.add(last_key, _vi(off) + _vi(size))
            bb = _BlockBuilder()
    if bb.n:
        off, size = emit(bb.finish())
        index.add(last_key, _vi(off) + _vi(size))
    moff, msize = emit(_BlockBuilder().finish())  # empty meta-index
    ioff, isize = emit(index.finish())
    footer = _vi(moff) + _vi(msize) + _vi(ioff) + _vi(isize)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    f.extend(footer)
    with open(path, "wb") as fh:
        fh.write(f)


def _shape_proto(shape):
    return b"".join(_field(2, 2, _field(1, 0, d)) for d in shape)


def write_bundle(prefix, tensors):
    """tensors: {key: ndarray | bytes (scalar string tensor)}"""
    data = bytearray()
    items = [(b"", _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1)))]  # num_shards=1, little endian (default), version.producer=1
    for key in sorted(tensors):
        t = tensors[key]
        off = len(data)
        if isinstance(t, bytes):
            lens = _vi(len(t))
            payload = lens + struct.pack("<I", _mask(_crc32c(lens))) + t
            entry = _field(1, 0, 7) + _field(2, 2, b"")
            crc = None
        else:
            t = np.asarray(t)
            payload = t.tobytes()  # C order
            entry = _field(1, 0, _DT[t.dtype.name]) + _field(2, 2, _shape_proto(t.shape))
            crc = _fast_crc32c(payload)
        data += payload
        if off:
            entry += _field(4, 0, off)
        entry += _field(5, 0, len(payload))
        entry += _field(6, 5, _mask(crc) if crc is not None else 0)  # strings: checked through their length checksum
        items.append((key.encode(), entry))
    _write_table(prefix + ".index", items)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)


def object_graph(tree):
    """tree: nested dict; a leaf is (full_name, checkpoint_key).  Returns the serialized TrackableObjectGraph with node 0
    = root, nodes numbered breadth-first like TF does."""
    nodes = [tree]
    protos = []
    i = 0
    while i < len(nodes):
        n = nodes[i]
        body = b""
        if isinstance(n, dict):
            for name, child in n.items():
                nodes.append(child)
                body += _field(1, 2, _field(1, 0, len(nodes) - 1) + _field(2, 2, name.encode()))
        else:
            full, ckey = n
            body += _field(2, 2, _field(1, 2, b"VARIABLE_VALUE") + _field(2, 2, full.encode()) + _field(3, 2, ckey.encode()))
        protos.append(_field(1, 2, body))
        i += 1
    return b"".join(protos)


def write_checkpoint_state(ckpt_dir, name):
    with open(os.path.join(ckpt_dir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
