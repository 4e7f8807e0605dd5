"""Stand-alone reader for the TensorFlow-2 object-graph checkpoints the reference restores its weights from
 -- no TensorFlow involved.

What the reference does (src/predictor.py:77-86, same in src/trainer.py:192-198,836):

    checkpoint = tf.train.Checkpoint(generator_optimizer=..., discriminator_optimizer=...,
                                     feature_extractor=<Keras ResNet50>, generator3d=<Sequential of 3 Dense>,
                                     discriminator=..., inital_theta=<Variable [1,85]>)
    checkpoint.restore(tf.train.latest_checkpoint(checkpoint_dir)).expect_partial()

On disk that is ``<dir>/checkpoint`` (a text file naming the newest prefix) plus a *TensorBundle*
``ckpt-N.index`` / ``ckpt-N.data-00000-of-00001``:

  * the index is a LevelDB-format sorted string table: data blocks of prefix-compressed key/value entries with a
    restart array, each block followed by a 1-byte compression tag and a masked CRC-32C; a 48-byte footer holds the
    handles of the meta-index and index blocks and the magic 0xdb4775248b80fb57;
  * key "" -> BundleHeaderProto (num_shards, endianness); every other key -> BundleEntryProto
    (dtype, shape, shard_id, offset, size, masked crc32c) pointing into a data shard of raw little-endian tensors;
  * key ``_CHECKPOINTABLE_OBJECT_GRAPH`` is a string tensor holding a TrackableObjectGraph proto: the tree of
    Python attribute names (``feature_extractor`` -> ``layer_with_weights-3`` -> ``kernel``) with, per variable, its
    ``full_name`` (the Keras variable name, e.g. ``res2a_branch2a/kernel``) and its ``checkpoint_key``.

``load_hmr_weights`` walks that graph, so Keras layer names -- not positional guesses -- decide which tensor feeds
which ``hpe_load_conv`` / ``hpe_load_dense`` slot; when a checkpoint has no graph entry the positional order of
keras_applications 1.0.8 / tf.keras >= 2.2 is used and cross-checked against every kernel shape.

PARITY UNPINNED: no TensorFlow-written checkpoint exists in this environment (the trained weights are "contact the
authors", README.md:76) -- the reader follows the published formats (LevelDB table_format.md; tensor_bundle.proto;
trackable_object_graph.proto) and is tested against bundles produced by the independent writer in
tests/tf_bundle_writer.py.
"""
from __future__ import annotations

import os
import re

import numpy as np

from .resnet_spec import CONV_SPECS

TABLE_MAGIC = 0xDB4775248B80FB57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VAR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"

_DTYPES = {1: "<f4", 2: "<f8", 3: "<i4", 4: "u1", 5: "<i2", 6: "i1", 9: "<i8", 10: "?", 17: "<u2", 19: "<f2", 22: "<u4", 23: "<u8"}
DT_STRING = 7


class CheckpointError(ValueError):
    pass


def _guard(fn):
    """A damaged index (bad UTF-8 in a key, a field with the wrong wire type, lengths past the end ...) surfaces as
    CheckpointError, never as a stray exception type."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except (CheckpointError, FileNotFoundError, KeyError):
            raise
        except (ValueError, TypeError, IndexError, OverflowError, MemoryError, UnicodeError) as e:
            raise CheckpointError("corrupt checkpoint structure: %s: %s" % (type(e).__name__, e)) from e

    return wrapped


# ----------------------------------------------------------------------------------------------- CRC-32C (Castagnoli)
def _crc_table():
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_NP = _crc_table()
_CRC_T = [int(x) for x in _CRC_NP]


def _crc_bytewise(data, state):
    t = _CRC_T
    for b in bytes(data):
        state = t[(state ^ b) & 0xFF] ^ (state >> 8)
    return state


def _mat_apply(cols, v):
    """GF(2) 32x32 matrix (32 uint32 columns) times vector(s): xor of the columns selected by the bits of v."""
    if isinstance(v, np.ndarray):
        out = np.zeros_like(v)
        for bit in range(32):
            out ^= np.where((v >> np.uint32(bit)) & np.uint32(1), np.uint32(cols[bit]), np.uint32(0)).astype(np.uint32)
        return out
    out = 0
    for bit in range(32):
        if (v >> bit) & 1:
            out ^= cols[bit]
    return out


def _mat_square(cols):
    return [_mat_apply(cols, c) for c in cols]


_ZERO_BYTE = [_crc_bytewise(b"\0", 1 << bit) for bit in range(32)]  # register update for one zero byte, as a matrix


_SHIFT_POW2 = [_ZERO_BYTE]  # [k] = matrix advancing the register over 2^k zero bytes


def _shift_pow2(k):
    while len(_SHIFT_POW2) <= k:
        _SHIFT_POW2.append(_mat_square(_SHIFT_POW2[-1]))
    return _SHIFT_POW2[k]


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected 0x82F63B78).  Large buffers use the register's linearity over GF(2): the initial
    register value is folded into the first four bytes, the buffer is zero-padded at the FRONT (a zero register stays
    zero over zero bytes) to lanes x L bytes, all lanes advance in lock-step with one vectorised table lookup per byte
    column, and lanes are merged pairwise with the zero-byte shift matrices of 2^k bytes."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data).reshape(-1).view(np.uint8)
    else:
        a = np.frombuffer(bytes(data), dtype=np.uint8)
    n = a.shape[0]
    init = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    if n < 8192:
        return _crc_bytewise(a.tobytes(), init) ^ 0xFFFFFFFF
    logL = 8 if n < (1 << 19) else (10 if n < (1 << 23) else 12)
    L = 1 << logL
    lanes = 1
    while lanes * L < n:
        lanes *= 2
    buf = np.zeros(lanes * L, dtype=np.uint8)
    buf[lanes * L - n:] = a
    head = buf[lanes * L - n:lanes * L - n + 4]
    head ^= np.frombuffer(init.to_bytes(4, "little"), dtype=np.uint8)
    cols = np.ascontiguousarray(buf.reshape(lanes, L).T)
    reg = np.zeros(lanes, dtype=np.uint32)
    m8, s8 = np.uint32(0xFF), np.uint32(8)
    for j in range(L):
        reg = _CRC_NP[(reg ^ cols[j]) & m8] ^ (reg >> s8)
    k = logL
    while reg.shape[0] > 1:
        reg = _mat_apply(_shift_pow2(k), reg[0::2]) ^ reg[1::2]
        k += 1
    return int(reg[0]) ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------------------------- varints / protobuf wire format
def _varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise CheckpointError("varint too long")


def _proto_fields(buf):
    """Yield (field_number, wire_type, value) -- value is int for varint/fixed, bytes for length-delimited."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = int.from_bytes(buf[pos:pos + 8], "little")
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise CheckpointError("truncated protobuf field")
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ----------------------------------------------------------------------------------------------- snappy (block format), decode only
def _snappy_decompress(src):
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):  # overlapping copies are legal
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# ----------------------------------------------------------------------------------------------- sorted string table (.index)
class _Table:
    def __init__(self, buf, verify=True):
        self.buf = buf
        self.verify = verify
        if len(buf) < 48:
            raise CheckpointError("index file too short for a table footer")
        footer = buf[-48:]
        if int.from_bytes(footer[40:], "little") != TABLE_MAGIC:
            raise CheckpointError("not a TensorBundle index (bad table magic)")
        _mo, p = _varint(footer, 0)
        _ms, p = _varint(footer, p)
        io, p = _varint(footer, p)
        isz, p = _varint(footer, p)
        self.index_handle = (io, isz)

    def _block(self, off, size):
        if off + size + 5 > len(self.buf):
            raise CheckpointError("block handle outside the file")
        raw = self.buf[off:off + size]
        ctype = self.buf[off + size]
        stored = int.from_bytes(self.buf[off + size + 1:off + size + 5], "little")
        if self.verify and mask_crc(crc32c(self.buf[off:off + size + 1])) != stored:
            raise CheckpointError("block checksum mismatch at offset %d" % off)
        if ctype == 1:
            raw = _snappy_decompress(raw)
        elif ctype != 0:
            raise CheckpointError("unknown block compression %d" % ctype)
        return raw

    @staticmethod
    def _entries(block):
        if len(block) < 4:
            raise CheckpointError("block too short")
        nrestart = int.from_bytes(block[-4:], "little")
        end = len(block) - 4 - 4 * nrestart
        if end < 0:
            raise CheckpointError("bad restart array")
        pos, key = 0, b""
        while pos < end:
            shared, pos = _varint(block, pos)
            non_shared, pos = _varint(block, pos)
            vlen, pos = _varint(block, pos)
            if shared > len(key) or pos + non_shared + vlen > end:
                raise CheckpointError("corrupt block entry")
            key = key[:shared] + block[pos:pos + non_shared]
            pos += non_shared
            yield key, block[pos:pos + vlen]
            pos += vlen

    def items(self):
        for _k, handle in self._entries(self._block(*self.index_handle)):
            off, p = _varint(handle, 0)
            size, _p = _varint(handle, p)
            yield from self._entries(self._block(off, size))


# ----------------------------------------------------------------------------------------------- TensorBundle
class _Entry:
    __slots__ = ("dtype", "shape", "shard", "offset", "size", "crc", "sliced")


class BundleReader:
    """``BundleReader(prefix)`` -- prefix as given to ``Checkpoint.save`` (``.../ckpt-7``)."""

    @_guard
    def __init__(self, prefix, verify=True):
        self.prefix = str(prefix)
        self.verify = verify
        ipath = self.prefix + ".index"
        if not os.path.exists(ipath):
            raise FileNotFoundError(ipath)
        with open(ipath, "rb") as f:
            table = _Table(f.read(), verify=verify is not False)
        self.entries = {}
        self.num_shards = None
        for k, v in table.items():
            if k == b"":
                for fn, _wt, val in _proto_fields(v):
                    if fn == 1:
                        self.num_shards = val
                    elif fn == 2 and val != 0:
                        raise CheckpointError("big-endian bundles are not supported")
                continue
            e = _Entry()
            e.dtype, e.shape, e.shard, e.offset, e.size, e.crc, e.sliced = 0, (), 0, 0, 0, None, False
            for fn, _wt, val in _proto_fields(v):
                if fn == 1:
                    e.dtype = val
                elif fn == 2:
                    dims = []
                    for fn2, _w2, v2 in _proto_fields(val):
                        if fn2 == 2:
                            size = 0
                            for fn3, _w3, v3 in _proto_fields(v2):
                                if fn3 == 1:
                                    size = _signed64(v3)
                            dims.append(size)
                        elif fn2 == 3 and v2:
                            raise CheckpointError("tensor of unknown rank in the bundle")
                    e.shape = tuple(dims)
                elif fn == 3:
                    e.shard = val
                elif fn == 4:
                    e.offset = val
                elif fn == 5:
                    e.size = val
                elif fn == 6:
                    e.crc = val
                elif fn == 7:
                    e.sliced = True
            self.entries[k.decode("utf-8")] = e
        if self.num_shards is None:
            raise CheckpointError("bundle header entry is missing")
        self._shards = {}

    def keys(self):
        return sorted(self.entries)

    def __contains__(self, key):
        return key in self.entries

    def shape(self, key):
        return self.entries[key].shape

    def _shard(self, i):
        if i not in self._shards:
            path = "%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards)
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            self._shards[i] = np.memmap(path, dtype=np.uint8, mode="r")
        return self._shards[i]

    def _raw(self, key):
        if key not in self.entries:
            raise KeyError("%s not in checkpoint %s" % (key, self.prefix))
        e = self.entries[key]
        if e.sliced:
            raise CheckpointError("%s: partitioned (sliced) variables are not supported" % key)
        shard = self._shard(e.shard)
        if e.offset + e.size > shard.shape[0]:
            raise CheckpointError("%s: data shard is shorter than the index says" % key)
        raw = shard[e.offset:e.offset + e.size]
        if self.verify is not False and e.crc is not None and e.dtype != DT_STRING and mask_crc(crc32c(raw)) != e.crc:
            raise CheckpointError("%s: tensor checksum mismatch" % key)
        return e, raw

    @_guard
    def get(self, key):
        e, raw = self._raw(key)
        if e.dtype == DT_STRING:
            return self._strings(key, e, bytes(raw))
        if e.dtype not in _DTYPES:
            raise CheckpointError("%s: unsupported dtype enum %d" % (key, e.dtype))
        dt = np.dtype(_DTYPES[e.dtype])
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if n * dt.itemsize != e.size:
            raise CheckpointError("%s: %d bytes stored for shape %s %s" % (key, e.size, e.shape, dt))
        return np.frombuffer(raw, dtype=dt).reshape(e.shape).copy()

    @staticmethod
    def _strings(key, e, raw):
        n = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        pos, lens = 0, []
        for _ in range(n):
            ln, pos = _varint(raw, pos)
            lens.append(ln)
        pos += 4  # masked crc32c of the lengths
        out = []
        for ln in lens:
            if pos + ln > len(raw):
                raise CheckpointError("%s: truncated string tensor" % key)
            out.append(raw[pos:pos + ln])
            pos += ln
        return out[0] if not e.shape else np.array(out, dtype=object).reshape(e.shape)


def latest_checkpoint(checkpoint_dir, latest_filename="checkpoint"):
    """``tf.train.latest_checkpoint``: the prefix named by ``model_checkpoint_path`` in <dir>/checkpoint, or None."""
    path = os.path.join(checkpoint_dir, latest_filename)
    if not os.path.exists(path):
        return None
    with open(path, "r") as f:
        m = re.search(r'^\s*model_checkpoint_path:\s*"((?:[^"\\]|\\.)*)"', f.read(), re.M)
    if not m:
        return None
    name = m.group(1).encode("utf-8").decode("unicode_escape")
    prefix = name if os.path.isabs(name) else os.path.join(checkpoint_dir, name)
    return prefix if os.path.exists(prefix + ".index") else None


# ----------------------------------------------------------------------------------------------- object graph
class ObjectGraph:
    """nodes[i] = {'children': {local_name: node_id}, 'attributes': [(name, full_name, checkpoint_key)]}"""

    @_guard
    def __init__(self, blob):
        self.nodes = []
        for fn, _wt, val in _proto_fields(blob):
            if fn != 1:
                continue
            node = {"children": {}, "attributes": []}
            for fn2, _w2, v2 in _proto_fields(val):
                if fn2 == 1:
                    nid, lname = 0, ""
                    for fn3, _w3, v3 in _proto_fields(v2):
                        if fn3 == 1:
                            nid = v3
                        elif fn3 == 2:
                            lname = v3.decode("utf-8")
                    node["children"][lname] = nid
                elif fn2 == 2:
                    name = full = ckey = ""
                    for fn3, _w3, v3 in _proto_fields(v2):
                        if fn3 == 1:
                            name = v3.decode("utf-8")
                        elif fn3 == 2:
                            full = v3.decode("utf-8")
                        elif fn3 == 3:
                            ckey = v3.decode("utf-8")
                    node["attributes"].append((name, full, ckey))
            self.nodes.append(node)
        if not self.nodes:
            raise CheckpointError("empty object graph")

    def child(self, node_id, name):
        return self.nodes[node_id]["children"].get(name)

    def variable(self, node_id):
        """(full_name, checkpoint_key) of a variable node, or None"""
        for name, full, ckey in self.nodes[node_id]["attributes"]:
            if name == "VARIABLE_VALUE":
                return full, ckey
        return None


# ----------------------------------------------------------------------------------------------- the reference's checkpoint -> Keras-layout dict
def keras_weighted_layer_order(shortcut_first=False):
    """Names of the ResNet50 layers that own weights in ``model.layers`` order = the ``layer_with_weights-N`` numbering.
    keras_applications 1.0.8 (TF 2.0, the reference's pin): ... 2c conv, shortcut conv, 2c bn, shortcut bn (the shortcut is
    the *second* input of the add); tf.keras >= 2.2 builds the shortcut first: shortcut conv, 2c conv, shortcut bn, 2c bn."""
    names = ["conv1", "bn_conv1"]
    by_block = {}
    for s in CONV_SPECS[1:]:
        by_block.setdefault(s.name.split("_branch")[0], []).append(s)
    for blk in by_block.values():
        d = {s.name.split("_branch")[1]: s for s in blk}
        names += [d["2a"].name, d["2a"].bn_name, d["2b"].name, d["2b"].bn_name]
        tail = [d["2c"]] + ([d["1"]] if "1" in d else [])
        if shortcut_first:
            tail.reverse()
        names += [s.name for s in tail] + [s.bn_name for s in tail]
    return names


_BN_VARS = ("gamma", "beta", "moving_mean", "moving_variance")
# tf.keras >= 2.2 layer names -> the keras_applications 1.0.8 names the C ABI uses (resnet_spec.CONV_SPECS)


def _modern_to_legacy(layer):
    if layer == "conv1_conv":
        return "conv1"
    if layer == "conv1_bn":
        return "bn_conv1"
    m = re.fullmatch(r"conv([2-5])_block(\d)_([0-3])_(conv|bn)", layer)
    if not m:
        return None
    stage, blk, idx, kind = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    branch = {0: "1", 1: "2a", 2: "2b", 3: "2c"}[idx]
    return "%s%d%s_branch%s" % ("res" if kind == "conv" else "bn", stage, "abcdef"[blk - 1], branch)


def _shape_ok(name, arr_shape):
    for s in CONV_SPECS:
        if s.name == name:
            return tuple(arr_shape) == (s.kh, s.kw, s.cin, s.cout)
    return True


def load_hmr_weights(prefix_or_dir, verify=True):
    """-> (weights, info).  ``weights`` is the Keras-layout dict ``HpeEngine.load_encoder`` / ``load_regressor`` take
    (``<layer>/kernel`` HWIO, ``<layer>/bias``, ``<bn>/gamma|beta|moving_mean|moving_variance``, ``dense_{0,1,2}/kernel|bias``)
    plus ``inital_theta`` [1,85] when the checkpoint has it (src/predictor.py:84 -- the attribute name is misspelt in the
    reference and therefore in its checkpoints).  ``info`` says how names were resolved."""
    prefix = str(prefix_or_dir)
    if os.path.isdir(prefix):
        p = latest_checkpoint(prefix)
        if p is None:
            raise FileNotFoundError("no TensorFlow checkpoint state in %s" % prefix)
        prefix = p
    rd = BundleReader(prefix, verify=verify)
    out, info = {}, {"prefix": prefix, "resolved_by": None}
    known = {s.name for s in CONV_SPECS} | {s.bn_name for s in CONV_SPECS}

    def put_layer(layer, var, ckey):
        out["%s/%s" % (layer, var)] = rd.get(ckey)

    graph = ObjectGraph(rd.get(OBJECT_GRAPH_KEY)) if OBJECT_GRAPH_KEY in rd else None
    enc_done = False
    if graph is not None:
        fe = graph.child(0, "feature_extractor")
        if fe is not None:
            for lname, nid in graph.nodes[fe]["children"].items():
                if not lname.startswith("layer_with_weights-"):
                    continue
                for var, vid in graph.nodes[nid]["children"].items():
                    v = graph.variable(vid)
                    if v is None:
                        continue
                    full, ckey = v
                    layer = full.split("/")[-2] if "/" in full else ""
                    layer = layer if layer in known else (_modern_to_legacy(layer) or layer)
                    if layer in known and ckey in rd:
                        put_layer(layer, var, ckey)
            enc_done = all(("%s/kernel" % s.name) in out and ("%s/gamma" % s.bn_name) in out for s in CONV_SPECS)
            if enc_done:
                info["resolved_by"] = "object graph (Keras variable names)"
    if not enc_done:
        # positional: feature_extractor/layer_with_weights-N in model.layers order; decide the order of the two equal-depth
        # 1x1 convs at a block end from a block where their shapes differ (res3a: shortcut [1,1,256,512] vs 2c [1,1,128,512])
        base = "feature_extractor/layer_with_weights-%d/%s" + VAR_SUFFIX
        for shortcut_first in (False, True):
            order = keras_weighted_layer_order(shortcut_first)
            i3 = order.index("res3a_branch1")
            k = base % (i3, "kernel")
            if k in rd and rd.shape(k) == (1, 1, 256, 512):
                break
        else:
            raise CheckpointError("feature_extractor weights not found in %s (neither by name nor by position)" % prefix)
        for i, layer in enumerate(order):
            for var in (("kernel", "bias") if not layer.startswith("bn") else _BN_VARS):
                put_layer(layer, var, base % (i, var))
        info["resolved_by"] = "position (%s order)" % ("tf.keras>=2.2" if shortcut_first else "keras_applications 1.0.8")
    for s in CONV_SPECS:
        if not _shape_ok(s.name, out[s.name + "/kernel"].shape):
            raise CheckpointError("%s/kernel has shape %s, expected %s" % (s.name, out[s.name + "/kernel"].shape, (s.kh, s.kw, s.cin, s.cout)))
    # regressor: a Sequential -- its three Dense layers are layer_with_weights-0..2 whatever their auto-generated names
    for i, name in enumerate(("dense_0", "dense_1", "dense_2")):
        for var in ("kernel", "bias"):
            ckey = "generator3d/layer_with_weights-%d/%s%s" % (i, var, VAR_SUFFIX)
            if graph is not None:
                g3 = graph.child(0, "generator3d")
                nid = graph.child(g3, "layer_with_weights-%d" % i) if g3 is not None else None
                vid = graph.child(nid, var) if nid is not None else None
                v = graph.variable(vid) if vid is not None else None
                if v is not None:
                    ckey = v[1]
            out["%s/%s" % (name, var)] = rd.get(ckey)
    ckey = "inital_theta" + VAR_SUFFIX
    if ckey in rd:
        out["inital_theta"] = rd.get(ckey)
    return out, info
