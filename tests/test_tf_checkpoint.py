"""TF object-graph checkpoint reader (reference: src/predictor.py:77-86).  PARITY UNPINNED: no
TensorFlow-written checkpoint exists here; bundles come from the independent writer in tests/tf_bundle_writer.py."""
import os

import numpy as np
import pytest

import tf_bundle_writer as W
from hpe_amd import synthetic, tf_checkpoint as T
from hpe_amd.resnet_spec import CONV_SPECS

SUF = T.VAR_SUFFIX


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C
    assert T.crc32c(b"\0" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"123456789") == 0xE3069283
    assert W._crc32c(b"123456789") == 0xE3069283
    rng = np.random.default_rng(0)
    for n in (0, 1, 4095, 16384, 16385, 70001, 300000):  # vectorised path (>= 16 KiB) against the bitwise loop
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(b) == W._crc32c(b), n
    assert T.crc32c(b"world", T.crc32c(b"hello ")) == T.crc32c(b"hello world")
    big = rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()
    assert T.crc32c(big[20000:], T.crc32c(big[:20000])) == T.crc32c(big)
    # LevelDB's mask: documented example crc32c("foo") masked is not the identity and is invertible
    c = T.crc32c(b"foo")
    m = T.mask_crc(c)
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    assert ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF == c and m != c


def test_snappy_decoder():
    # hand-assembled stream: literal "abcd", copy(off=4,len=4) with 1-byte offset, literal "XY", 2-byte-offset copy len 6 off 10
    body = bytes([3 << 2]) + b"abcd" + bytes([((4 - 4) << 2) | 1 | (0 << 5), 4]) + bytes([1 << 2]) + b"XY" + bytes([((6 - 1) << 2) | 2, 10, 0])
    want = b"abcdabcdXY" + b"abcdab"
    assert T._snappy_decompress(bytes([len(want)]) + body) == want
    # overlapping copy (run-length): literal "z" then copy off=1 len=7
    assert T._snappy_decompress(bytes([8, 0 << 2]) + b"z" + bytes([((7 - 4) << 2) | 1, 1])) == b"z" * 8


def _hmr_tensors(enc, reg, order, modern_names=False, with_graph=True, seed=0):
    rng = np.random.default_rng(seed)
    tensors, tree = {}, {}
    fe = {}
    for i, layer in enumerate(order):
        node = {}
        for var in (("kernel", "bias") if not layer.startswith("bn") else T._BN_VARS):
            ckey = "feature_extractor/layer_with_weights-%d/%s%s" % (i, var, SUF)
            tensors[ckey] = enc["%s/%s" % (layer, var)]
            node[var] = ("%s/%s" % (layer, var), ckey)
        fe["layer_with_weights-%d" % i] = node
        fe["layer-%d" % (2 * i + 1)] = node  # TF lists every layer twice (layer-N and layer_with_weights-M)
    g3 = {}
    for i in range(3):
        node = {}
        for var in ("kernel", "bias"):
            ckey = "generator3d/layer_with_weights-%d/%s%s" % (i, var, SUF)
            tensors[ckey] = reg["dense_%d/%s" % (i, var)]
            node[var] = ("dense_%d/%s" % (i + 7, var), ckey)  # auto-generated Keras names need not start at 0
        g3["layer_with_weights-%d" % i] = node
    tensors["inital_theta" + SUF] = rng.normal(size=(1, 85)).astype(np.float32)
    tensors["save_counter" + SUF] = np.array(3, dtype=np.int64)
    # optimizer slots share the variable's key prefix (must not confuse the walk)
    k0 = "feature_extractor/layer_with_weights-0/kernel/.OPTIMIZER_SLOT/generator_optimizer/m" + SUF
    tensors[k0] = np.zeros((7, 7, 3, 64), np.float32)
    tensors["discriminator/layer_with_weights-0/kernel" + SUF] = rng.normal(size=(10, 5)).astype(np.float32)
    tree = {"feature_extractor": fe, "generator3d": g3, "inital_theta": ("Variable", "inital_theta" + SUF),
            "save_counter": ("save_counter", "save_counter" + SUF)}
    if with_graph:
        tensors[T.OBJECT_GRAPH_KEY] = W.object_graph(tree)
    return tensors


@pytest.fixture(scope="module")
def params():
    return synthetic.make_encoder_params(seed=1), synthetic.make_regressor_params(seed=2)


def _assert_weights(w, enc, reg):
    for s in CONV_SPECS:
        for var in ("kernel", "bias"):
            np.testing.assert_array_equal(w["%s/%s" % (s.name, var)], enc["%s/%s" % (s.name, var)])
        for var in T._BN_VARS:
            np.testing.assert_array_equal(w["%s/%s" % (s.bn_name, var)], enc["%s/%s" % (s.bn_name, var)])
    for i in range(3):
        for var in ("kernel", "bias"):
            np.testing.assert_array_equal(w["dense_%d/%s" % (i, var)], reg["dense_%d/%s" % (i, var)])


def test_roundtrip_by_object_graph(tmp_path, params):
    enc, reg = params
    order = T.keras_weighted_layer_order(False)
    assert len(order) == 106 and order[:2] == ["conv1", "bn_conv1"]
    # summary() order of keras_applications 1.0.8: ... res2a_branch2c, res2a_branch1, bn2a_branch2c, bn2a_branch1
    assert order[6:10] == ["res2a_branch2c", "res2a_branch1", "bn2a_branch2c", "bn2a_branch1"]
    # scramble positions: the graph's Keras names, not the index, must decide
    scr = list(reversed(order))
    W.write_bundle(str(tmp_path / "ckpt-3"), _hmr_tensors(enc, reg, scr))
    W.write_checkpoint_state(str(tmp_path), "ckpt-3")
    assert T.latest_checkpoint(str(tmp_path)) == str(tmp_path / "ckpt-3")
    w, info = T.load_hmr_weights(str(tmp_path))
    assert "object graph" in info["resolved_by"]
    _assert_weights(w, enc, reg)
    assert w["inital_theta"].shape == (1, 85)


@pytest.mark.parametrize("shortcut_first", [False, True])
def test_roundtrip_by_position(tmp_path, params, shortcut_first):
    enc, reg = params
    order = T.keras_weighted_layer_order(shortcut_first)
    W.write_bundle(str(tmp_path / "ckpt-1"), _hmr_tensors(enc, reg, order, with_graph=False))
    w, info = T.load_hmr_weights(str(tmp_path / "ckpt-1"))
    assert "position" in info["resolved_by"] and ("2.2" in info["resolved_by"]) == shortcut_first
    _assert_weights(w, enc, reg)


def test_bundle_reader_details(tmp_path):
    rng = np.random.default_rng(5)
    tensors = {"a/b%s" % SUF: rng.normal(size=(3, 4)).astype(np.float32), "a/c": np.arange(7, dtype=np.int64),
               "s": b"hello world", "z/scalar": np.array(2.5, dtype=np.float64), "b": np.array([True, False, True])}
    for i in range(200):  # many keys with shared prefixes -> several data blocks and restart points
        tensors["layer/with/a/long/common/prefix/%03d" % i] = np.full((i % 3 + 1,), i, dtype=np.int32)
    W.write_bundle(str(tmp_path / "x"), tensors)
    rd = T.BundleReader(str(tmp_path / "x"), verify=True)
    assert rd.keys() == sorted(tensors)
    for k, v in tensors.items():
        got = rd.get(k)
        if isinstance(v, bytes):
            assert got == v
        else:
            assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    with pytest.raises(KeyError):
        rd.get("nope")


def test_corruption_is_detected(tmp_path):
    W.write_bundle(str(tmp_path / "x"), {"w": np.arange(100, dtype=np.float32)})
    raw = bytearray(open(tmp_path / "x.data-00000-of-00001", "rb").read())
    raw[17] ^= 0x40
    open(tmp_path / "x.data-00000-of-00001", "wb").write(raw)
    with pytest.raises(T.CheckpointError, match="checksum"):
        T.BundleReader(str(tmp_path / "x")).get("w")
    idx = bytearray(open(tmp_path / "x.index", "rb").read())
    idx[5] ^= 0x01
    open(tmp_path / "x.index", "wb").write(idx)
    with pytest.raises(T.CheckpointError):
        T.BundleReader(str(tmp_path / "x"))
    open(tmp_path / "y.index", "wb").write(b"\0" * 100)
    with pytest.raises(T.CheckpointError, match="magic"):
        T.BundleReader(str(tmp_path / "y"))
    assert T.latest_checkpoint(str(tmp_path)) is None
    with pytest.raises(FileNotFoundError):
        T.load_hmr_weights(str(tmp_path))


def test_wrong_architecture_is_rejected(tmp_path, params):
    enc, reg = params
    enc = dict(enc)
    enc["res4c_branch2b/kernel"] = np.zeros((3, 3, 256, 128), np.float32)
    W.write_bundle(str(tmp_path / "ckpt-1"), _hmr_tensors(enc, reg, T.keras_weighted_layer_order(False)))
    with pytest.raises(T.CheckpointError, match="res4c_branch2b"):
        T.load_hmr_weights(str(tmp_path / "ckpt-1"))


def test_byte_flips_in_the_index_never_escape_as_other_exceptions(tmp_path):
    rng = np.random.default_rng(5)
    tensors = {"a/b" + SUF: rng.normal(size=(3, 4)).astype(np.float32), "s": b"hello",
               T.OBJECT_GRAPH_KEY: W.object_graph({"x": {"y": ("n", "a/b" + SUF)}})}
    for i in range(60):
        tensors["k/%03d" % i] = np.full((i % 3 + 1,), i, np.int32)
    W.write_bundle(str(tmp_path / "x"), tensors)
    idx = open(tmp_path / "x.index", "rb").read()
    os.link(tmp_path / "x.data-00000-of-00001", tmp_path / "y.data-00000-of-00001")
    for it in range(600):
        b = bytearray(idx)
        for _k in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        open(tmp_path / "y.index", "wb").write(b)
        try:
            rd = T.BundleReader(str(tmp_path / "y"), verify=bool(it & 1))  # without CRC checks damage reaches the parsers
            for k in rd.keys():
                rd.get(k)
            if T.OBJECT_GRAPH_KEY in rd:
                T.ObjectGraph(rd.get(T.OBJECT_GRAPH_KEY))
        except (T.CheckpointError, FileNotFoundError):
            pass
