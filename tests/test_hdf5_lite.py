"""hdf5_lite against files written by the real HDF5 library.

Fixtures under tests/golden/hdf5/ were written with h5py 3.3 / libhdf5 1.10 by tests/golden/make_hdf5_golden.py and read
back through h5py into expected.npz; PyTables-written files (what deepdish produces, src/predictor.py:93-105) are
checked in place when this container's /opt/conda copy of the PyTables test-suite data is present.
"""
import os

import numpy as np
import pytest

from hpe_amd import hdf5_lite as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
PYTABLES = "/opt/conda/lib/python3.9/site-packages/tables"


def _walk(node, prefix=""):
    for k, v in node.items():
        if isinstance(v, dict):
            yield from _walk(v, prefix + "/" + k)
        else:
            yield prefix + "/" + k, v


@pytest.mark.parametrize("fname", ["mean_params.h5", "layouts.h5", "v2hdr.h5"])
def test_against_h5py_reading(fname):
    exp = np.load(os.path.join(GOLD, "expected.npz"), allow_pickle=False)
    want = {k.split(":", 1)[1]: exp[k] for k in exp.files if k.startswith(fname + ":")}
    got = dict(_walk(H.load(os.path.join(GOLD, fname))))
    assert sorted(got) == sorted(want)
    for k, w in want.items():
        g = got[k]
        assert g.shape == w.shape, k
        if w.dtype.kind == "S":
            assert [x.rstrip(b"\0") for x in g.ravel().tolist()] == [x.rstrip(b"\0") for x in w.ravel().tolist()], k
        else:
            assert g.dtype.kind == w.dtype.kind and g.dtype.itemsize == w.dtype.itemsize, k
            assert np.array_equal(g, w), k  # bit exact


def test_mean_params_layout_feeds_predictor_loader(tmp_path):
    """neutral_smpl_mean_params.h5 next to the SMPL model is read without h5py (predictor._load_mean_file)."""
    import shutil

    from hpe_amd import predictor as P

    shutil.copy(os.path.join(GOLD, "mean_params.h5"), tmp_path / "neutral_smpl_mean_params.h5")
    mv = P._load_mean_file(str(tmp_path / "model.pkl"))
    exp = np.load(os.path.join(GOLD, "expected.npz"), allow_pickle=False)
    assert np.array_equal(mv["pose"], exp["mean_params.h5:/pose"]) and mv["pose"].shape == (72,)
    assert np.array_equal(mv["shape"], exp["mean_params.h5:/shape"]) and mv["shape"].shape == (10,)


@pytest.mark.skipif(not os.path.isdir(PYTABLES), reason="PyTables test data not present on this machine")
def test_pytables_written_files():
    exp = np.load(os.path.join(GOLD, "expected_pytables.npz"), allow_pickle=False)
    by_file = {}
    for k in exp.files:
        f, path = k.split(":", 1)
        by_file.setdefault(f, {})[path] = exp[k]
    assert by_file
    n_checked = 0
    for fname, want in by_file.items():
        f = H.File(os.path.join(PYTABLES, fname))
        got = {p: d for p, d in _walk(f.root)}
        if fname == "tests/float.h5":
            # it also holds 96/128-bit floats: those datasets are rejected on read, the rest of the file still reads
            assert isinstance(got["/longdouble"], H.Unsupported)
            with pytest.raises(H.Hdf5Error):
                got["/longdouble"].read()
            with pytest.raises(H.Hdf5Error):
                H.load(os.path.join(PYTABLES, fname))
        for path, w in want.items():
            g = got[path].read()
            assert g.shape == w.shape and np.array_equal(g.astype(w.dtype), w), (fname, path)
            n_checked += 1
    assert n_checked >= 90  # contiguous + chunked(+zlib/shuffle) arrays, old layout versions, both byte orders


def test_rejects_garbage_and_truncation():
    with pytest.raises(H.Hdf5Error):
        H.File(b"not an hdf5 file at all" * 40)
    raw = open(os.path.join(GOLD, "mean_params.h5"), "rb").read()
    with pytest.raises(H.Hdf5Error):
        H.load(raw[:1500])  # object headers / data cut off


def test_user_block_offset():
    raw = open(os.path.join(GOLD, "mean_params.h5"), "rb").read()
    shifted = b"\0" * 512 + raw  # superblock search at 512; addresses relative to it
    a, b = H.load(raw), H.load(shifted)
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["shape"], b["shape"])


def test_byte_flips_never_escape_as_other_exceptions():
    """Robustness: random damage either still parses or raises Hdf5Error -- nothing else (bad UTF-8 names, corrupt deflate
    streams, absurd dimensions, out-of-range addresses)."""
    rng = np.random.default_rng(7)
    for fname, rounds in (("mean_params.h5", 1500), ("layouts.h5", 600)):
        raw = open(os.path.join(GOLD, fname), "rb").read()
        for _ in range(rounds):
            b = bytearray(raw)
            for _k in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            try:
                H.load(bytes(b))
            except H.Hdf5Error:
                pass
