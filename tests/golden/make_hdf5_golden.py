#!/opt/conda/bin/python3.9
"""Generates tests/golden/hdf5/*.h5 + expected.npz with the real HDF5 library (h5py 3.3 / libhdf5 1.10 that happens to
sit in this container's /opt/conda; it is NOT importable from the interpreter the package runs under, and it does not
exist on the GPU box).  Run once, here:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py

Files written (all data is synthetic, seeded):
  mean_params.h5   the layout deepdish/PyTables gives neutral_smpl_mean_params.h5 (src/predictor.py:93-105): root group
                   with two small contiguous float arrays 'pose'[72] and 'shape'[10], libver earliest
  layouts.h5       nested groups; contiguous / compact-ish small / chunked+gzip+shuffle+fletcher32 / big-endian / ints /
                   scalar / fixed strings / never-written dataset
  latest.h5        libver='latest' would use layout v4 + dense groups for >8 links: here only a compact group written
                   with libver=('earliest','v108') and track_order to exercise v2 object headers + link messages
  expected.npz     every dataset read back through h5py, keyed "<file>:<path>"
Additionally expected_pytables.npz holds the h5py reading of a few PyTables-written files from
/opt/conda/lib/python3.9/site-packages/tables/tests (the test reads those files in place when they exist).
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(20260401)
expected = {}


def record(fname):
    with h5py.File(os.path.join(HERE, fname), "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                a = obj[()]
                if isinstance(a, bytes):
                    a = np.array(a)
                expected["%s:/%s" % (fname, name)] = np.asarray(a)
        f.visititems(visit)


with h5py.File(os.path.join(HERE, "mean_params.h5"), "w", libver="earliest") as f:
    pose = rng.normal(0, 0.2, 72)
    pose[:3] = [3.1, 0.05, -0.02]
    f.create_dataset("pose", data=pose)
    f.create_dataset("shape", data=rng.normal(0, 0.5, 10))
record("mean_params.h5")

with h5py.File(os.path.join(HERE, "layouts.h5"), "w", libver="earliest") as f:
    g = f.create_group("a/b")
    g.create_dataset("f32", data=rng.normal(size=(5, 7)).astype("<f4"))
    g.create_dataset("f64_be", data=rng.normal(size=(3, 4)).astype(">f8"))
    f.create_dataset("i16", data=rng.integers(-3000, 3000, (11,)).astype("<i2"))
    f.create_dataset("u8", data=rng.integers(0, 255, (4, 3, 2)).astype("u1"))
    f.create_dataset("i64_be", data=rng.integers(-10**12, 10**12, (6,)).astype(">i8"))
    f.create_dataset("scalar", data=np.float32(2.5))
    f.create_dataset("chunked", data=rng.normal(size=(37, 29)).astype("<f4"), chunks=(8, 16), compression="gzip",
                     shuffle=True, fletcher32=True)
    f.create_dataset("chunked_plain", data=rng.integers(0, 9, (10, 10, 3)).astype("<i4"), chunks=(4, 4, 2))
    f.create_dataset("sparse_chunks", shape=(20, 20), dtype="<f4", chunks=(5, 5))
    f["sparse_chunks"][12:14, 3:9] = 7.0
    f.create_dataset("never_written", shape=(4,), dtype="<f8")
    f.create_dataset("strs", data=np.array([b"abc", b"de", b"fghij"], dtype="S5"))
    many = f.create_group("many")  # > 1 SNOD / B-tree levels
    for i in range(80):
        many.create_dataset("d%03d" % i, data=np.arange(i % 5 + 1, dtype="<i4") + i)
record("layouts.h5")

with h5py.File(os.path.join(HERE, "v2hdr.h5"), "w", libver=("v108", "v108"), track_order=True) as f:
    f.create_dataset("x", data=rng.normal(size=(6,)))
    f.create_group("g").create_dataset("y", data=rng.integers(0, 100, (2, 2)).astype("<i4"))
record("v2hdr.h5")

np.savez_compressed(os.path.join(HERE, "expected.npz"), **expected)
print(len(expected), "datasets")

import glob

pt = "/opt/conda/lib/python3.9/site-packages/tables"
pexp = {}
for p in sorted(glob.glob(os.path.join(pt, "**", "*.h5"), recursive=True)):
    fname = os.path.relpath(p, pt)
    try:
        with h5py.File(p, "r") as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    try:
                        dt = obj.dtype
                        if dt.kind in "fiu" and dt.itemsize <= 8 and obj.size <= 20000:
                            pexp["%s:/%s" % (fname, name)] = obj[()]
                    except Exception as e:  # types / filters h5py itself cannot represent here
                        print("  skip", fname, name, str(e)[:60])
            f.visititems(visit)
    except Exception as e:
        print("skip", fname, e)
np.savez_compressed(os.path.join(HERE, "expected_pytables.npz"), **pexp)
print(len(pexp), "pytables datasets in", len({k.split(":")[0] for k in pexp}), "files")
