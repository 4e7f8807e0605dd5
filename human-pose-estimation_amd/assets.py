"""Real-asset ingestion for the hot path: the user's own licensed SMPL ``model.pkl`` and
``neutral_smpl_mean_params.h5`` -> the host arrays ``hpe_load_smpl`` / ``hpe_load_mean_theta`` take.

Reference behaviour restated:
  * src/tf_smpl/batch_smpl.py:31-81 -- ``pickle.load(f, encoding='latin1')`` of a dict with ``v_template`` [6890,3],
    ``shapedirs`` [6890,3,10], ``posedirs`` [6890,3,207], scipy-sparse ``J_regressor`` [24,6890] and
    ``cocoplus_regressor`` [19,6890], ``weights`` [6890,24], ``kintree_table`` [2,24] (row 0 = parents).
  * src/predictor.py:93-105 -- ``deepdish.io.load`` of the mean-parameter file: ``pose`` [72], ``shape`` [10].

A pickle executes whatever its stream names, so the SMPL file is read with an allow-list unpickler: NumPy array /
dtype / scalar reconstruction, scipy.sparse matrix classes and plain containers -- nothing else resolves.  The
original SMPL release stores several fields as ``chumpy`` objects; chumpy is not needed (and not installable) here:
those classes are mapped to an inert holder whose ``x`` attribute is the array.
"""
from __future__ import annotations

import io
import os
import pickle
from os.path import dirname, join

import numpy as np

SMPL_KEYS = ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "kintree_table")


class AssetError(ValueError):
    pass


class _Inert:
    """Stand-in for chumpy.* classes: keeps the pickled state, runs nothing."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):
            self.__dict__.update(state[1])
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
        else:
            self.__dict__["_state"] = state


_ALLOWED = {
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "complex"), ("builtins", "bool"), ("builtins", "str"),
    ("builtins", "bytes"), ("builtins", "bytearray"), ("builtins", "slice"), ("builtins", "object"),
    ("__builtin__", "dict"), ("__builtin__", "list"), ("__builtin__", "tuple"), ("__builtin__", "set"),
    ("__builtin__", "int"), ("__builtin__", "long"), ("__builtin__", "float"), ("__builtin__", "bool"),
    ("__builtin__", "str"), ("__builtin__", "unicode"), ("__builtin__", "object"),
    ("collections", "OrderedDict"),
    ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"),
    ("_codecs", "encode"),
}
_NUMPY_NAMES = {"_reconstruct", "ndarray", "dtype", "scalar", "_frombuffer"}
_NUMPY_MODULES = {"numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric"}
_SPARSE_CLASSES = {"csc_matrix", "csr_matrix", "coo_matrix", "csc_array", "csr_array", "coo_array"}


class _AllowListUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            if module == "__builtin__":
                import builtins

                return {"long": int, "unicode": str}.get(name) or getattr(builtins, name)
            if module == "copy_reg":
                module = "copyreg"
            return super().find_class(module, name)
        root = module.split(".")[0]
        if root == "numpy" and name in _NUMPY_NAMES and module in _NUMPY_MODULES:
            if name in ("ndarray", "dtype"):
                return getattr(np, name)
            try:
                import numpy._core.multiarray as ma
                import numpy._core.numeric as nu
            except ImportError:  # NumPy < 2
                import numpy.core.multiarray as ma
                import numpy.core.numeric as nu
            return getattr(nu if name == "_frombuffer" else ma, name)
        if root == "scipy" and module.startswith("scipy.sparse") and name in _SPARSE_CLASSES:
            import scipy.sparse as sp

            return getattr(sp, name)
        if root == "chumpy":
            return _Inert
        raise pickle.UnpicklingError("refusing to resolve %s.%s while reading an SMPL model file" % (module, name))


def _dense(v, key):
    if isinstance(v, _Inert):
        for attr in ("x", "_x"):
            if attr in v.__dict__:
                v = v.__dict__[attr]
                break
        else:
            raise AssetError("%s is a chumpy expression without a stored value; export it as a plain array" % key)
    if hasattr(v, "todense"):  # scipy sparse (batch_smpl.py:50-54, :75-79 call .T.todense())
        v = np.asarray(v.todense())
    return np.asarray(v)


def load_smpl_model(path, joint_type="cocoplus"):
    """-> dict of float32/int32 arrays in the layouts ``HpeSmplModel`` (include/hpe.h) documents.  ``.npz`` files carry
    the same keys as dense arrays."""
    if joint_type != "cocoplus":  # the reference drops into a debugger for anything else (batch_smpl.py:83-86)
        raise AssetError("joint_type %r: only 'cocoplus' (19 keypoints) is on this path" % (joint_type,))
    if str(path).endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            dd = {k: z[k] for k in z.files}
    else:
        with open(path, "rb") as f:
            raw = f.read()
        try:
            dd = _AllowListUnpickler(io.BytesIO(raw), encoding="latin1").load()
        except pickle.UnpicklingError:
            raise
        except Exception as e:
            raise AssetError("%s is not a readable SMPL pickle: %s: %s" % (path, type(e).__name__, e)) from e
    if not isinstance(dd, dict):
        raise AssetError("%s: expected a dict of SMPL fields, got %s" % (path, type(dd).__name__))
    kp_key = "cocoplus_regressor"
    missing = [k for k in SMPL_KEYS + (kp_key,) if k not in dd and not (k == "kintree_table" and "parents" in dd)]
    if missing:
        raise AssetError("%s lacks SMPL field(s) %s" % (path, missing))
    out = {}
    out["v_template"] = _dense(dd["v_template"], "v_template").astype(np.float32)
    V = out["v_template"].shape[0]
    out["shapedirs"] = _dense(dd["shapedirs"], "shapedirs").astype(np.float32)[..., :10]  # 300-beta models: first 10
    out["posedirs"] = _dense(dd["posedirs"], "posedirs").astype(np.float32)
    out["J_regressor"] = _dense(dd["J_regressor"], "J_regressor").astype(np.float32)
    out["weights"] = _dense(dd["weights"], "weights").astype(np.float32)
    out[kp_key] = _dense(dd[kp_key], kp_key).astype(np.float32)
    if "kintree_table" in dd:
        out["kintree_table"] = _dense(dd["kintree_table"], "kintree_table").astype(np.int64)
        parents = out["kintree_table"][0].astype(np.int32)  # batch_smpl.py:65 (entry 0 wraps to -1 / 2^32-1: unused)
    else:
        parents = _dense(dd["parents"], "parents").astype(np.int32)
    parents = parents.copy()
    parents[0] = -1
    out["parents"] = parents
    out["kintree_table"] = np.stack([parents.astype(np.int64), np.arange(24, dtype=np.int64)])
    want = {"v_template": (V, 3), "shapedirs": (V, 3, 10), "posedirs": (V, 3, 207), "J_regressor": (24, V),
            "weights": (V, 24), kp_key: (19, V), "parents": (24,)}
    for k, shp in want.items():
        if out[k].shape != shp:
            raise AssetError("%s: %s has shape %s, expected %s" % (path, k, out[k].shape, shp))
    if V != 6890:
        raise AssetError("%s: %d vertices (the path is built for SMPL's 6890)" % (path, V))
    if not ((parents[1:] >= 0) & (parents[1:] < np.arange(1, 24))).all():
        raise AssetError("%s: kintree_table parents must precede their children" % path)
    return out


def load_mean_params(smpl_model_path):
    """``neutral_smpl_mean_params.{npz,h5}`` next to the SMPL model (src/predictor.py:93-95) -> {'pose','shape'}."""
    base = join(dirname(smpl_model_path), "neutral_smpl_mean_params")
    if os.path.exists(base + ".npz"):
        with np.load(base + ".npz", allow_pickle=False) as z:
            mv = {"pose": z["pose"], "shape": z["shape"]}
    elif os.path.exists(base + ".h5"):
        # PyTables-written (deepdish) root group with the arrays 'pose' and 'shape'; read by the package's own reader
        from . import hdf5_lite

        root = hdf5_lite.File(base + ".h5").root
        missing = [k for k in ("pose", "shape") if k not in root]
        if missing:
            raise AssetError("%s.h5 has no dataset %s (found: %s)" % (base, missing, sorted(root)))
        mv = {k: np.asarray(root[k].read()) for k in ("pose", "shape")}
    else:
        raise FileNotFoundError(base + ".{npz,h5}")
    pose, shape = np.asarray(mv["pose"], np.float64).reshape(-1), np.asarray(mv["shape"], np.float64).reshape(-1)
    if pose.shape != (72,) or shape.shape != (10,):
        raise AssetError("%s: pose %s / shape %s, expected (72,) / (10,)" % (base, pose.shape, shape.shape))
    return {"pose": pose, "shape": shape}
