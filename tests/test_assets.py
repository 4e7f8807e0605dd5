"""Real-asset ingestion: SMPL model.pkl through the allow-list unpickler, mean params from .h5/.npz.
The SMPL model itself is licensed and absent: the files are SMPL-shaped synthetic models pickled the ways the real ones
are (protocol 2 "python-2 style", scipy-sparse regressors, chumpy-wrapped arrays)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp

from hpe_amd import assets, synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")


@pytest.fixture(scope="module")
def model():
    return synthetic.make_smpl_model(seed=3)


def _as_pickle_dict(m):
    d = {k: np.asarray(v) for k, v in m.items()}
    d["J_regressor"] = sp.csc_matrix(d["J_regressor"])
    d["cocoplus_regressor"] = sp.csc_matrix(d["cocoplus_regressor"])
    return d


def _check(out, m):
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "cocoplus_regressor"):
        assert out[k].dtype == np.float32
        np.testing.assert_array_equal(out[k], np.asarray(m[k], np.float32))
    np.testing.assert_array_equal(out["parents"][1:], np.asarray(m["kintree_table"])[0][1:].astype(np.int32))
    assert out["parents"][0] == -1 and out["kintree_table"].shape == (2, 24)


@pytest.mark.parametrize("protocol", [2, 4])
def test_pickle_with_sparse_regressors(tmp_path, model, protocol):
    p = tmp_path / "model.pkl"
    with open(p, "wb") as f:
        pickle.dump(_as_pickle_dict(model), f, protocol=protocol)
    _check(assets.load_smpl_model(str(p)), model)


def test_npz_model(tmp_path, model):
    p = tmp_path / "model.npz"
    np.savez(p, **{k: np.asarray(v) for k, v in model.items()})
    _check(assets.load_smpl_model(str(p)), model)


def test_chumpy_wrapped_fields_need_no_chumpy(tmp_path, model):
    """The original SMPL release wraps arrays in chumpy.ch.Ch (state dict with the array under 'x')."""
    mod_root, mod = types.ModuleType("chumpy"), types.ModuleType("chumpy.ch")

    class Ch(object):
        def __init__(self, x):
            self.x = x
            self._dirty = True

    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    mod.Ch = Ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod_root, mod
    try:
        d = _as_pickle_dict(model)
        for k in ("v_template", "shapedirs", "posedirs", "weights"):
            d[k] = Ch(d[k])
        blob = pickle.dumps(d, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    assert b"chumpy" in blob
    p = tmp_path / "model.pkl"
    p.write_bytes(blob)
    _check(assets.load_smpl_model(str(p)), model)


def test_hostile_pickle_is_refused(tmp_path):
    class Boom(object):
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    p = tmp_path / "model.pkl"
    p.write_bytes(pickle.dumps({"v_template": Boom()}))
    with pytest.raises(pickle.UnpicklingError):
        assets.load_smpl_model(str(p))
    assert not (tmp_path / "pwned").exists()


def test_shape_and_key_validation(tmp_path, model):
    d = _as_pickle_dict(model)
    del d["posedirs"]
    p = tmp_path / "a.pkl"
    p.write_bytes(pickle.dumps(d))
    with pytest.raises(assets.AssetError, match="posedirs"):
        assets.load_smpl_model(str(p))
    d = _as_pickle_dict(model)
    d["weights"] = d["weights"][:, :23]
    p.write_bytes(pickle.dumps(d))
    with pytest.raises(assets.AssetError, match="weights"):
        assets.load_smpl_model(str(p))
    d = _as_pickle_dict(model)
    kt = np.asarray(d["kintree_table"]).copy()
    kt[0, 5] = 9  # parent after child
    d["kintree_table"] = kt
    p.write_bytes(pickle.dumps(d))
    with pytest.raises(assets.AssetError, match="parents"):
        assets.load_smpl_model(str(p))
    with pytest.raises(assets.AssetError, match="cocoplus"):
        assets.load_smpl_model(str(p), joint_type="lsp17")


def test_mean_params_h5_and_npz(tmp_path):
    import shutil

    exp = np.load(os.path.join(GOLD, "expected.npz"), allow_pickle=False)
    with pytest.raises(FileNotFoundError):
        assets.load_mean_params(str(tmp_path / "model.pkl"))
    shutil.copy(os.path.join(GOLD, "mean_params.h5"), tmp_path / "neutral_smpl_mean_params.h5")
    mv = assets.load_mean_params(str(tmp_path / "model.pkl"))
    np.testing.assert_array_equal(mv["pose"], exp["mean_params.h5:/pose"])
    np.testing.assert_array_equal(mv["shape"], exp["mean_params.h5:/shape"])
    np.savez(tmp_path / "neutral_smpl_mean_params.npz", pose=np.zeros(72), shape=np.ones(10))  # .npz wins
    mv = assets.load_mean_params(str(tmp_path / "model.pkl"))
    assert mv["shape"].sum() == 10
    np.savez(tmp_path / "neutral_smpl_mean_params.npz", pose=np.zeros(71), shape=np.ones(10))
    with pytest.raises(assets.AssetError):
        assets.load_mean_params(str(tmp_path / "model.pkl"))
