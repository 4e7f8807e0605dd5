"""Error behaviour and edge cases of the C-ABI boundary on the GPU (the reference's own error convention is
'exceptions propagate' / ipdb; here every bad call must fail loudly and leave the ctx usable)."""
import ctypes as C

import numpy as np
import pytest

import hpe_amd
from hpe_amd import _lib, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def smpl_engine():
    e = hpe_amd.HpeEngine(device=0, max_batch=8)
    e.load_smpl(synthetic.make_smpl_model())
    e.finalize()
    yield e
    e.close()


def test_partial_context_rejects_missing_parts(smpl_engine):
    import torch

    img = torch.zeros((1, 224, 224, 3), device="cuda")
    with pytest.raises(hpe_amd.HpeError, match="encoder weights"):
        smpl_engine.encoder(img)
    with pytest.raises(hpe_amd.HpeError, match="regressor"):
        smpl_engine.regress_stage(torch.zeros((1, 2048), device="cuda"))
    # still usable afterwards
    out = smpl_engine.smpl(torch.zeros((2, 85), device="cuda"), want=("joints",))
    assert tuple(out["joints"].shape) == (2, 19, 3)


def test_batch_bounds_and_types(smpl_engine):
    import torch

    with pytest.raises(hpe_amd.HpeError, match="outside"):
        smpl_engine.smpl(torch.zeros((9, 85), device="cuda"))  # > max_batch
    with pytest.raises(ValueError):
        smpl_engine.smpl(torch.zeros((2, 84), device="cuda"))
    with pytest.raises(TypeError):
        smpl_engine.smpl(torch.zeros((2, 85), device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError, match="GPU"):
        smpl_engine.smpl(torch.zeros((2, 85)))


def test_call_order_is_enforced():
    e = hpe_amd.HpeEngine(device=0, max_batch=2)
    with pytest.raises(hpe_amd.HpeError, match="nothing was loaded"):
        e.finalize()
    e.load_smpl(synthetic.make_smpl_model())
    e.finalize()
    with pytest.raises(hpe_amd.HpeError, match="already finalized"):
        e.load_smpl(synthetic.make_smpl_model())
    e.close()
    e.close()  # idempotent


def test_bad_smpl_model_is_rejected():
    m = dict(synthetic.make_smpl_model())
    k = m["kintree_table"].copy()
    k[0, 5] = 9  # parent after child: the reference's FK loop would index a result that does not exist yet
    m["kintree_table"] = k
    e = hpe_amd.HpeEngine(device=0, max_batch=2)
    with pytest.raises(hpe_amd.HpeError, match="parents"):
        e.load_smpl(m)
    with pytest.raises(ValueError, match="Unknown joint type"):
        e.load_smpl(synthetic.make_smpl_model(), joint_type="coco")
    e.close()


def test_lsp_joint_type(smpl_engine):
    import torch

    from oracle import hmr_oracle as O

    m = synthetic.make_smpl_model()
    e = hpe_amd.HpeEngine(device=0, max_batch=4)
    e.load_smpl(m, joint_type="lsp")
    e.finalize()
    th = synthetic.make_thetas(3, seed=4)
    out = e.smpl(torch.from_numpy(th).cuda(), want=("joints", "kp2d"))
    ref = O.SMPL(m, joint_type="lsp")(th[:, 75:], th[:, 3:75])
    assert tuple(out["joints"].shape) == (3, 14, 3) and tuple(out["kp2d"].shape) == (3, 14, 2)
    assert np.abs(out["joints"].cpu().numpy() - ref).max() < 1e-5
    e.close()


def test_extreme_poses_stay_finite_and_match(smpl_engine):
    """zero pose (angle = sqrt(3)*1e-8 quirk), huge angles, tiny angles"""
    import torch

    from oracle import hmr_oracle as O

    m = synthetic.make_smpl_model()
    th = np.zeros((4, 85), np.float32)
    th[:, 0] = 1.0
    th[1, 3:75] = 25.0  # many turns
    th[2, 3:75] = 1e-6
    th[3, 3:75] = -np.pi
    th[:, 75:] = np.array([[0] * 10, [5] * 10, [-5] * 10, [3, -3] * 5], np.float32)
    out = smpl_engine.smpl(torch.from_numpy(th).cuda(), want=("verts", "joints", "Rs"))
    v, j, Rs = O.SMPL(m, dtype=np.float64)(th[:, 75:].astype(np.float64), th[:, 3:75].astype(np.float64), get_skin=True)
    assert np.isfinite(out["verts"].cpu().numpy()).all()
    assert np.abs(out["Rs"].cpu().numpy() - Rs).max() < 2e-5  # sin/cos of 43 rad in fp32
    assert np.abs(out["verts"].cpu().numpy() - v).max() / np.abs(v).max() < 1e-4
