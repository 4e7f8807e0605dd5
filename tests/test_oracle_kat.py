"""Closed-form known-answer tests that pin the CPU oracle.  The reference ships no
tests/fixtures (F2), so these + the fp64 self-consistency run are what the oracle is pinned by
("parity unpinned" w.r.t. the reference's own outputs)."""
import math

import numpy as np

import hpe_amd
from hpe_amd import resnet_spec, synthetic
from oracle import hmr_oracle as O


def test_rodrigues_zero_is_identity():
    R = O.batch_rodrigues(np.zeros((3, 3), np.float32))
    np.testing.assert_allclose(R, np.tile(np.eye(3, dtype=np.float32), (3, 1, 1)), atol=1e-7)


def test_rodrigues_pi_about_x():
    R = O.batch_rodrigues(np.array([[math.pi, 0, 0]], np.float32))
    np.testing.assert_allclose(R[0], np.diag([1.0, -1.0, -1.0]), atol=2e-6)


def test_rodrigues_is_rotation():
    th = synthetic.make_thetas(16)[:, 3:75].reshape(-1, 3).astype(np.float64)
    R = O.batch_rodrigues(th)
    np.testing.assert_allclose(np.matmul(R, R.transpose(0, 2, 1)), np.tile(np.eye(3), (len(R), 1, 1)), atol=2e-7)  # r = theta/||theta+1e-8|| is unit only to ~1e-8 (the quirk)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=2e-7)


def test_skew_layout():
    S = O.batch_skew(np.array([[1.0, 2.0, 3.0]], np.float32))[0]
    np.testing.assert_array_equal(S, np.array([[0, -3, 2], [3, 0, -1], [-2, 1, 0]], np.float32))


def test_smpl_rest_pose(smpl_model):
    sm = O.SMPL(smpl_model, dtype=np.float64)
    verts, joints, Rs = sm(np.zeros((2, 10)), np.zeros((2, 72)), get_skin=True)
    np.testing.assert_allclose(verts[0], smpl_model["v_template"].astype(np.float64), atol=1e-6)
    J0 = smpl_model["J_regressor"].astype(np.float64) @ smpl_model["v_template"].astype(np.float64)
    np.testing.assert_allclose(sm.J_transformed[1], J0, atol=1e-6)
    np.testing.assert_allclose(joints[0], smpl_model["cocoplus_regressor"].astype(np.float64) @ verts[0], atol=1e-9)


def test_smpl_global_rotation_only(smpl_model):
    sm = O.SMPL(smpl_model, dtype=np.float64)
    theta = np.zeros((1, 72))
    theta[0, :3] = [0.3, -0.7, 0.2]
    verts, _, Rs = sm(np.zeros((1, 10)), theta, get_skin=True)
    R0 = Rs[0, 0]
    vt = smpl_model["v_template"].astype(np.float64)
    J0 = smpl_model["J_regressor"].astype(np.float64)[0] @ vt
    # posedirs see (R_j - I) of joints 1..23 only -> zero here; every joint gets the root transform
    np.testing.assert_allclose(verts[0], (vt - J0) @ R0.T + J0, atol=2e-6)


def test_lbs_partition_of_unity(smpl_model):
    assert np.allclose(smpl_model["weights"].sum(1), 1.0, atol=1e-6)
    assert np.allclose(smpl_model["J_regressor"].sum(1), 1.0, atol=1e-5)
    assert (smpl_model["weights"] >= 0).all()
    par = smpl_model["kintree_table"][0].astype(np.int32)
    assert par[0] == -1 and (par[1:] < np.arange(1, 24)).all()


def test_projection_closed_form():
    X = np.arange(2 * 5 * 3, dtype=np.float32).reshape(2, 5, 3) * 0.1
    cam = np.array([[0.9, 0.1, -0.2], [1.1, 0.0, 0.3]], np.float32)
    out = O.batch_orth_proj_idrot(X, cam)
    np.testing.assert_allclose(out, cam[:, None, :1] * (X[:, :, :2] + cam[:, None, 1:]), rtol=1e-6)
    px = O.reproject_vertices(X, np.array([[1.0, 0, 0], [1.0, 0, 0]], np.float32), np.array([224.0, 224.0], np.float32))
    Xe = np.zeros((2, 2, 3), np.float32)
    Xe[:, 0, :2] = -1
    Xe[:, 1, :2] = 1
    pe = O.reproject_vertices(Xe, np.array([[1.0, 0, 0], [1.0, 0, 0]], np.float32), np.array([224.0, 224.0], np.float32))
    np.testing.assert_allclose(pe[:, 0], 0.0)
    np.testing.assert_allclose(pe[:, 1], 224.0)
    assert px.shape == (2, 5, 2)


def test_kp_loss_closed_form():
    gt = np.zeros((1, 19, 3), np.float32)
    pred = np.ones((1, 19, 2), np.float32)
    assert O.kp_reprojection_loss(gt, pred) == 0.0  # nothing visible
    gt[0, 4] = [0.5, -0.25, 1.0]
    pred[0, 4] = [0.5 + 0.3, -0.25 - 0.1]
    np.testing.assert_allclose(O.kp_reprojection_loss(gt, pred), (0.3 + 0.1) / 2, rtol=1e-6)


def test_bidirectional_dist_closed_form():
    A = np.array([[1.0, 2.0], [3.0, 4.0], [0.0, 9.0]], np.float32)
    assert O.bidirectional_dist(A, A.copy()) == 0.0
    a = np.array([[1.0, 1.0]], np.float32)
    b = np.array([[4.0, 5.0]], np.float32)
    np.testing.assert_allclose(O.bidirectional_dist(a, b), 5.0 + 7.0)
    # denominator quirk: 3 + 6890
    sil_gt = np.array([[0, 1.0, 1.0]], np.float32)  # (b, y, x)
    pred = np.tile(b[None], (1, 6890, 1)).astype(np.float32)
    np.testing.assert_allclose(O.mesh_reprojection_loss(sil_gt, pred, 1), (6890 * 5.0 + 7.0) / 6893.0, rtol=1e-6)


def test_encoder_inventory():
    assert len(resnet_spec.CONV_SPECS) == 53
    assert resnet_spec.encoder_param_count() == 23587712
    assert resnet_spec.encoder_macs_per_image() == 3855925248
    n3 = sum(1 for s in resnet_spec.CONV_SPECS if s.kh == 3)
    n1s2 = sum(1 for s in resnet_spec.CONV_SPECS if s.kh == 1 and s.stride == 2)
    assert (n3, n1s2) == (16, 6)  # 3 blocks x (2a + shortcut) have stride 2


def test_mean_param_layout():
    mean = O.load_mean_param(synthetic.make_mean_params(zero=True))
    exp = np.zeros((1, 85), np.float32)
    exp[0, 0] = 0.9
    exp[0, 3] = math.pi
    np.testing.assert_array_equal(mean, exp)


def test_regressor_shapes_and_init():
    p = synthetic.make_regressor_params()
    assert p["dense_0/kernel"].shape == (2133, 1024) and p["dense_2/kernel"].shape == (1024, 85)
    assert abs(np.abs(p["dense_2/kernel"]).max() - math.sqrt(0.06 / 1109)) < 1e-4
    assert sum(v.size for v in p.values()) == 3321941


def test_fp32_oracle_close_to_fp64(smpl_model):
    th = synthetic.make_thetas(4)
    v32, j32, _ = O.SMPL(smpl_model, dtype=np.float32)(th[:, 75:], th[:, 3:75], get_skin=True)
    v64, j64, _ = O.SMPL(smpl_model, dtype=np.float64)(th[:, 75:], th[:, 3:75], get_skin=True)
    assert np.abs(v32 - v64).max() / np.abs(v64).max() < 5e-6
    assert np.abs(j32 - j64).max() / np.abs(j64).max() < 5e-6


def test_rodrigues_matches_scipy_rotvec():
    """independent implementation of the axis-angle map (scipy); the reference's 1e-8 quirk moves it by < 1e-7"""
    from scipy.spatial.transform import Rotation

    th = synthetic.make_thetas(32, seed=21)[:, 3:75].reshape(-1, 3).astype(np.float64)
    R = O.batch_rodrigues(th)
    np.testing.assert_allclose(R, Rotation.from_rotvec(th).as_matrix(), atol=2e-7)


def test_fk_matches_direct_chain_product(smpl_model):
    """batch_global_rigid_transformation against an explicit root-to-joint product of 4x4 matrices"""
    g = np.random.Generator(np.random.Philox(22))
    Rs = O.batch_rodrigues(g.normal(0, 0.5, (24, 3))).reshape(1, 24, 3, 3)
    Js = g.normal(0, 0.3, (1, 24, 3))
    par = smpl_model["kintree_table"][0].astype(np.int32)
    newJ, A = O.batch_global_rigid_transformation(Rs, Js, par)
    for j in (0, 5, 15, 23):
        chain = []
        k = j
        while k >= 0:
            chain.append(k)
            k = par[k]
        G = np.eye(4)
        for k in reversed(chain):
            T = np.eye(4)
            T[:3, :3] = Rs[0, k]
            T[:3, 3] = Js[0, k] - (Js[0, par[k]] if par[k] >= 0 else 0)
            G = G @ T
        np.testing.assert_allclose(newJ[0, j], G[:3, 3], atol=1e-12)
        np.testing.assert_allclose(A[0, j, :3, :3], G[:3, :3], atol=1e-12)
        np.testing.assert_allclose(A[0, j, :3, 3], G[:3, 3] - G[:3, :3] @ Js[0, j], atol=1e-12)


def test_lbs_matches_textbook_form(smpl_model):
    """SMPL.__call__ against the textbook linear-blend-skinning sum  v' = sum_j w_j (G_j [v_posed - J_j; 0] + t_Gj)
    written out per vertex (no relative-transform trick)."""
    sm = O.SMPL(smpl_model, dtype=np.float64)
    th = synthetic.make_thetas(2, seed=23).astype(np.float64)
    verts, _, Rs = sm(th[:, 75:], th[:, 3:75], get_skin=True)
    vt = smpl_model["v_template"].astype(np.float64)
    sd = smpl_model["shapedirs"].astype(np.float64)
    pd = smpl_model["posedirs"].astype(np.float64)
    W = smpl_model["weights"].astype(np.float64)
    Jr = smpl_model["J_regressor"].astype(np.float64)
    par = smpl_model["kintree_table"][0].astype(np.int32)
    for b in range(2):
        v_shaped = vt + sd @ th[b, 75:]
        J = Jr @ v_shaped
        pf = (Rs[b, 1:] - np.eye(3)).reshape(207)
        v_posed = v_shaped + pd @ pf
        G = [None] * 24
        for j in range(24):
            T = np.eye(4)
            T[:3, :3] = Rs[b, j]
            T[:3, 3] = J[j] - (J[par[j]] if par[j] >= 0 else 0)
            G[j] = T if par[j] < 0 else G[par[j]] @ T
        out = np.zeros_like(vt)
        for j in range(24):
            out += W[:, j:j + 1] * ((v_posed - J[j]) @ G[j][:3, :3].T + G[j][:3, 3])
        np.testing.assert_allclose(verts[b], out, atol=1e-10)


def test_bidirectional_dist_matches_kdtree():
    """Independent pin for src/ops.py:60-102: the oracle's expanded-form argmin (-2AB^T + |A|^2 + |B|^2) against scipy's
    cKDTree nearest neighbours on generic (tie-free) float point sets -- L2 from B to its neighbour in A, L1 from A to its
    neighbour in B -- and the per-image scaling by 3 + 6890 of mesh_reprojection_loss."""
    from scipy.spatial import cKDTree

    g = np.random.Generator(np.random.Philox(404))
    A = g.uniform(0, 224, (1500, 2))
    B = g.uniform(-20, 244, (6890, 2))
    da, ia = cKDTree(B).query(A)   # A -> nearest in B
    db, ib = cKDTree(A).query(B)   # B -> nearest in A
    ref = db.sum() + np.abs(A - B[ia]).sum()
    got = O.bidirectional_dist(A.astype(np.float64), B.astype(np.float64))
    assert abs(got - ref) / ref < 1e-12
    got32 = O.bidirectional_dist(A.astype(np.float32), B.astype(np.float32))
    assert abs(got32 - ref) / ref < 1e-5
    # mesh_reprojection_loss on a two-image batch: rows (b, y, x) as tf.where gives them; second image empty-ish (one pixel)
    seg = np.zeros((2, 224, 224, 1), np.float32)
    seg[0, 50:90, 60:100] = 1.0
    seg[1, 10, 20] = 1.0
    V = g.uniform(0, 224, (2, 6890, 2)).astype(np.float64)
    pts = O.silhouette_points(seg).astype(np.float64)
    total = 0.0
    for i in range(2):
        rows = pts[pts[:, 0] == i]
        P = np.stack([rows[:, 2], rows[:, 1]], 1)
        _, ja = cKDTree(V[i]).query(P)
        d2, _ = cKDTree(P).query(V[i])
        total += (d2.sum() + np.abs(P - V[i][ja]).sum()) / (3 + 6890)
    got = O.mesh_reprojection_loss(pts, V, 2)
    assert abs(got - total) / total < 1e-12
