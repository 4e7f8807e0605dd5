"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED** (see below).

A line-by-line CPU restatement (NumPy for the SMPL / regressor / projection / loss arithmetic,
torch-CPU ``conv2d`` for the encoder convolutions) of the reference's per-image forward hot path:

    src/predictor.py:114-158  ->  src/models.py:35-41,60-74  ->  src/tf_smpl/batch_smpl.py:88-160
    ->  src/tf_smpl/batch_lbs.py:15-64,91-152  ->  src/tf_smpl/projection.py:23-56  (+ src/ops.py:35-137)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product (``human-pose-estimation_amd``)
never imports it and has no CPU fallback.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, and
its own code cannot be imported in the build container (TensorFlow/Keras/absl/deepdish/opendr/cv2 are
absent -- plain absence, nothing was denied).  The encoder arithmetic is not in the reference repo at
all (it is ``tensorflow.keras.applications.ResNet50``, TF pinned by README to 2.0.0-beta1 wrapping
keras_applications 1.0.8 ``resnet50.py``); its topology is restated here from that published
definition.  What pins this oracle instead: the closed-form known-answer tests of SURVEY.md §4
(``tests/test_oracle_kat.py``), an fp64 run of the same restatement, and -- for the encoder topology -- agreement to 1e-9
(fp64) with an unrelated third-party implementation of ResNet-50 v1, Hugging Face transformers' PyTorch ``ResNetModel``
with ``downsample_in_bottleneck=True`` (``tests/test_oracle_vs_hf_resnet.py``).  None of these is the reference's own
TensorFlow run, hence 'unpinned'.

Every function cites the reference lines it follows.  ``dtype`` is float32 for the reference-equivalent
path and float64 for the "truth" used to rank fp32 implementations against each other.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# batch_lbs.py
# ----------------------------------------------------------------------------------------------


def batch_skew(vec):
    """reference: src/tf_smpl/batch_lbs.py:15-39.  vec [N,3] -> [N,3,3] skew-symmetric.
    scatter columns 1,2,3,5,6,7 <- [-z, y, z, -x, -y, x]."""
    n = vec.shape[0]
    res = np.zeros((n, 9), vec.dtype)
    col_inds = [1, 2, 3, 5, 6, 7]
    updates = np.stack([-vec[:, 2], vec[:, 1], vec[:, 2], -vec[:, 0], -vec[:, 1], vec[:, 0]], axis=1)
    res[:, col_inds] = updates
    return res.reshape(n, 3, 3)


def batch_rodrigues(theta):
    """reference: src/tf_smpl/batch_lbs.py:42-64.  theta [N,3] -> R [N,3,3].
    NB the quirk: angle = ||theta + 1e-8|| (epsilon added per component before the norm, :52), while
    r = theta / angle uses theta WITHOUT the epsilon (:53)."""
    dt = theta.dtype
    n = theta.shape[0]
    eps = dt.type(1e-8)
    angle = np.sqrt(np.sum(np.square(theta + eps), axis=1))[:, None]  # tf.norm(theta + 1e-8, axis=1)
    r = (theta / angle)[:, :, None]  # [N,3,1]
    angle = angle[:, :, None]
    cos = np.cos(angle)
    sin = np.sin(angle)
    outer = np.matmul(r, np.transpose(r, (0, 2, 1)))
    eyes = np.tile(np.eye(3, dtype=dt)[None], (n, 1, 1))
    R = cos * eyes + (dt.type(1) - cos) * outer + sin * batch_skew(r[:, :, 0])
    return R.astype(dt)


def batch_global_rigid_transformation(Rs, Js, parent):
    """reference: src/tf_smpl/batch_lbs.py:91-152 (rotate_base=False).
    Rs [N,24,3,3], Js [N,24,3] -> new_J [N,24,3], A [N,24,4,4]."""
    dt = Rs.dtype
    N = Rs.shape[0]
    root_rotation = Rs[:, 0, :, :]
    Js = Js[:, :, :, None]  # N x 24 x 3 x 1

    def make_A(R, t):
        R_homo = np.pad(R, [[0, 0], [0, 1], [0, 0]])  # N x 4 x 3
        t_homo = np.concatenate([t, np.ones((N, 1, 1), dt)], 1)  # N x 4 x 1
        return np.concatenate([R_homo, t_homo], 2)

    A0 = make_A(root_rotation, Js[:, 0])
    results = [A0]
    for i in range(1, parent.shape[0]):
        j_here = Js[:, i] - Js[:, parent[i]]
        A_here = make_A(Rs[:, i], j_here)
        res_here = np.matmul(results[parent[i]], A_here)
        results.append(res_here)
    results = np.stack(results, axis=1)  # N x 24 x 4 x 4
    new_J = results[:, :, :3, 3]
    Js_w0 = np.concatenate([Js, np.zeros((N, 24, 1, 1), dt)], 2)
    init_bone = np.matmul(results, Js_w0)
    init_bone = np.pad(init_bone, [[0, 0], [0, 0], [0, 0], [3, 0]])
    A = results - init_bone
    return new_J.astype(dt), A.astype(dt)


# ----------------------------------------------------------------------------------------------
# batch_smpl.py
# ----------------------------------------------------------------------------------------------


class SMPL(object):
    """reference: src/tf_smpl/batch_smpl.py:25-160.  ``dd`` is the dict the reference unpickles from
    model.pkl (here: already-dense numpy arrays, e.g. ``synthetic.make_smpl_model()``)."""

    def __init__(self, dd, joint_type="cocoplus", dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.v_template = np.asarray(dd["v_template"], dtype)  # :33-37
        self.size = [self.v_template.shape[0], 3]  # :39
        self.num_betas = dd["shapedirs"].shape[-1]  # :40
        self.shapedirs = np.reshape(np.asarray(dd["shapedirs"]), [-1, self.num_betas]).T.astype(dtype)  # :43-47
        self.J_regressor = np.asarray(dd["J_regressor"]).T.astype(dtype)  # :50-54  [6890,24]
        num_pose_basis = dd["posedirs"].shape[-1]  # :57
        self.posedirs = np.reshape(np.asarray(dd["posedirs"]), [-1, num_pose_basis]).T.astype(dtype)  # :59-62
        self.parents = np.asarray(dd["kintree_table"])[0].astype(np.int32)  # :65
        self.weights = np.asarray(dd["weights"], dtype)  # :68-72
        self.joint_regressor = np.asarray(dd["cocoplus_regressor"]).T.astype(dtype)  # :75-79 [6890,19]
        if joint_type == "lsp":
            self.joint_regressor = self.joint_regressor[:, :14]  # :80-81
        if joint_type not in ["cocoplus", "lsp"]:
            raise ValueError("BAD!! Unknown joint type: %s" % joint_type)  # :83-86 (ipdb in the reference)
        self.J_transformed = None

    def __call__(self, beta, theta, get_skin=False):
        dt = self.dtype
        beta = np.asarray(beta, dt)
        theta = np.asarray(theta, dt)
        num_batch = beta.shape[0]
        # 1. shape blend shapes (:110-112)
        v_shaped = np.reshape(np.matmul(beta, self.shapedirs), [-1, self.size[0], self.size[1]]) + self.v_template
        # 2. joint locations (:115-118)
        Jx = np.matmul(v_shaped[:, :, 0], self.J_regressor)
        Jy = np.matmul(v_shaped[:, :, 1], self.J_regressor)
        Jz = np.matmul(v_shaped[:, :, 2], self.J_regressor)
        J = np.stack([Jx, Jy, Jz], axis=2)
        # 3. pose blend shapes (:122-132)
        Rs = np.reshape(batch_rodrigues(np.reshape(theta, [-1, 3])), [-1, 24, 3, 3])
        pose_feature = np.reshape(Rs[:, 1:, :, :] - np.eye(3, dtype=dt), [-1, 207])
        v_posed = np.reshape(np.matmul(pose_feature, self.posedirs), [-1, self.size[0], self.size[1]]) + v_shaped
        # 4. global joint locations (:135)
        self.J_transformed, A = batch_global_rigid_transformation(Rs, J, self.parents)
        # 5. skinning (:139-149)
        W = np.reshape(np.tile(self.weights, [num_batch, 1]), [num_batch, -1, 24])
        T = np.reshape(np.matmul(W, np.reshape(A, [num_batch, 24, 16])), [num_batch, -1, 4, 4])
        v_posed_homo = np.concatenate([v_posed, np.ones([num_batch, v_posed.shape[1], 1], dt)], 2)
        v_homo = np.matmul(T, v_posed_homo[..., None])
        verts = v_homo[:, :, :3, 0]
        # keypoints (:152-155)
        joint_x = np.matmul(verts[:, :, 0], self.joint_regressor)
        joint_y = np.matmul(verts[:, :, 1], self.joint_regressor)
        joint_z = np.matmul(verts[:, :, 2], self.joint_regressor)
        joints = np.stack([joint_x, joint_y, joint_z], axis=2)
        if get_skin:
            return verts, joints, Rs
        return joints


# ----------------------------------------------------------------------------------------------
# projection.py
# ----------------------------------------------------------------------------------------------


def batch_orth_proj_idrot(X, camera):
    """reference: src/tf_smpl/projection.py:23-33.  X [N,P,3], camera [N,3] -> [N,P,2]."""
    camera = np.reshape(camera, [-1, 1, 3])
    X_trans = X[:, :, :2] + camera[:, :, 1:]
    shape = X_trans.shape
    return np.reshape(camera[:, :, 0] * np.reshape(X_trans, [shape[0], -1]), shape)


def reproject_vertices(verts, cam, im_size):
    """reference: src/tf_smpl/projection.py:45-56.  pixels = (proj + 1) * 0.5 * im_size."""
    verts_reprojected = batch_orth_proj_idrot(verts, cam)
    verts_calc = (verts_reprojected + np.ones_like(verts_reprojected)) * verts.dtype.type(0.5)
    return verts_calc * np.asarray(im_size, verts.dtype)


# ----------------------------------------------------------------------------------------------
# ops.py (forward only)
# ----------------------------------------------------------------------------------------------


def kp_reprojection_loss(kp_gt, kp_pred):
    """reference: src/ops.py:35-47 -> tf.compat.v1.losses.absolute_difference(gt[:, :2], pred, weights=vis)
    with the default Reduction.SUM_BY_NONZERO_WEIGHTS: sum(|d| * w) / #nonzero(w broadcast to |d|.shape),
    0 when nothing is visible (safe-div)."""
    kp_gt = np.reshape(kp_gt, (-1, 3))
    kp_pred = np.reshape(kp_pred, (-1, 2))
    vis = kp_gt[:, 2].astype(np.float32)[:, None]
    losses = np.abs(kp_pred - kp_gt[:, :2]) * vis
    num_present = np.count_nonzero(np.broadcast_to(vis, losses.shape))
    total = np.sum(losses, dtype=losses.dtype)
    return total / losses.dtype.type(num_present) if num_present > 0 else losses.dtype.type(0)


def find_nearest_neighbors(A, B):
    """reference: src/ops.py:60-71.  D = -2 A B^T + |A|^2 + |B|^2 ; argmin over each axis
    (lowest index wins ties)."""
    dists = (
        A.dtype.type(-2.0) * np.matmul(A, B.T)
        + np.sum(np.square(A), 1)[:, None]
        + np.sum(np.square(B), axis=1)[None, :]
    )
    return np.argmin(dists, 1), np.argmin(dists, 0)


def bidirectional_dist(A, B):
    """reference: src/ops.py:83-102.  L2 from B to its NN in A, L1 from A to its NN in B."""
    ind_AB, ind_BA = find_nearest_neighbors(A, B)
    d = B - A[ind_BA]
    dist_BA = np.sqrt(np.sum(d * d, axis=1))
    dist_AB = np.sum(np.abs(A - B[ind_AB]), axis=1)
    return np.sum(dist_BA) + np.sum(dist_AB)


def silhouette_points(seg_gts):
    """reference: src/trainer.py:291  tf.cast(tf.where(seg_gts > 0)[:, :3], float32) -> rows (b, y, x)."""
    idx = np.argwhere(seg_gts > 0.0)[:, :3]
    return idx.astype(np.float32)


def mesh_reprojection_loss(silhouette_gt, silhouette_pred, batch_size):
    """reference: src/ops.py:117-137.  Per image: x = column 2, y = column 1 of the (b,y,x) rows;
    bi_loss / (silhouette_gt.shape[1] + silhouette_pred.shape[1]) = bi_loss / (3 + 6890); summed."""
    loss = None
    for i in range(batch_size):
        rows = silhouette_gt[silhouette_gt[:, 0] == i]
        pts = np.stack([rows[:, 2], rows[:, 1]], axis=1).astype(silhouette_pred.dtype)
        bi_loss = bidirectional_dist(pts, silhouette_pred[i, :, :])
        bi_loss_scaled = bi_loss / (silhouette_gt.shape[1] + silhouette_pred.shape[1])
        loss = bi_loss_scaled if i == 0 else loss + bi_loss_scaled
    return loss


# ----------------------------------------------------------------------------------------------
# models.py
# ----------------------------------------------------------------------------------------------


def regression_network(state, p):
    """reference: src/models.py:60-74.  Dense(1024,relu) -> Dropout (identity at inference) ->
    Dense(1024,relu) -> Dropout -> Dense(85); Keras Dense: y = x @ kernel[in,out] + bias."""
    dt = state.dtype
    h = np.maximum(np.matmul(state, p["dense_0/kernel"].astype(dt)) + p["dense_0/bias"].astype(dt), 0)
    h = np.maximum(np.matmul(h, p["dense_1/kernel"].astype(dt)) + p["dense_1/bias"].astype(dt), 0)
    return np.matmul(h, p["dense_2/kernel"].astype(dt)) + p["dense_2/bias"].astype(dt)


def _t(x, dtype):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def resnet50_features(images, p, eps=1e-3, dtype=np.float32, return_taps=False, act_round=None, bf16_folded=()):
    """reference: src/models.py:35-41 -> keras.applications.ResNet50(include_top=False, pooling='avg')
    (keras_applications 1.0.8 resnet50.py; BN epsilon 1e-3 there, 1.001e-5 in tf.keras>=2.2 resnet.py --
    hence the parameter).  images [B,224,224,3] NHWC in [-1,1] -> features [B,2048].

    ZeroPad(3) -> conv1 7x7/2 valid + bias -> BN -> ReLU -> ZeroPad(1) -> MaxPool 3x3/2 valid ->
    stages [3,4,6,3] of bottlenecks (conv_block: stride on 2a and on the projection shortcut; stage 2
    block a has stride 1) -> GlobalAveragePooling2D."""
    import torch
    import torch.nn.functional as F

    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    taps = {}
    # act_round="bf16": emulation of the bf16 encoder variant (BASELINE config 4) -- the input, the conv kernels and
    # every stored activation (after BN / residual add / ReLU) are rounded to bfloat16 (RNE); accumulation, BN
    # scale/shift and the pools stay in `dtype`.  Not a reference behaviour: a checker for the bf16 HIP path.
    # bf16_folded (emulation only): layer names whose HIP kernels fold the BN scale into the weights BEFORE rounding them to
    # bfloat16 (the dual-source conv_block GEMM: "<block>2c" and "<block>1"; there the shortcut is also summed in the fp32
    # accumulator instead of being stored as a bf16 tensor).  Mathematically the same layer, different rounding points.
    rnd = (lambda v: v.to(torch.bfloat16).to(tdt)) if act_round == "bf16" else (lambda v: v)

    def conv(x, name, stride=1, padding=0):
        w = _t(p[name + "/kernel"], tdt)
        if name in bf16_folded:
            bnn = "bn" + name[3:]
            w = w * (_t(p[bnn + "/gamma"], tdt) * torch.rsqrt(_t(p[bnn + "/moving_variance"], tdt) + eps))  # HWIO: scale the O axis
        w = rnd(w).permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW
        return F.conv2d(x, w, None if name in bf16_folded else _t(p[name + "/bias"], tdt), stride=stride, padding=padding)

    def bn(x, name):
        if "res" + name[2:] in bf16_folded:  # scale already in the weights: only the shift ((bias - mean) * scale + beta) is left
            sc_ = _t(p[name + "/gamma"], tdt) * torch.rsqrt(_t(p[name + "/moving_variance"], tdt) + eps)
            sh_ = (_t(p["res" + name[2:] + "/bias"], tdt) - _t(p[name + "/moving_mean"], tdt)) * sc_ + _t(p[name + "/beta"], tdt)
            return x + sh_[None, :, None, None]
        g = _t(p[name + "/gamma"], tdt)[None, :, None, None]
        b = _t(p[name + "/beta"], tdt)[None, :, None, None]
        m = _t(p[name + "/moving_mean"], tdt)[None, :, None, None]
        v = _t(p[name + "/moving_variance"], tdt)[None, :, None, None]
        return (x - m) * torch.rsqrt(v + eps) * g + b

    with torch.no_grad():
        x = rnd(_t(images, tdt)).permute(0, 3, 1, 2)
        x = F.pad(x, (3, 3, 3, 3))
        x = rnd(torch.relu(bn(conv(x, "conv1", stride=2), "bn_conv1")))
        taps["conv1"] = x
        x = F.pad(x, (1, 1, 1, 1))
        x = F.max_pool2d(x, 3, stride=2)
        taps["pool1"] = x
        for stage, nblk in ((2, 3), (3, 4), (4, 6), (5, 3)):
            for b in range(nblk):
                blk = "abcdef"[b]
                cn = "res%d%s_branch" % (stage, blk)
                bnn = "bn%d%s_branch" % (stage, blk)
                s = 2 if (b == 0 and stage > 2) else 1
                y = rnd(torch.relu(bn(conv(x, cn + "2a", stride=s), bnn + "2a")))
                y = rnd(torch.relu(bn(conv(y, cn + "2b", padding=1), bnn + "2b")))
                y = bn(conv(y, cn + "2c"), bnn + "2c")
                if b == 0:
                    sc = bn(conv(x, cn + "1", stride=s), bnn + "1")
                    if cn + "1" not in bf16_folded:
                        sc = rnd(sc)  # stored as a bf16 tensor by the two-launch path; summed in fp32 by the dual-source GEMM
                else:
                    sc = x
                x = rnd(torch.relu(y + sc))
                taps["res%d%s" % (stage, blk)] = x
        feat = x.mean(dim=(2, 3))
    out = feat.numpy().astype(dtype)
    if return_taps:
        return out, {k: v.permute(0, 2, 3, 1).numpy() for k, v in taps.items()}
    return out


def conv2d_nhwc(x, kernel_hwio, bias, stride, padding, dtype=np.float32):
    """Single Keras Conv2D (NHWC, HWIO) for per-layer kernel parity tests."""
    import torch
    import torch.nn.functional as F

    tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    with torch.no_grad():
        y = F.conv2d(
            _t(x, tdt).permute(0, 3, 1, 2),
            _t(kernel_hwio, tdt).permute(3, 2, 0, 1).contiguous(),
            None if bias is None else _t(bias, tdt),
            stride=stride,
            padding=padding,
        )
    return y.permute(0, 2, 3, 1).numpy()


# ----------------------------------------------------------------------------------------------
# predictor.py
# ----------------------------------------------------------------------------------------------


def load_mean_param(mean_vals, total_params=85, dtype=np.float32):
    """reference: src/predictor.py:88-110.  mean_vals = {'pose': [72], 'shape': [10]} (the h5 content)."""
    mean = np.zeros((1, total_params))
    mean[0, 0] = 0.9
    mean_pose = np.array(mean_vals["pose"], dtype=np.float64).copy()
    mean_pose[:3] = 0.0
    mean_shape = np.array(mean_vals["shape"], dtype=np.float64)
    mean_pose[0] = np.pi
    mean[0, 3:] = np.hstack((mean_pose, mean_shape))
    return mean.astype(dtype)


def predict(images, enc_p, reg_p, smpl, mean_var, num_stage=3, batch_size=None, eps=1e-3, dtype=np.float32,
            features=None, all_stages=False):
    """reference: src/predictor.py:114-158 (NHWC branch).  Returns the reference's result dict
    (last stage only) plus, as a superset, 'theta', 'J_transformed', 'generated_kp2d'
    (= proj_fn(joints, cams) exactly as Trainer.val_step does, src/trainer.py:274) and 'features'.
    With all_stages=True also returns the per-stage lists (what val_step keeps)."""
    dt = np.dtype(dtype)
    images = np.asarray(images)
    batch_size = images.shape[0] if batch_size is None else batch_size
    extracted_features = resnet50_features(images, enc_p, eps=eps, dtype=dt) if features is None else features.astype(dt)
    theta_prev = np.tile(mean_var.astype(dt), [batch_size, 1])  # :126 (tiles by config.batch_size)
    all_verts, all_cams, all_kps, all_theta, all_j24, all_kp2d = [], [], [], [], [], []
    for _ in range(num_stage):
        state = np.concatenate([extracted_features, theta_prev], 1)
        delta_theta = regression_network(state, reg_p)
        theta_here = theta_prev + delta_theta
        cams = theta_here[:, :3]
        poses = theta_here[:, 3:75]
        shapes = theta_here[:, 75:]
        verts, joints, _Rs = smpl(shapes, poses, get_skin=True)
        all_kps.append(joints)
        all_verts.append(verts)
        all_cams.append(cams)
        all_theta.append(theta_here)
        all_j24.append(smpl.J_transformed)
        all_kp2d.append(batch_orth_proj_idrot(joints, cams))
        theta_prev = theta_here
    result = {
        "generated_joints": all_kps[-1],
        "generated_verts": all_verts[-1],
        "generated_cams": all_cams[-1],
        "theta": all_theta[-1],
        "J_transformed": all_j24[-1],
        "generated_kp2d": all_kp2d[-1],
        "features": extracted_features,
    }
    if all_stages:
        result.update(stage_joints=all_kps, stage_verts=all_verts, stage_cams=all_cams, stage_theta=all_theta,
                      stage_kp2d=all_kp2d, stage_J_transformed=all_j24)
    return result


def val_losses(stage_verts, stage_cams, stage_kp2d, seg_gts, kp2d_gts, img_size=224, kpr_loss_weight=60.0,
               mr_loss_weight=0.001, use_mesh_repro_loss=True):
    """reference: src/trainer.py:272-298 (critic-free part of Trainer.val_step): per stage
    kpr = 60 * kp_reprojection_loss(kp2d_gts, proj(joints, cams)); mr = 0.001 * mesh_reprojection_loss(
    where(seg>0), reproject_vertices(verts, cams, [224,224]), batch)."""
    out = {"kpr_losses": [], "mr_losses": []}
    sil_gt = silhouette_points(seg_gts)
    B = stage_verts[0].shape[0]
    for verts, cams, kp2d in zip(stage_verts, stage_cams, stage_kp2d):
        out["kpr_losses"].append(kpr_loss_weight * kp_reprojection_loss(kp2d_gts, kp2d))
        if use_mesh_repro_loss:
            sil_pred = reproject_vertices(verts, cams, np.array([img_size, img_size], verts.dtype))
            out["mr_losses"].append(mesh_reprojection_loss(sil_gt, sil_pred, B) * mr_loss_weight)
    return out
