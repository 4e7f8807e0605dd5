"""Predictor -- drop-in for the reference's inference driver (reference: src/predictor.py:26-163).

Same constructor argument (a ``config`` object read by attribute), same public attributes
(``num_cam=3, num_theta=72, total_params=85, num_joints=14, proj_fn, smpl, mean_var``), same methods
(``predict``, ``predict_single_image``, ``load_mean_param``) and the same result dict keys
(``generated_joints [B,19,3]``, ``generated_verts [B,6890,3]``, ``generated_cams [B,3]`` -- last IEF stage),
plus a superset: ``theta``, ``J_transformed``, ``generated_kp2d`` (= ``proj_fn(joints, cams)``).

Differences, all deliberate:
  * results are torch CUDA tensors (the reference returns TF eager tensors);
  * ``load_path`` / ``pretrained_model_path`` are optional (no reference flag defines them, F9);
  * any batch size <= ``config.batch_size`` works (the reference requires equality, F10); larger inputs are
    processed in chunks of ``config.batch_size``;
  * the renderer, optimizers and critic the reference instantiates as import/ctor side effects are dropped;
  * assets come from ``config.smpl_model_path`` (the SMPL pickle, read by an allow-list unpickler, or .npz) /
    ``neutral_smpl_mean_params.{npz,h5}`` next to it / ``<checkpoint_dir>``: ``weights.npz`` (Keras-layout names) or the
    reference's own ``tf.train.Checkpoint`` files (``tf_checkpoint.py``, no TensorFlow needed),
    or are passed directly as dicts (``smpl_model=``, ``mean_params=``, ``encoder_params=``, ``regressor_params=``).
    Missing weights are an error here (the reference silently keeps random init when no checkpoint exists).
"""
from __future__ import annotations

import os
from os.path import dirname, join

import numpy as np

from . import engine as _engine
from .projection import batch_orth_proj_idrot
from .smpl import SMPL
from . import assets


def _load_smpl_file(path):
    return assets.load_smpl_model(path)  # allow-list unpickler / .npz (batch_smpl.py:31-81)


def _load_mean_file(smpl_model_path):
    return assets.load_mean_params(smpl_model_path)  # .npz or the deepdish .h5 (predictor.py:93-95)


class Predictor(object):
    def __init__(self, config, smpl_model=None, mean_params=None, encoder_params=None, regressor_params=None, device=None):
        import torch

        # ---- config information (reference :31-41; load_path / pretrained_model_path optional, F9)
        self.model_dir = getattr(config, "model_dir", None)
        self.load_path = getattr(config, "load_path", None)
        self.data_format = getattr(config, "data_format", "NHWC")
        self.smpl_model_path = getattr(config, "smpl_model_path", None)
        self.pretrained_model_path = getattr(config, "pretrained_model_path", None)
        self.img_size = getattr(config, "img_size", 224)
        self.num_stage = getattr(config, "num_stage", 3)
        self.batch_size = getattr(config, "batch_size", 8)
        self.checkpoint_dir = getattr(config, "checkpoint_dir", None)
        if self.img_size != 224:
            raise ValueError("img_size must be 224 (the encoder plan is built for 224x224 inputs)")
        self.num_joints = 14
        self.proj_fn = batch_orth_proj_idrot
        self.num_cam = 3
        self.num_theta = 72
        self.total_params = self.num_theta + self.num_cam + 10

        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.engine = _engine.HpeEngine(
            device=device, max_batch=self.batch_size, num_stage=self.num_stage, bn_eps=getattr(config, "bn_eps", 1e-3),
            encoder_dtype=getattr(config, "encoder_dtype", "fp32"), **(getattr(config, "plan_options", None) or {}),
        )
        # ---- SMPL (reference :55)
        if smpl_model is None:
            smpl_model = getattr(config, "smpl_model", None)
        if smpl_model is None:
            smpl_model = _load_smpl_file(self.smpl_model_path)
        self.engine.load_smpl(smpl_model, joint_type="cocoplus")  # the reference never reads config.joint_type (F7)
        # ---- mean theta (reference :61,85)
        self._mean_params = mean_params if mean_params is not None else getattr(config, "mean_params", None)
        self.mean_np = self.load_mean_param()
        self.engine.load_mean_theta(self.mean_np)
        # ---- networks (reference :73-86: restore feature_extractor / generator3d from the checkpoint)
        w, self.checkpoint_info = None, None
        if encoder_params is None or regressor_params is None:
            w = getattr(config, "weights", None)
            if w is None and self.checkpoint_dir and os.path.exists(join(self.checkpoint_dir, "weights.npz")):
                with np.load(join(self.checkpoint_dir, "weights.npz"), allow_pickle=False) as z:
                    w = {k: z[k] for k in z.files}
            if w is None and self.checkpoint_dir and os.path.exists(join(self.checkpoint_dir, "checkpoint")):
                # what the reference restores (:77-86): the newest tf.train.Checkpoint in checkpoint_dir, read without TF
                from . import tf_checkpoint

                w, self.checkpoint_info = tf_checkpoint.load_hmr_weights(self.checkpoint_dir)
            if w is None:
                raise FileNotFoundError(
                    "no encoder/regressor weights: pass encoder_params/regressor_params, config.weights, or put a "
                    "TensorFlow checkpoint (checkpoint + ckpt-N.index/.data-*) or weights.npz (Keras-layout names) in "
                    "config.checkpoint_dir"
                )
            encoder_params = encoder_params or w
            regressor_params = regressor_params or w
        self.engine.load_encoder(encoder_params)
        self.engine.load_regressor(regressor_params)
        self.engine.finalize()
        self.smpl = SMPL(None, engine=self.engine)
        self.mean_var = torch.from_numpy(self.mean_np).to(self.engine.tdev)
        # the reference tracks `theta_prev` in its checkpoint as "inital_theta" (:84) but predicts from `mean_var` (:126)
        self.theta_prev = self.mean_var
        if w is not None and "inital_theta" in w:
            self.theta_prev = torch.from_numpy(np.asarray(w["inital_theta"], np.float32)).to(self.engine.tdev)

    def load_mean_param(self):
        """reference: src/predictor.py:88-110 -- zeros(1,85); [0,0]=0.9; pose[:3]=0 then pose[0]=pi; shape."""
        mean = np.zeros((1, self.total_params))
        mean[0, 0] = 0.9
        mean_vals = self._mean_params if self._mean_params is not None else _load_mean_file(self.smpl_model_path)
        mean_pose = np.array(mean_vals["pose"], dtype=np.float64).copy()
        mean_pose[:3] = 0.0
        mean_shape = np.array(mean_vals["shape"], dtype=np.float64)
        mean_pose[0] = np.pi
        mean[0, 3:] = np.hstack((mean_pose, mean_shape))
        return mean.astype(np.float32)

    def _to_device(self, images):
        import torch

        if isinstance(images, np.ndarray):
            images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32))
        if not images.is_cuda:
            images = images.to(self.engine.tdev, non_blocking=True)
        return images.float()

    def predict(self, images, all_stages=False):
        """images [B,224,224,3] NHWC float32 in [-1,1] (numpy or torch; reference: src/predictor.py:114-158).
        The reference transposes to NCHW itself when data_format == 'NCHW' -- the input is always NHWC."""
        import torch

        images = self._to_device(images)
        if images.dim() != 4 or tuple(images.shape[1:]) != (224, 224, 3):
            raise ValueError("images must be [B,224,224,3] NHWC, got %s" % (tuple(images.shape),))
        B = images.shape[0]
        chunks = []
        many = B > self.batch_size  # several engine calls: software-pipeline them (tail of chunk k under the encoder of chunk k+1)
        try:
            for lo in range(0, B, self.batch_size):
                chunks.append(self.engine.forward(images[lo : lo + self.batch_size], all_stages=all_stages, pipelined=many))
        finally:
            # the caller's stream waits for the last tail: results are then ordered like any torch op -- also when a later chunk
            # raised, so that the earlier chunks' outputs are not freed (and recycled by the allocator) under a tail still writing
            if many:
                self.engine.join()
        if len(chunks) == 1:
            stages = chunks[0]
        else:
            stages = [{k: torch.cat([c[i][k] for c in chunks], 0) for k in chunks[0][i]} for i in range(len(chunks[0]))]
        last = stages[-1]
        result = {
            "generated_joints": last["joints"],
            "generated_verts": last["verts"],
            "generated_cams": last["cams"],
            "theta": last["theta"],
            "J_transformed": last["J_transformed"],
            "generated_kp2d": last["kp2d"],
        }
        if all_stages:
            result["stages"] = stages
        return result

    def val_step(self, images, seg_gts, kp2d_gts, use_mesh_repro_loss=True, kpr_loss_weight=60.0, mr_loss_weight=0.001,
                 reduce_fn=None):
        """Forward + reprojection losses of every IEF stage, i.e. the critic-free part of the reference's
        ``Trainer.val_step`` (src/trainer.py:226-348; BASELINE config 5): per stage
        ``kpr = 60 * kp_reprojection_loss(kp2d_gts, proj_fn(joints, cams))`` (:274-281) and
        ``mr = 0.001 * mesh_reprojection_loss(where(seg > 0), reproject_vertices(verts, cams, [224, 224]), B)``
        (:285-296).  All stages' (kp numerator, kp count, kp loss, mesh sum) come back from ONE library call
        (``hpe_val_losses``: the silhouette-only work is done once per step) as a ``[n_stage, 4]`` tensor;
        ``reduce_fn(packed) -> packed`` (``distributed.reduce_losses``) lets ranks sum that block in a single all-reduce
        before the kp division; default is single-process."""
        import torch

        images = self._to_device(images)
        seg = self._to_device(seg_gts)
        if seg.dim() == 4:
            seg = seg[..., 0]
        seg = seg.contiguous()
        kp_gt = self._to_device(kp2d_gts).contiguous()
        B = images.shape[0]
        if B > self.batch_size:
            raise ValueError("val_step needs B <= config.batch_size")
        want = ("verts", "joints", "cams", "theta", "kp2d") + (("verts2d",) if use_mesh_repro_loss else ())
        stages = self.engine.forward(images, all_stages=True, want=want)
        packed = self.engine.val_losses(kp_gt, [st["kp2d"] for st in stages], seg if use_mesh_repro_loss else None,
                                        [st["verts2d"] for st in stages] if use_mesh_repro_loss else None)
        if reduce_fn is not None:
            packed = reduce_fn(packed)
        kpr = [packed[i, 2] * kpr_loss_weight for i in range(len(stages))]
        result = {"kpr_losses": kpr, "pred_keypoints": torch.stack([s["kp2d"] for s in stages], 1),
                  "generated_verts": torch.stack([s["verts"] for s in stages], 1),
                  "generated_cams": torch.stack([s["cams"] for s in stages], 1), "loss_parts": packed}
        if use_mesh_repro_loss:
            result["mr_losses"] = [packed[i, 3] * mr_loss_weight for i in range(len(stages))]
        return result

    def predict_single_image(self, image):
        """reference: src/predictor.py:160-163"""
        image = self._to_device(image)
        pred_results = self.predict(image.unsqueeze(0))
        return pred_results["generated_verts"], pred_results["generated_cams"], pred_results["generated_joints"]
