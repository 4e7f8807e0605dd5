"""SMPL -- GPU mesh generator with the reference's call surface (reference: src/tf_smpl/batch_smpl.py:25-160):
``SMPL(pkl_path_or_dict)(beta, theta, get_skin=False)`` -> joints, or (verts, joints, Rs) with get_skin=True,
and the side attribute ``J_transformed`` ([N,24,3]) updated per call, exactly like the reference (and, like
it, therefore not thread-safe).  beta [N,10], theta [N,72] are torch CUDA float32 tensors (or numpy).
"""
from __future__ import annotations

import numpy as np

from . import engine as _engine


class SMPL(object):
    def __init__(self, pkl_path, joint_type="cocoplus", dtype=None, engine=None, device=None, max_batch=64):
        if engine is None:
            import torch

            if isinstance(pkl_path, dict):
                model = pkl_path
            else:
                from .predictor import _load_smpl_file

                model = _load_smpl_file(pkl_path)
            if device is None:
                device = torch.cuda.current_device() if torch.cuda.is_available() else 0
            engine = _engine.HpeEngine(device=device, max_batch=max_batch)
            engine.load_smpl(model, joint_type=joint_type)
            engine.finalize()
        self.engine = engine
        self.size = [_engine.NUM_VERTS, 3]
        self.num_betas = 10
        self.J_transformed = None

    def __call__(self, beta, theta, get_skin=False, name=None):
        import torch

        dev = self.engine.tdev
        beta = torch.as_tensor(np.asarray(beta) if not isinstance(beta, torch.Tensor) else beta, dtype=torch.float32).to(dev)
        theta = torch.as_tensor(np.asarray(theta) if not isinstance(theta, torch.Tensor) else theta, dtype=torch.float32).to(dev)
        N = beta.shape[0]
        if tuple(beta.shape) != (N, 10) or tuple(theta.shape) != (N, 72):
            raise ValueError("beta must be [N,10] and theta [N,72]")
        full = torch.zeros((N, 85), dtype=torch.float32, device=dev)
        full[:, 0] = 1.0
        full[:, 3:75] = theta
        full[:, 75:] = beta
        outs = []
        mb = self.engine.max_batch
        for lo in range(0, N, mb):
            outs.append(self.engine.smpl(full[lo : lo + mb], want=("verts", "joints", "J_transformed", "Rs")))
        cat = {k: torch.cat([o[k] for o in outs], 0) if len(outs) > 1 else outs[0][k] for k in outs[0]}
        self.J_transformed = cat["J_transformed"]
        if get_skin:
            return cat["verts"], cat["joints"], cat["Rs"]
        return cat["joints"]
