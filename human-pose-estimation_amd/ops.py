"""Loss forwards on the path's outputs (reference: src/ops.py:35-137), HIP-backed (BASELINE config 5)."""
from __future__ import annotations

from . import engine as _engine


def kp_reprojection_loss(kp_gt, kp_pred, scale=1.0, name="kp_reprojection_loss", return_parts=False):
    """kp_gt [N,K,3] (x, y, vis), kp_pred [N,K,2] -> sum(vis*|d|) / (2*#visible), 0 if none visible
    (tf.compat.v1.losses.absolute_difference, SUM_BY_NONZERO_WEIGHTS; src/ops.py:35-47).
    return_parts=True returns the tensor [numerator, count, loss] so ranks can all-reduce before dividing."""
    parts = _engine.kp_loss_parts(kp_gt, kp_pred)
    return parts if return_parts else parts[2]


def mesh_reprojection_loss(engine, seg_gts, silhouette_pred, name="mesh_reprojection_loss"):
    """seg_gts [N,H,W(,1)] (> 0 = silhouette), silhouette_pred [N,6890,2] pixels -> scalar
    sum_i bidirectional_dist_i / (3 + 6890)   (src/ops.py:117-137 with src/trainer.py:291 folded in:
    the reference first builds tf.where(seg > 0); here the compaction is a kernel of the same call)."""
    if seg_gts.dim() == 4:
        seg_gts = seg_gts[..., 0]
    return engine.mesh_loss(seg_gts.contiguous(), silhouette_pred)
