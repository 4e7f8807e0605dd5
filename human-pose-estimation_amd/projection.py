"""Orthographic reprojection operators (reference: src/tf_smpl/projection.py:23-56), HIP-backed."""
from __future__ import annotations

from . import engine as _engine


def batch_orth_proj_idrot(X, camera, name=None):
    """X [N,P,3], camera [N,3] = (s, tx, ty) -> s * (X[:, :, :2] + [tx, ty])   (projection.py:23-33)"""
    return _engine.orth_proj(X, camera)


def reproject_vertices(verts, cam, im_size, name=None):
    """verts [N,6890,3], cam [N,3], im_size (w, h) -> pixels = (proj + 1) * 0.5 * im_size  (projection.py:45-56)"""
    try:
        w, h = float(im_size[0]), float(im_size[1])
    except TypeError:
        w = h = float(im_size)
    return _engine.reproject(verts, cam, w, h)
