"""GPU versions of the steps right before / after the hot path, same names and argument
meaning as the reference: ``preprocess_image`` (preview.py:18-35 -> src/util/image.py:7-39) and ``get_original``
(src/util/renderer.py:260-283).  HIP-backed through include/hpe.h; no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def preprocess_image(img, config=None, img_size=224):
    """img: uint8 [H,W,3|4] (numpy or torch, any device) -> (crop [224,224,3] float32 CUDA tensor in [-1,1],
    proc_param dict {'scale','start_pt','end_pt','img_size'}, img)."""
    import torch

    if config is not None:
        img_size = getattr(config, "img_size", img_size)
    if img_size != 224:
        raise ValueError("img_size must be 224")
    t = torch.as_tensor(np.ascontiguousarray(img) if isinstance(img, np.ndarray) else img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] not in (3, 4):
        raise ValueError("img must be uint8 [H,W,3|4]")
    if not t.is_cuda:
        t = t.cuda()
    t = t.contiguous()
    H, W, Cn = (int(x) for x in t.shape)
    out = torch.empty((224, 224, 3), dtype=torch.float32, device=t.device)
    pp = (C.c_int * 5)()
    with torch.cuda.device(t.device):
        st = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        _lib.check(_lib.load().hpe_preprocess_u8(t.data_ptr(), H, W, Cn, out.data_ptr(), pp, st))
    mx = max(H, W)
    scale = float(img_size) / mx if mx != img_size else 1.0
    proc_param = {"scale": scale, "start_pt": np.array([pp[0], pp[1]]), "end_pt": np.array([pp[2], pp[3]]), "img_size": pp[4]}
    return out, proc_param, img


def preprocess_batch(frames, out=None, img_size=224):
    """``preprocess_image`` for a whole batch in ONE launch (hpe_preprocess_u8_batch).

    frames: a uint8 tensor / array [B,H,W,3|4] (equal frames, e.g. a video stream; host or device), or a list of uint8
    [H_i,W_i,C] arrays / tensors of different sizes (same C; they are packed into one device buffer here).
    -> (crops [B,224,224,3] float32 CUDA tensor in [-1,1], list of per-image proc_param dicts).  ``out`` may be a preallocated
    [B,224,224,3] float32 CUDA tensor (steady-state serving: no allocation, see bench.py --from-host)."""
    import torch

    if img_size != 224:
        raise ValueError("img_size must be 224")
    lib = _lib.load()
    if isinstance(frames, (list, tuple)):
        ts = [torch.as_tensor(np.ascontiguousarray(f) if isinstance(f, np.ndarray) else f) for f in frames]
        if not ts or any(t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != ts[0].shape[2] for t in ts):
            raise ValueError("frames must be uint8 [H,W,C] with one C")
        Cn = int(ts[0].shape[2])
        sizes = [(int(t.shape[0]), int(t.shape[1])) for t in ts]
        offs, total = [], 0
        for t in ts:
            offs.append(total)
            total += (t.numel() + 15) // 16 * 16
        dev = ts[0].device if ts[0].is_cuda else torch.device("cuda", torch.cuda.current_device())
        buf = torch.empty(total, dtype=torch.uint8, device=dev)
        for t, o in zip(ts, offs):
            buf[o:o + t.numel()].copy_(t.reshape(-1), non_blocking=True)
        B = len(ts)
        offsets = (C.c_longlong * B)(*offs)
        hw = (C.c_int * (2 * B))(*[v for s_ in sizes for v in s_])
        table = torch.empty(32 * B, dtype=torch.uint8, device=dev)
    else:
        t = torch.as_tensor(np.ascontiguousarray(frames) if isinstance(frames, np.ndarray) else frames)
        if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] not in (3, 4):
            raise ValueError("frames must be uint8 [B,H,W,3|4]")
        buf = (t if t.is_cuda else t.cuda()).contiguous()
        dev = buf.device
        B, Cn = int(t.shape[0]), int(t.shape[3])
        sizes = [(int(t.shape[1]), int(t.shape[2]))] * B
        offsets, table = None, None
        hw = (C.c_int * 2)(int(t.shape[1]), int(t.shape[2]))
    if Cn not in (3, 4):
        raise ValueError("frames must have 3 or 4 channels")
    if out is None:
        out = torch.empty((B, 224, 224, 3), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (B, 224, 224, 3) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 CUDA tensor [B,224,224,3]")
    pp = (C.c_int * (5 * B))()
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.hpe_preprocess_u8_batch(buf.data_ptr(), offsets, hw, B, Cn, out.data_ptr(), pp,
                                               None if table is None else table.data_ptr(), st))
    params = []
    for b, (H, W) in enumerate(sizes):
        mx = max(H, W)
        params.append({"scale": float(img_size) / mx if mx != img_size else 1.0, "start_pt": np.array([pp[5 * b], pp[5 * b + 1]]),
                       "end_pt": np.array([pp[5 * b + 2], pp[5 * b + 3]]), "img_size": pp[5 * b + 4]})
    return out, params


def get_original(proc_param, verts, cam, joints, img_size=224):
    """verts [P,3] or [B,P,3], cam [3] or [B,3] (CUDA float32), joints [K,2] / [B,K,2] (2-D keypoints in crop pixels;
    numpy or torch) -> (cam_for_render [3] numpy, vert_shifted CUDA tensor, kp_original numpy)."""
    import torch

    img_size = int(proc_param["img_size"])
    single = verts.dim() == 2
    v = (verts[None] if single else verts).contiguous().float()
    c = (cam.reshape(1, 3) if single else cam).contiguous().float()
    j = joints.detach().cpu().numpy() if isinstance(joints, torch.Tensor) else np.asarray(joints)
    j = np.ascontiguousarray((j[None] if single else j), dtype=np.float32)
    B, P, K = v.shape[0], v.shape[1], j.shape[1]
    out = torch.empty_like(v)
    cfr = (C.c_float * 3)()
    kp = np.empty((B, K, 2), np.float32)
    sp = (C.c_int * 2)(int(proc_param["start_pt"][0]), int(proc_param["start_pt"][1]))
    with torch.cuda.device(v.device):
        st = C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
        _lib.check(_lib.load().hpe_get_original(v.data_ptr(), c.data_ptr(), B, P, K, sp, float(proc_param["scale"]), img_size,
                                                out.data_ptr(), cfr, kp.ctypes.data_as(C.c_void_p), j.ctypes.data_as(C.c_void_p), st))
    return np.array(list(cfr), np.float32), (out[0] if single else out), (kp[0] if single else kp)
