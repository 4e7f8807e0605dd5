"""CPU restatement (test infrastructure only) of the steps right before / after the hot path:

  preprocess_image  (reference: preview.py:18-35)  ->  scale_and_crop / resize_img (reference: src/util/image.py:7-39)
  get_original      (reference: src/util/renderer.py:260-283)

PARITY UNPINNED for ``resize_linear_u8``: the reference calls ``cv2.resize`` (OpenCV is not installed here), whose 8-bit
INTER_LINEAR path is restated from the published algorithm: half-pixel centres, 11-bit fixed point coefficients
(INTER_RESIZE_COEF_BITS), horizontal pass into ints, vertical pass
``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.  Everything else follows the reference lines cited.
"""
import numpy as np


def _axis(dsize, ssize):
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    lo = sx < 0
    fx[lo] = 0.0
    sx[lo] = 0
    hi = sx >= ssize - 1
    fx[hi] = 0.0
    sx[hi] = ssize - 1
    x1 = np.minimum(sx + 1, ssize - 1)
    a1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(2048.0)).astype(np.int64)
    return sx, x1, a0, a1


def resize_linear_u8(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h)) for uint8 images, INTER_LINEAR (the cv2 default)."""
    H, W = img.shape[:2]
    if (new_h, new_w) == (H, W):
        return img.copy()
    x0, x1, a0, a1 = _axis(new_w, W)
    y0, y1, b0, b1 = _axis(new_h, H)
    src = img.astype(np.int64)
    S0 = src[y0][:, x0] * a0[None, :, None] + src[y0][:, x1] * a1[None, :, None]
    S1 = src[y1][:, x0] * a0[None, :, None] + src[y1][:, x1] * a1[None, :, None]
    d = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(d, 0, 255).astype(np.uint8)


def resize_img(img, scale_factor):
    """reference: src/util/image.py:7-15"""
    new_size = (np.floor(np.array(img.shape[0:2]) * scale_factor)).astype(int)
    new_img = resize_linear_u8(img, int(new_size[0]), int(new_size[1]))
    actual_factor = [new_size[0] / float(img.shape[0]), new_size[1] / float(img.shape[1])]
    return new_img, actual_factor


def scale_and_crop(image, scale, center, img_size):
    """reference: src/util/image.py:17-39"""
    image_scaled, scale_factors = resize_img(image, scale)
    scale_factors = [scale_factors[1], scale_factors[0]]
    center_scaled = np.round(center * scale_factors).astype(int)
    margin = int(img_size / 2)
    image_pad = np.pad(image_scaled, ((margin,), (margin,), (0,)), mode="edge")
    center_pad = center_scaled + margin
    start_pt = center_pad - margin
    end_pt = center_pad + margin
    crop = image_pad[start_pt[1] : end_pt[1], start_pt[0] : end_pt[0], :]
    proc_param = {"scale": scale, "start_pt": start_pt, "end_pt": end_pt, "img_size": img_size}
    return crop, proc_param


def preprocess_image(img, img_size=224):
    """reference: preview.py:18-35"""
    if img.shape[2] == 4:
        img = img[:, :, :3]
    if np.max(img.shape[:2]) != img_size:
        scale = float(img_size) / np.max(img.shape[:2])
    else:
        scale = 1.0
    center = np.round(np.array(img.shape[:2]) / 2).astype(int)
    center = center[::-1]
    crop, proc_param = scale_and_crop(img, scale, center, img_size)
    crop = 2 * ((crop / 255.0) - 0.5)
    return crop, proc_param, img


def get_original(proc_param, verts, cam, joints, img_size):
    """reference: src/util/renderer.py:260-283 (single image: verts [P,3], cam [3], joints [K,2])"""
    img_size = proc_param["img_size"]
    undo_scale = 1.0 / np.array(proc_param["scale"])
    cam_s = cam[0]
    cam_pos = cam[1:]
    principal_pt = np.array([img_size, img_size]) / 2.0
    flength = 500.0
    tz = flength / (0.5 * img_size * cam_s)
    trans = np.hstack([cam_pos, tz])
    vert_shifted = verts + trans
    start_pt = proc_param["start_pt"] - 0.5 * img_size
    final_principal_pt = (principal_pt + start_pt) * undo_scale
    cam_for_render = np.hstack([np.mean(flength * undo_scale), final_principal_pt])
    margin = int(img_size / 2)
    kp_original = (joints + proc_param["start_pt"] - margin) * undo_scale
    return cam_for_render, vert_shifted, kp_original
