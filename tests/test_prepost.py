"""Steps right before / after the path: preprocess_image / get_original.
cv2 is absent, so the bilinear resize is 'parity unpinned' (checked against a restatement of OpenCV's published 8-bit
algorithm); index arithmetic, edge padding, crop and normalisation follow the reference lines."""
import numpy as np
import pytest

from oracle import prepost_oracle as P


def _img(h, w, c=3, seed=0):
    g = np.random.Generator(np.random.Philox(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    base = (yy[..., None] * 3 + xx[..., None] * 5 + np.arange(c) * 40) % 256
    return ((base + g.integers(0, 30, (h, w, c))) % 256).astype(np.uint8)


def test_oracle_resize_kats():
    img = _img(224, 224)
    crop, pp, _ = P.preprocess_image(img)
    np.testing.assert_allclose(crop, 2 * (img / 255.0 - 0.5))  # already 224: identity resize, centred crop
    assert pp["scale"] == 1.0 and tuple(pp["start_pt"]) == (112, 112)  # padded-image coordinates
    flat = np.full((300, 500, 3), 77, np.uint8)
    c2, pp2, _ = P.preprocess_image(flat)
    assert c2.shape == (224, 224, 3) and np.allclose(c2, 2 * (77 / 255.0 - 0.5))  # constant image stays constant
    up = P.resize_linear_u8(np.array([[[0], [255]]], np.uint8), 1, 4)
    assert up[0, 0, 0] == 0 and up[0, 3, 0] == 255 and up[0, 1, 0] < up[0, 2, 0]


def test_oracle_get_original_closed_form():
    pp = {"scale": 0.5, "start_pt": np.array([10, 20]), "end_pt": np.array([234, 244]), "img_size": 224}
    verts = np.zeros((4, 3))
    cam = np.array([0.8, 0.1, -0.2])
    joints = np.array([[112.0, 112.0]])
    cfr, vs, kp = P.get_original(pp, verts, cam, joints, 224)
    np.testing.assert_allclose(vs[0], [0.1, -0.2, 500.0 / (0.5 * 224 * 0.8)])
    np.testing.assert_allclose(cfr, [1000.0, (112 + 10 - 112) * 2.0, (112 + 20 - 112) * 2.0])
    np.testing.assert_allclose(kp[0], [(112 + 10 - 112) * 2.0, (112 + 20 - 112) * 2.0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(224, 224, 3), (480, 640, 3), (640, 480, 4), (100, 37, 3), (1080, 1920, 3), (224, 100, 3)])
def test_preprocess_matches_oracle(shape):
    import hpe_amd

    img = _img(*shape, seed=shape[0])
    ref, pp_ref, _ = P.preprocess_image(img)
    out, pp, _ = hpe_amd.preprocess_image(img)
    assert tuple(out.shape) == (224, 224, 3)
    assert tuple(pp["start_pt"]) == tuple(pp_ref["start_pt"]) and tuple(pp["end_pt"]) == tuple(pp_ref["end_pt"])
    assert abs(pp["scale"] - pp_ref["scale"]) < 1e-12
    np.testing.assert_allclose(out.cpu().numpy(), ref.astype(np.float32), rtol=0, atol=1e-6)  # same uint8 -> same float


@pytest.mark.gpu
def test_preprocess_batch_matches_oracle():
    """hpe_preprocess_u8_batch: one launch for a batch -- frames of different sizes through the per-image table, and a stream
    of equal frames (no table) -- bit-equal to the oracle's per-image preprocess_image and to the single-frame kernel."""
    import hpe_amd

    shapes = [(224, 224, 3), (480, 640, 3), (100, 37, 3), (300, 500, 3), (224, 100, 3), (721, 333, 3)]
    imgs = [_img(*sh, seed=7 + i) for i, sh in enumerate(shapes)]
    out, params = hpe_amd.preprocess_batch(imgs)
    assert tuple(out.shape) == (len(imgs), 224, 224, 3)
    for i, im in enumerate(imgs):
        ref, pp_ref, _ = P.preprocess_image(im)
        assert tuple(params[i]["start_pt"]) == tuple(pp_ref["start_pt"]) and tuple(params[i]["end_pt"]) == tuple(pp_ref["end_pt"])
        assert abs(params[i]["scale"] - pp_ref["scale"]) < 1e-12
        np.testing.assert_allclose(out[i].cpu().numpy(), ref.astype(np.float32), rtol=0, atol=1e-6)
        one, _, _ = hpe_amd.preprocess_image(im)
        np.testing.assert_array_equal(out[i].cpu().numpy(), one.cpu().numpy())
    for sh in ((224, 224, 3), (360, 480, 4)):
        vid = np.stack([_img(*sh, seed=40 + k) for k in range(5)])
        out, params = hpe_amd.preprocess_batch(vid)
        for k in range(5):
            ref, pp_ref, _ = P.preprocess_image(vid[k])
            assert tuple(params[k]["start_pt"]) == tuple(pp_ref["start_pt"])
            np.testing.assert_allclose(out[k].cpu().numpy(), ref.astype(np.float32), rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_get_original_matches_oracle():
    import torch

    import hpe_amd

    g = np.random.Generator(np.random.Philox(3))
    pp = {"scale": 224.0 / 640.0, "start_pt": np.array([0, -28]), "end_pt": np.array([224, 196]), "img_size": 224}
    verts = g.normal(0, 1, (6890, 3)).astype(np.float32)
    cam = np.array([0.9, 0.05, -0.1], np.float32)
    joints = g.uniform(0, 224, (19, 2)).astype(np.float32)
    cfr, vs, kp = hpe_amd.get_original(pp, torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda(), joints)
    rc, rv, rk = P.get_original(pp, verts.astype(np.float64), cam.astype(np.float64), joints.astype(np.float64), 224)
    np.testing.assert_allclose(cfr, rc, rtol=1e-6)
    np.testing.assert_allclose(vs.cpu().numpy(), rv, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(kp, rk, rtol=1e-6)
