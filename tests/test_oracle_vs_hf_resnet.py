"""Independent cross-check of the oracle's ResNet-50 v1 restatement.

The reference takes its encoder from `tensorflow.keras.applications.ResNet50` (src/models.py:39), which is not in the repo
and not installable here.  What IS installed is Hugging Face transformers' PyTorch `ResNetModel`, an unrelated third-party
implementation of the same published architecture; with `downsample_in_bottleneck=True` it puts the stride-2 on the first
1x1 of a stage's first block and on the projection shortcut, i.e. ResNet v1 as Keras builds it (torchvision's v1.5 puts it on
the 3x3).  Loading the oracle's Keras-layout weights into it (conv bias folded into the BN running mean, BN eps 1e-3) must
reproduce the oracle's features.  This pins the TOPOLOGY of the restatement against independent code; it is still not the
reference's own TensorFlow run, so the oracle stays 'parity unpinned' in the strict sense."""
import numpy as np
import pytest
import torch

from hpe_amd import resnet_spec, synthetic
from oracle import hmr_oracle as O

transformers = pytest.importorskip("transformers")


def _load(hf_layer, p, conv, bn, eps):
    w = torch.from_numpy(np.ascontiguousarray(p[conv + "/kernel"])).permute(3, 2, 0, 1).contiguous()
    hf_layer.convolution.weight.data.copy_(w)
    n = hf_layer.normalization
    n.eps = eps
    n.weight.data.copy_(torch.from_numpy(p[bn + "/gamma"]))
    n.bias.data.copy_(torch.from_numpy(p[bn + "/beta"]))
    n.running_mean.data.copy_(torch.from_numpy(p[bn + "/moving_mean"] - p[conv + "/bias"]))  # BN(conv + b) == BN'(conv)
    n.running_var.data.copy_(torch.from_numpy(p[bn + "/moving_variance"]))


@pytest.mark.parametrize("trivial_bn", [False, True])
def test_oracle_encoder_matches_hf_resnet_v1(trivial_bn):
    from transformers import ResNetConfig, ResNetModel

    eps = 1e-3
    p = synthetic.make_encoder_params(seed=11, trivial_bn=trivial_bn)
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=True)
    m = ResNetModel(cfg).eval().double()
    pd = {k: v.astype(np.float64) for k, v in p.items()}
    _load(m.embedder.embedder, pd, "conv1", "bn_conv1", eps)
    for si, stage in enumerate(m.encoder.stages):
        for bi, blk in enumerate(stage.layers):
            base = "res%d%s_branch" % (si + 2, "abcdef"[bi])
            bn = "bn%d%s_branch" % (si + 2, "abcdef"[bi])
            for li, suffix in enumerate(("2a", "2b", "2c")):
                _load(blk.layer[li], pd, base + suffix, bn + suffix, eps)
            if bi == 0:
                _load(blk.shortcut, pd, base + "1", bn + "1", eps)
            else:
                assert isinstance(blk.shortcut, torch.nn.Identity)
    n_loaded = sum(1 for s in resnet_spec.CONV_SPECS)
    assert n_loaded == 53 and sum(q.numel() for q in m.parameters() if q.dim() == 4) == sum(
        s.kh * s.kw * s.cin * s.cout for s in resnet_spec.CONV_SPECS)
    img = synthetic.make_images(1, seed=12)
    with torch.no_grad():
        hf = m(torch.from_numpy(img).double().permute(0, 3, 1, 2)).pooler_output[:, :, 0, 0].numpy()
    ours = O.resnet50_features(img, p, eps=eps, dtype=np.float64)
    err = np.abs(hf - ours).max() / np.abs(ours).max()
    assert err < 1e-9, err
    ours32 = O.resnet50_features(img, p, eps=eps, dtype=np.float32)
    assert np.abs(ours32 - hf).max() / np.abs(hf).max() < 1e-5
