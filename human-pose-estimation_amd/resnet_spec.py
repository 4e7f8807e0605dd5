"""ResNet-50 v1 conv-layer table for the encoder the reference instantiates with
``tensorflow.keras.applications.ResNet50(include_top=False, pooling='avg')``
(reference: src/models.py:35-41; topology per SURVEY.md §8(a) row 1).

This is the canonical layer ORDER of the C-ABI (``include/hpe.h``: ``hpe_load_conv`` takes the index
into this table).  Layer names are the Keras ones (``conv1``, ``res2a_branch2a`` … with the matching
``bn*`` BatchNorm), so a Keras-layout weight dict keyed by layer name maps 1:1.

Each entry: (name, bn_name, KH, KW, Cin, Cout, stride, Hin, Hout).
"""
from __future__ import annotations

from collections import namedtuple

ConvSpec = namedtuple("ConvSpec", "name bn_name kh kw cin cout stride hin hout")

STAGE_BLOCKS = {2: 3, 3: 4, 4: 6, 5: 3}
STAGE_FILTERS = {2: (64, 64, 256), 3: (128, 128, 512), 4: (256, 256, 1024), 5: (512, 512, 2048)}


def resnet50_conv_specs():
    specs = [ConvSpec("conv1", "bn_conv1", 7, 7, 3, 64, 2, 224, 112)]
    cin, h = 64, 56  # after 3x3/2 max-pool
    for stage in (2, 3, 4, 5):
        f1, f2, f3 = STAGE_FILTERS[stage]
        for b in range(STAGE_BLOCKS[stage]):
            blk = "abcdef"[b]
            first = b == 0
            s = 2 if (first and stage > 2) else 1
            hout = h // s
            base = "res%d%s_branch" % (stage, blk)
            bn = "bn%d%s_branch" % (stage, blk)
            specs.append(ConvSpec(base + "2a", bn + "2a", 1, 1, cin, f1, s, h, hout))
            specs.append(ConvSpec(base + "2b", bn + "2b", 3, 3, f1, f2, 1, hout, hout))
            specs.append(ConvSpec(base + "2c", bn + "2c", 1, 1, f2, f3, 1, hout, hout))
            if first:
                specs.append(ConvSpec(base + "1", bn + "1", 1, 1, cin, f3, s, h, hout))
            cin, h = f3, hout
    return specs


CONV_SPECS = resnet50_conv_specs()
CONV_INDEX = {s.name: i for i, s in enumerate(CONV_SPECS)}


def encoder_param_count():
    n = 0
    for s in CONV_SPECS:
        n += s.kh * s.kw * s.cin * s.cout + s.cout  # kernel + bias
        n += 4 * s.cout  # BN gamma, beta, moving_mean, moving_variance
    return n


def encoder_macs_per_image():
    return sum(s.kh * s.kw * s.cin * s.cout * s.hout * s.hout for s in CONV_SPECS)


def encoder_min_bytes_per_image(elem_bytes=4):
    """Algorithmic HBM bytes of the encoder per image when every activation tensor is written once and read once
    by each consumer (no on-chip fusion across layers): conv inputs actually touched (1x1/s2 convs read one pixel
    in four), conv outputs, residual reads of the 2c layers, the two pools; plus the fp32 input image and its padded
    copy.  Weights (read once per batch) are not included."""
    total = 224 * 224 * 3 * 4  # fp32 image read by the pad kernel
    total += 2 * 230 * 232 * 4 * elem_bytes  # padded copy written + read by conv1
    for s in CONV_SPECS:
        if s.name == "conv1":
            total += s.hout * s.hout * s.cout * elem_bytes  # conv1 out
            total += s.hout * s.hout * s.cout * elem_bytes + 56 * 56 * 64 * elem_bytes  # max-pool read + write
            continue
        total += s.hout * s.hout * s.cin * elem_bytes  # input pixels touched once (3x3 taps re-read from L2)
        total += s.hout * s.hout * s.cout * elem_bytes  # output
        if s.name.endswith("2c"):
            total += s.hout * s.hout * s.cout * elem_bytes  # residual
    total += 7 * 7 * 2048 * elem_bytes + 2048 * 4  # avg-pool
    return total
