"""MI355X-native per-image forward hot path of maxpit/human-pose-estimation.

Host side mirrors the reference's interface for this path: ``Predictor`` (src/predictor.py), ``SMPL``
(src/tf_smpl/batch_smpl.py), ``batch_orth_proj_idrot`` / ``reproject_vertices`` (src/tf_smpl/projection.py),
``kp_reprojection_loss`` / ``mesh_reprojection_loss`` (src/ops.py).  All arithmetic runs in the C-ABI
library ``lib/libhpe_hip.so`` (include/hpe.h); there is no CPU fallback.
"""
from . import resnet_spec, synthetic  # noqa: F401
from ._lib import HpeError  # noqa: F401
from .engine import HpeEngine  # noqa: F401
from .image import get_original, preprocess_image  # noqa: F401
from .ops import kp_reprojection_loss, mesh_reprojection_loss  # noqa: F401
from .predictor import Predictor  # noqa: F401
from .projection import batch_orth_proj_idrot, reproject_vertices  # noqa: F401
from .smpl import SMPL  # noqa: F401
