"""MI355X-native per-image forward hot path of maxpit/human-pose-estimation.

Host side mirrors the reference's interface for this path: ``Predictor`` (src/predictor.py), ``SMPL``
(src/tf_smpl/batch_smpl.py), ``batch_orth_proj_idrot`` / ``reproject_vertices`` (src/tf_smpl/projection.py),
``kp_reprojection_loss`` / ``mesh_reprojection_loss`` (src/ops.py).  All arithmetic runs in the C-ABI
library ``lib/libhpe_hip.so`` (include/hpe.h); there is no CPU fallback.
"""
import os as _os

# Batch-chunk streams + RCCL's own streams must not share hardware queues (DESIGN.md "Multi-GPU"); only effective when the
# package is imported before the first HIP call of the process, which is the normal order.
try:
    if int(_os.environ.get("GPU_MAX_HW_QUEUES", "0")) < 8:
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"
except ValueError:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

from . import resnet_spec, synthetic  # noqa: F401,E402
from ._lib import HpeError  # noqa: F401
from .engine import HpeEngine  # noqa: F401
from .image import get_original, preprocess_batch, preprocess_image  # noqa: F401
from .ops import kp_reprojection_loss, mesh_reprojection_loss  # noqa: F401
from .predictor import Predictor  # noqa: F401
from .projection import batch_orth_proj_idrot, reproject_vertices  # noqa: F401
from .smpl import SMPL  # noqa: F401
