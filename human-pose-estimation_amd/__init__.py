"""MI355X-native per-image forward hot path of maxpit/human-pose-estimation."""
from . import resnet_spec, synthetic  # noqa: F401
