"""`tensorlayerx.vision` subset used by the inference demos (demo/image_classification/predict*.py:2-3,21-29)."""
from . import transforms  # noqa: F401
from .transforms.utils import load_image  # noqa: F401
