from .darknet import DarkNet  # noqa: F401
from .mobilenet_v1 import MobileNet  # noqa: F401
