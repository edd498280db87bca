from .backbones.darknet import DarkNet, ConvBNLayer  # noqa: F401
from .backbones.mobilenet_v1 import MobileNet  # noqa: F401
from .yolov3 import YOLOv3, YOLOv3FPN, YOLOv3Head, YoloDetBlock  # noqa: F401
from .detr import MultiHeadAttention  # noqa: F401
