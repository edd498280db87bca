from .backbones.darknet import DarkNet, ConvBNLayer  # noqa: F401
from .yolov3 import YOLOv3, YOLOv3FPN, YOLOv3Head, YoloDetBlock  # noqa: F401
