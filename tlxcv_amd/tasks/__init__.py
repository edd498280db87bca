from .image_classification import ImageClassification  # noqa: F401
from .object_detection import ObjectDetection  # noqa: F401
