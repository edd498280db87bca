from .image_classification import ImageClassification  # noqa: F401
