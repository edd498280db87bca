"""ObjectDetection task wrapper — same surface as tlxcv/tasks/object_detection.py:6-22."""
from typing import Any

from .. import engine as E, tlx


class ObjectDetection(tlx.nn.Module):
    def __init__(self, backbone: tlx.nn.Module) -> None:
        super().__init__()
        self.backbone = backbone

    def loss_fn(self, output: Any, target: Any) -> Any:
        raise NotImplementedError("training losses are out of scope for the inference engine (SURVEY.md §8f)")

    def forward(self, inputs: Any) -> Any:
        return self.backbone(E.to_model_device(inputs, self))

    def predict(self, inputs: Any, **kwargs) -> Any:
        self.set_eval()
        return self.backbone(E.to_model_device(inputs, self), **kwargs)
