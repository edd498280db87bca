"""ImageClassification task wrapper — same surface as tlxcv/tasks/image_classification.py:6-23."""
from typing import Any

from .. import engine as E, tlx


class ImageClassification(tlx.nn.Module):
    def __init__(self, backbone: tlx.nn.Module) -> None:
        super().__init__()
        self.backbone = backbone

    def loss_fn(self, output: Any, target: Any) -> Any:
        raise NotImplementedError("training losses are out of scope for the inference engine (SURVEY.md §8f)")

    def forward(self, inputs: Any) -> Any:
        return self.backbone(E.to_model_device(inputs, self))

    def predict(self, inputs: Any) -> Any:
        self.set_eval()
        outputs = self.backbone(E.to_model_device(inputs, self))
        return tlx.argmax(outputs, axis=-1)
