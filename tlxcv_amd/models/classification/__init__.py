from .resnet import (ResNet, resnet18, resnet34, resnet50, resnet101, resnet152, wide_resnet50_2,  # noqa: F401
                     wide_resnet101_2)
