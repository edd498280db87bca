from .resnet import (ResNet, resnet18, resnet34, resnet50, resnet101, resnet152, wide_resnet50_2,  # noqa: F401
                     wide_resnet101_2)
from .vision_transformer import (VisionTransformer, vit_small_patch16_224, vit_base_patch16_224,  # noqa: F401
                                 vit_base_patch16_384, vit_base_patch32_384, vit_large_patch16_224,
                                 vit_large_patch16_384, vit_large_patch32_384)
from .swin_transformer import (SwinTransformer, swintransformer_tiny_patch4_window7_224,  # noqa: F401
                               swintransformer_small_patch4_window7_224, swintransformer_base_patch4_window7_224,
                               swintransformer_large_patch4_window7_224, swintransformer_base_patch4_window12_384,
                               swintransformer_large_patch4_window12_384)
from .mobilenetv1 import MobileNetV1  # noqa: F401
from .mobilenetv2 import (MobileNetV2, mobilenet_v2, MobileNetV3Small, MobileNetV3Large, mobilenet_v3_small,  # noqa: F401
                          mobilenet_v3_large)
from .vgg import VGG, vgg11, vgg13, vgg16, vgg19  # noqa: F401
from .alexnet import AlexNet, alexnet  # noqa: F401
from .resnext import (ResNeXt, resnext50_32x4d, resnext50_64x4d, resnext101_32x4d, resnext101_64x4d,  # noqa: F401
                      resnext152_32x4d, resnext152_64x4d)
from .efficientnet import efficientnet, EfficientNet  # noqa: F401
from .resnest import resnest50_fast_1s1x64d, resnest50, resnest101, ResNeSt  # noqa: F401
