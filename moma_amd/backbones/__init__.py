"""Backbone registry (reference: models/__init__.py model_dict + model_def.py name mapping).  Only the
architectures the MoMA configs need are shipped; all obey `model(x, is_feat=True) -> (feats, logits)`."""
from .resnet_cifar import (resnet8, resnet14, resnet20, resnet32, resnet44, resnet56, resnet110, resnet8x4,
                           resnet32x4)
from .efficientnet import efficientnet_b0
from .resnet_imagenet import ResNet18, ResNet34, ResNet50, resnet101
from .vit import (vit_tiny_patch16_224, vit_small_patch16_224, vit_base_patch16_224, vit_tiny_patch16_384,
                  vit_base_patch16_384)

model_dict = {
    "resnet8": resnet8, "resnet14": resnet14, "resnet20": resnet20, "resnet32": resnet32, "resnet44": resnet44,
    "resnet56": resnet56, "resnet110": resnet110, "resnet8x4": resnet8x4, "resnet32x4": resnet32x4,
    "effiB0": efficientnet_b0,
    # ImageNet-style ResNets (reference names, model_def.py:59-64)
    "ResNet18": ResNet18, "ResNet34": ResNet34, "ResNet50": ResNet50, "resnet101": resnet101,
    # ViTs (reference names model_def.py:80-109; deit_* are the same architectures; + ViT-S for BASELINE config 3)
    "vit_tiny_patch16_224": vit_tiny_patch16_224, "deit_tiny_patch16_224": vit_tiny_patch16_224,
    "vit_small_patch16_224": vit_small_patch16_224,
    "vit_base_patch16_224": vit_base_patch16_224, "deit_base_patch16_224": vit_base_patch16_224,
    "vit_tiny_patch16_384": vit_tiny_patch16_384,
    "vit_base_patch16_384": vit_base_patch16_384, "deit_base_patch16_384": vit_base_patch16_384,
}
