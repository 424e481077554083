"""Backbone registry (reference: models/__init__.py model_dict + model_def.py name mapping).  Only the
architectures the MoMA configs need are shipped; all obey `model(x, is_feat=True) -> (feats, logits)`."""
from .resnet_cifar import (resnet8, resnet14, resnet20, resnet32, resnet44, resnet56, resnet110, resnet8x4,
                           resnet32x4)
from .efficientnet import efficientnet_b0

model_dict = {
    "resnet8": resnet8, "resnet14": resnet14, "resnet20": resnet20, "resnet32": resnet32, "resnet44": resnet44,
    "resnet56": resnet56, "resnet110": resnet110, "resnet8x4": resnet8x4, "resnet32x4": resnet32x4,
    "effiB0": efficientnet_b0,
}
