"""Name -> backbone constructor (reference: model_def.py:7-111, same call signature).

The reference hard-codes author-local checkpoint paths and `pretrained=True` downloads; neither exists
offline, so `pretrain` is honoured only when it is a path to a local checkpoint file, otherwise the model is
random-init with the current torch seed."""
import os

import torch

from .backbones import model_dict
from .helper.util import load_pretrained_weights


def load_model(model_name, pretrain, n_cls, strict=True, gpu=None, multiprocessing_distributed=False):
    if model_name not in model_dict:
        raise NotImplementedError("backbone not shipped: {} (have: {})".format(model_name, sorted(model_dict)))
    model = model_dict[model_name](num_classes=n_cls)
    if isinstance(pretrain, str) and os.path.isfile(pretrain):
        state = torch.load(pretrain, map_location="cpu")
        print("==> loading weights from", pretrain, load_pretrained_weights(model, state, strict))
    return model
