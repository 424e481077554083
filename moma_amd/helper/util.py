"""Meters, accuracy, LR schedule, metric all-reduce and json logging for the MoMA trainer
(reference: helper/util.py:37-139)."""
from __future__ import print_function

import json
import math
import os

import numpy as np
import torch
import torch.distributed as dist


def adjust_learning_rate(epoch, opt, optimizer):
    """The epoch schedule of the reference trainer (helper/util.py:37-50, called at train_student_moma.py:484):
    --cosine -> cosine from lr down to eta_min = lr * rate^3 over opt.epochs, otherwise step decay
    lr * rate^(#decay epochs passed); the lr of EVERY param group is rewritten each epoch.  Returns the lr."""
    lr = opt.learning_rate
    if getattr(opt, "cosine", False):
        eta_min = lr * (opt.lr_decay_rate ** 3)
        lr = eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / opt.epochs)) / 2
    else:
        steps = np.sum(epoch > np.asarray(opt.lr_decay_epochs))
        if steps > 0:
            lr = lr * (opt.lr_decay_rate ** steps)
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


class AverageMeter(object):
    """Running mean.  Accepts 0-d device tensors so the hot loop never forces a host sync; `.avg` is then a
    tensor and is converted with float() where it is printed / returned."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(output, target, topk=(1,)):
    """Top-k accuracy in percent (helper/util.py:71-85)."""
    with torch.no_grad():
        maxk = max(topk)
        bsz = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        hit = pred.t().eq(target.view(1, -1).expand(maxk, bsz))
        return [hit[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / bsz) for k in topk]


def reduce_tensor(tensor, world_size=1, op="avg"):
    """all_reduce(sum) of a metric tensor, divided by world size for 'avg' (helper/util.py:134-139)."""
    rt = tensor.clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    if world_size > 1 and op == "avg":
        rt /= world_size
    return rt


def save_dict_to_json(d, json_path):
    with open(json_path, "w") as f:
        json.dump(d, f, indent=4, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))


def update_dict_to_json(epoch, d, json_path):
    """Add `d` under key `epoch` to the json file (created on first use) -- reference helper/util.py:87-107."""
    data = {}
    if os.path.isfile(json_path):
        with open(json_path) as f:
            data = json.load(f)
    data[str(epoch)] = d
    save_dict_to_json(data, json_path)


def load_json_to_dict(json_path):
    with open(json_path, "r") as f:
        return json.load(f)


def load_pretrained_weights(model, state, strict_flag=True):
    """Load a checkpoint the way the reference does (helper/util.py:141-162): strip the DDP `module.` prefix;
    with strict_flag False (CLI --std_strict / --tec_strict, store_false) drop the classifier so a different
    n_cls can be fine-tuned."""
    if "model" in state:
        state = state["model"]
    clean = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
    if not strict_flag:
        for k in ("classifier_.1.weight", "classifier_.1.bias", "fc.weight", "fc.bias", "head.weight", "head.bias"):
            clean.pop(k, None)
    return model.load_state_dict(clean, strict=strict_flag)
