"""Meters / top-k accuracy used by the contrast trainer (reference: learning/util.py:7-41)."""
import torch


class AverageMeter(object):
    """Running mean; `val` may be a Python number or a 0-d tensor (kept on device, no host sync)."""

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
    """Top-k accuracy in percent, one 1-element tensor per k (learning/util.py:25-41)."""
    with torch.no_grad():
        maxk = max(topk)
        bsz = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        hit = pred.t().eq(target.view(1, -1).expand(maxk, bsz))
        return [hit[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / bsz) for k in topk]
