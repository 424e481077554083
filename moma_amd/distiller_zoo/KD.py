"""Vanilla KD term of the MoMA step (reference: distiller_zoo/KD.py:7-17).  [B,n_cls] only -> stock torch."""
import torch.nn as nn
import torch.nn.functional as F


class DistillKL(nn.Module):
    """KL(log_softmax(y_s/T) || softmax(y_t/T)) * T^2, reduction batchmean."""

    def __init__(self, T):
        super().__init__()
        self.T = T

    def forward(self, y_s, y_t):
        log_p_s = F.log_softmax(y_s / self.T, dim=1)
        p_t = F.softmax(y_t / self.T, dim=1)
        return F.kl_div(log_p_s, p_t, reduction="batchmean") * (self.T ** 2)
