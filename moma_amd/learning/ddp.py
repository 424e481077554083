"""Data-parallel wrap of the student (reference train_student_moma.py:345-349 wraps `model_s` in stock DistributedDataParallel).

Same semantics as the stock wrap -- replicas start from rank 0's parameters, gradients are averaged over the ranks before the
optimizer step, module buffers (BatchNorm running statistics and batch counters) are broadcast from rank 0 before every forward.
What is different is HOW, because the stock reducer's cost on this step is not communication (measured on a one-rank RCCL group,
EfficientNet-B0, rocprofv3 kernel trace of `MOMA_BENCH_FORCE_DIST=1 bench.py`: +365 launches and +1.2 ms of GPU time per step):
  * the reducer moves every gradient into its bucket with one fused scale-and-copy PER PARAMETER (214 launches: `zero_grad(
    set_to_none=True)`, the reference's call, hands it fresh gradient tensors every step; with a communication hook they become
    213 plain copies) -- to overlap an all-reduce of 16 MB that takes ~0.3 ms on xGMI with a 40 ms step;
  * `broadcast_buffers=True` copies each buffer back with one memcpy PER BUFFER (147 launches).
`FlatDataParallel` (the default) does neither: the buffers travel as one flat tensor per dtype, and the step's gradients --
student AND the trainable criterion modules, which the reference leaves un-synchronised -- are reduced by ONE flat
all-reduce behind the backward (`ContrastTrainer.allreduce_grads`: a concatenation, the collective, a multi-tensor copy back):
one collective per step at a fixed point of the program, nothing issued from autograd hooks, nothing to order against a
reducer.  `wrap_student(..., mode="ddp")` (`--dp ddp`, `MOMA_DP=ddp`) keeps the stock reducer with the flat buffer broadcast.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import nn


def _flat_broadcast(tensors, group=None):
    """rank 0's values -> every rank, one broadcast per (dtype, device), scattered back by a multi-tensor copy"""
    by_kind = {}
    for t in tensors:
        if t is not None and t.numel() > 0:
            by_kind.setdefault((t.dtype, t.device), []).append(t)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    with torch.no_grad():
        for ts in by_kind.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            torch._foreach_copy_(ts, [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in ts]), ts)])


class FlatBufferBroadcast:
    """Forward pre-hook: rank 0's module buffers -> every rank (what DDP's `broadcast_buffers` does per forward)."""

    def __init__(self, module: nn.Module, group=None):
        self.group = group
        self.buffers = [b for b in module.buffers() if b is not None and b.numel() > 0]

    def __call__(self, _module=None, _args=None):
        if self.buffers:
            _flat_broadcast(self.buffers, self.group)


class FlatDataParallel(nn.Module):
    """`.module` + a forward that first broadcasts the buffers; the gradients are reduced by the training loop with ONE flat
    all-reduce per step (`ContrastTrainer.allreduce_grads(flat_dp.grad_params() + criterion parameters)`)."""

    def __init__(self, module: nn.Module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        _flat_broadcast(list(module.parameters()) + list(module.buffers()), group)      # replicas start from rank 0 (as DDP's constructor)
        self.flat_buffer_broadcast = FlatBufferBroadcast(module, group)

    def forward(self, *args, **kwargs):
        self.flat_buffer_broadcast()
        return self.module(*args, **kwargs)

    def grad_params(self):
        return [p for p in self.module.parameters() if p.requires_grad]


def wrap_student(model: nn.Module, device_ids=None, group=None, mode: str | None = None) -> nn.Module:
    mode = mode or os.environ.get("MOMA_DP", "flat")
    if mode == "flat":
        return FlatDataParallel(model, group)
    if mode != "ddp":
        raise ValueError(f"unknown data-parallel mode {mode!r} (flat | ddp)")
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=device_ids, gradient_as_bucket_view=True, broadcast_buffers=False,
                                              process_group=group)
    sync = FlatBufferBroadcast(model, group)
    if sync.buffers:
        ddp.register_forward_pre_hook(sync)
    ddp.flat_buffer_broadcast = sync
    return ddp
