"""DistributedDataParallel wrap of the student (reference train_student_moma.py:345-349 wraps `model_s` in stock DDP).

Same semantics as the stock wrap -- bucketed gradient all-reduce averaged over the ranks, module buffers (BatchNorm running
statistics and batch counters) broadcast from rank 0 before every forward.  Measured on a one-rank RCCL group (EfficientNet-B0,
rocprofv3 kernel trace of `MOMA_BENCH_FORCE_DIST=1 bench.py`): stock DDP adds 365 launches and 1.2 ms of GPU time per step --
one fused scale-and-copy of every gradient into its bucket (214 launches: the reducer's design, `zero_grad(set_to_none=True)`
hands it fresh gradient tensors every step; a communication hook only turns them into 213 plain copies, so none is used) and one
memcpy PER BUFFER behind `broadcast_buffers=True` (147 launches).  The second storm is removed here: the buffers are broadcast as
one flat tensor per dtype and scattered back by a multi-tensor copy.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn


class FlatBufferBroadcast:
    """Forward pre-hook: rank 0's module buffers -> every rank, one broadcast per dtype (what DDP's `broadcast_buffers` does per
    forward, reference wrap: stock default)."""

    def __init__(self, module: nn.Module, group=None):
        self.group = group
        by_dtype = {}
        for b in module.buffers():
            if b is not None and b.numel() > 0:
                by_dtype.setdefault((b.dtype, b.device), []).append(b)
        self.groups = list(by_dtype.values())

    def __call__(self, _module=None, _args=None):
        with torch.no_grad():
            for bufs in self.groups:
                flat = torch.cat([b.reshape(-1) for b in bufs])
                dist.broadcast(flat, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                torch._foreach_copy_(bufs, [v.view_as(b) for v, b in zip(flat.split([b.numel() for b in bufs]), bufs)])


def wrap_student(model: nn.Module, device_ids=None, group=None) -> nn.parallel.DistributedDataParallel:
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=device_ids, gradient_as_bucket_view=True, broadcast_buffers=False,
                                              process_group=group)
    sync = FlatBufferBroadcast(model, group)
    if sync.groups:
        ddp.register_forward_pre_hook(sync)
    ddp.flat_buffer_broadcast = sync
    return ddp
