"""Data-parallel wrap of the student (reference train_student_moma.py:345-349 wraps `model_s` in stock DistributedDataParallel).

Same semantics as the stock wrap -- replicas start from rank 0's parameters, gradients are averaged over the ranks before the
optimizer step, module buffers (BatchNorm running statistics and batch counters) are broadcast from rank 0 before every forward.
What is different is HOW, because the stock reducer's cost on this step is not communication (measured on a one-rank RCCL group,
EfficientNet-B0, rocprofv3 kernel trace of `MOMA_BENCH_FORCE_DIST=1 bench.py`: +365 launches and +1.2 ms of GPU time per step):
  * the reducer moves every gradient into its bucket with one fused scale-and-copy PER PARAMETER (214 launches: `zero_grad(
    set_to_none=True)`, the reference's call, hands it fresh gradient tensors every step; with a communication hook they become
    213 plain copies) -- to overlap an all-reduce of 16 MB that takes ~0.3 ms on xGMI with a 40 ms step;
  * `broadcast_buffers=True` copies each buffer back with one memcpy PER BUFFER (147 launches).
`FlatDataParallel` does neither: the buffers travel as one flat tensor per dtype, and the step's gradients -- student AND the
trainable criterion modules, which the reference leaves un-synchronised -- are reduced by ONE flat all-reduce behind
the backward (`FlatDataParallel.allreduce_grads`: a concatenation, the collective, a multi-tensor copy back): one collective
per step at a fixed point of the program, nothing issued from autograd hooks, nothing to order against a reducer -- which is
also what lets the student's forward + backward live in a HIP graph (helper/step_graph.py).

What the stock wrap checks and this one has to check itself:
  * construction: every rank must hold the same parameter / buffer list (count, shapes, dtypes) -- `_verify_replicas`, one
    all_gather of a fingerprint, RuntimeError on every rank on a mismatch (DDP: `_verify_param_shape_across_processes`);
  * first contact with a communicator: `collective_self_test` runs the two collectives the wrap uses (flat broadcast, flat
    sum all-reduce) on a known pattern and compares with the closed form.  `wrap_student(mode=None)` = "auto": flat when the
    self-test passes, the stock reducer (with a notice) when it does not;
  * every step: the set of gradients that goes into the flat buffer must be the same on every rank or the collective's sizes
    differ (a hang, or silent corruption).  The (count, element total, index hash) of the set is agreed on EVERY step, before
    the flat all-reduce is issued, over a host-side control group (`control_group`: gloo, CPU tensors -- no GPU work, no stream
    sync; ~0.1 ms of host time in a step whose host has tens of ms to spare): a MAX all-reduce of (sig, -sig), so max != min
    raises on every rank alike.  (Round 4 exchanged it only when it changed on the LOCAL rank: a set that changed on one rank
    alone sent that rank into an all_gather while the others entered the all-reduce -- ADVICE r4.)
`wrap_student(..., mode="ddp")` (`--dp ddp`, `MOMA_DP=ddp`) keeps the stock reducer with the flat buffer broadcast.
"""
from __future__ import annotations

import os
import zlib

import torch
import torch.distributed as dist
from torch import nn

from .. import ops


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


def _fingerprint(tensors) -> list:
    """(count, element total, crc of the shape / dtype list) of a tensor list -- what has to agree across the ranks"""
    desc = ";".join(f"{tuple(t.shape)}:{t.dtype}" for t in tensors)
    return [len(tensors), sum(t.numel() for t in tensors), zlib.crc32(desc.encode())]


def _all_agree(values, device, group=None, what="replicas"):
    """all_gather of a small int64 vector; RuntimeError ON EVERY RANK when the ranks do not hold the same one"""
    mine = torch.tensor(values, dtype=torch.int64, device=device)
    world = dist.get_world_size(group)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    rows = [g.tolist() for g in got]
    if any(r != rows[0] for r in rows):
        raise RuntimeError(f"data-parallel {what} differ across ranks (count, elements, layout hash per rank): {rows}")


_control_groups = {}       # (id of the data group, fresh?) -> (the data group object itself, its control group)


def control_group(group=None, force_new=False):
    """Host-side agreement channel next to the data group: the group itself when it runs on gloo, else a gloo group over the same
    ranks (made once per data group; COLLECTIVE: every rank of `group` must call it at the same point -- the wrap's constructor
    and wrap_student do).  Carries a few int64 on CPU tensors: verdicts and fingerprints, never data.
    The cache entry holds the data group OBJECT next to its control group: an id() cannot come back for another group while the
    object is alive, and an entry whose object is not the caller's group (a torn-down and re-initialised job) is dropped."""
    pg = group if group is not None else dist.group.WORLD
    key = (id(pg), bool(force_new))
    hit = _control_groups.get(key)
    if hit is None or hit[0] is not pg:
        if dist.get_backend(group) == "gloo" and not force_new:      # (force_new: tests walk the RCCL-side branch on a gloo job)
            ctl = pg
        else:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            ctl = dist.new_group(ranks=ranks, backend="gloo")
        _control_groups[key] = (pg, ctl)
    return _control_groups[key][1]


def agreed_control_group(device, group=None, force_new=False):
    """control_group() with ONE verdict for all ranks: creating the gloo group is a collective that may fail on one rank alone
    (no usable transport there); that rank would skip the per-step agreement while the others block in it.  Every rank reports
    whether it holds a group, the MIN over the DATA group (which exists) decides: the control group on every rank, or None on
    every rank (the caller then falls back to the device-side check)."""
    ctl, err = None, None
    try:
        ctl = control_group(group, force_new)
    except Exception as e:                                   # (reported below, by every rank that failed)
        err = e
    ok = torch.tensor([1 if ctl is not None else 0], dtype=torch.int64, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) != 1:
        if err is not None:
            print(f"[moma] data-parallel control group unavailable on rank {dist.get_rank()} ({type(err).__name__}: {err})")
        return None
    return ctl


def forget_control_groups():
    """drop the cache (called when a process group is torn down: the next job makes its own)"""
    _control_groups.clear()


def _host_agree(values, ctl, what):
    """every rank holds the same int64 vector, or RuntimeError on EVERY rank: MAX all-reduce of (v, -v) on the control group"""
    v = torch.tensor(list(values) + [-x for x in values], dtype=torch.int64)
    mine = v.clone()
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=ctl)
    n = len(values)
    if not torch.equal(v[:n], -v[n:]):                       # max != min somewhere: the same verdict on every rank
        raise RuntimeError(f"data-parallel {what} differ across ranks: this rank holds {mine[:n].tolist()}, the ranks' maxima are "
                           f"{v[:n].tolist()} and minima {(-v[n:]).tolist()} (count, elements, index hash)")


def broadcast_module_state(modules, group=None):
    """rank 0's parameters AND buffers of `modules` -> every rank, one flat broadcast per dtype.  The student gets this from its
    wrap; the trainable criterion modules (atts_q, embed_s) and the EMA teacher are not under any wrap -- in the reference they
    agree across ranks only because every rank seeds identically (train_student_moma.py:241-246)."""
    ts = []
    for m in modules:
        if m is not None:
            ts += list(m.parameters()) + list(m.buffers())
    if ts and dist.is_available() and dist.is_initialized():
        _all_agree(_fingerprint(ts), ts[0].device, group, "module states")
        _flat_broadcast(ts, group)


def collective_self_test(device, group=None) -> bool:
    """The wrap's two collectives on a known pattern, on the live communicator: a flat broadcast of rank 0's ramp and a flat sum
    all-reduce of (rank + 1) * ramp, compared with the closed form on every rank; the verdicts are AND-ed over the ranks (a MIN
    all-reduce), so every rank returns the same answer."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = 1 << 16
    ramp = torch.arange(n, device=device, dtype=torch.float32) % 251.0
    a = [torch.full((7,), float(rank + 1), device=device), (ramp * (rank + 1)).clone(), torch.full((3, 5), -float(rank), device=device)]
    _flat_broadcast(a, group)
    ok = bool(torch.equal(a[0], torch.full((7,), 1.0, device=device)) and torch.equal(a[1], ramp) and
              torch.equal(a[2], torch.zeros(3, 5, device=device)))
    flat = torch.cat([(ramp * (rank + 1)), torch.full((13,), float(rank + 1), device=device)])
    dist.all_reduce(flat, group=group)
    s = world * (world + 1) / 2.0
    ok = ok and bool(torch.equal(flat[:n], ramp * s) and torch.equal(flat[n:], torch.full((13,), s, device=device)))
    verdict = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
    return bool(verdict.item() == 1)


class FlatBufferBroadcast:
    """Forward pre-hook: rank 0's module buffers -> every rank (what DDP's `broadcast_buffers` does per forward).  The buffer
    list is taken at wrap time (as the stock reducer's is); `refresh()` re-reads it after a module surgery."""

    def __init__(self, module: nn.Module, group=None):
        self.group = group
        self.module = module
        self.refresh()

    def refresh(self):
        self.buffers = [b for b in self.module.buffers() if b is not None and b.numel() > 0]

    def __call__(self, _module=None, _args=None):
        if self.buffers:
            with ops._timed("dp_buffer_broadcast"):
                _flat_broadcast(self.buffers, self.group)


class FlatDataParallel(nn.Module):
    """`.module` + a forward that first broadcasts the buffers; the gradients are reduced by the training loop with ONE flat
    all-reduce per step (`flat_dp.allreduce_grads(flat_dp.grad_params() + criterion parameters)`)."""

    def __init__(self, module: nn.Module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        state = list(module.parameters()) + list(module.buffers())
        if state:
            _all_agree(_fingerprint(state), state[0].device, group, "student replicas")     # as DDP's constructor verifies
        _flat_broadcast(state, group)                                                        # replicas start from rank 0
        self.flat_buffer_broadcast = FlatBufferBroadcast(module, group)
        # (collective: all ranks construct the wrap.  MOMA_DP_CONTROL=new: a gloo group of its own even on a gloo job -- the branch
        #  an RCCL job takes --, for the CPU rehearsals)
        self._ctl, self._last_sig = None, None
        if dist.get_world_size(group) > 1:
            dev = state[0].device if state else torch.device("cpu")
            self._ctl = agreed_control_group(dev, group, os.environ.get("MOMA_DP_CONTROL") == "new")
            if self._ctl is None and dist.get_rank() == 0:       # (the same verdict on every rank)
                print("[moma] no host-side control group: gradient sets are verified on the device, at the first step and when "
                      "they change on this rank")
        self.allreduce_launches = 0

    def forward(self, *args, **kwargs):
        self.flat_buffer_broadcast()
        return self.module(*args, **kwargs)

    def grad_params(self):
        return [p for p in self.module.parameters() if p.requires_grad]

    def allreduce_grads(self, params, single_rank=False):
        """ONE flat all-reduce (average) over the gradients of `params` (the student's and the trainable criterion modules'),
        issued behind the backward.  -> number of collectives launched (0 without a group / at world size 1 unless
        single_rank: the one-rank rehearsal of the collective path)."""
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        world = dist.get_world_size(self.group)
        if world == 1 and not single_rank:
            return 0
        by_kind, which = {}, []
        for i, p in enumerate(params):
            if p.grad is not None:
                by_kind.setdefault((p.grad.dtype, p.grad.device), []).append(p.grad)
                which.append(i)
        if not which:
            sig = (0, 0, 0)
        else:
            sig = (len(which), sum(params[i].grad.numel() for i in which), zlib.crc32(repr(which).encode()))
        if self._ctl is not None:
            # agreed on every step, on the host, BEFORE the collective whose size depends on it
            _host_agree(sig, self._ctl, "gradient sets")
        elif world > 1 and sig != self._last_sig:
            # no host channel (agreed_control_group said so on every rank alike): the device-side exchange of round 4 -- at the first
            # step and whenever the set changes on this rank (a change on ANOTHER rank alone is not seen: the degraded mode)
            _all_agree(sig, params[0].device if params else torch.device("cpu"), self.group, "gradient sets")
        self._last_sig = sig
        with ops._timed("dp_allreduce_grads"):
            for grads in by_kind.values():                      # (one group in practice: every gradient here is fp32)
                flat = torch.cat([g.reshape(-1) for g in grads])
                dist.all_reduce(flat, group=self.group)
                flat.div_(world)
                torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
        self.allreduce_launches += len(by_kind)
        return len(by_kind)


def wrap_student(model: nn.Module, device_ids=None, group=None, mode: str | None = None) -> nn.Module:
    """mode: 'flat' | 'ddp' | None = MOMA_DP from the environment, else 'auto' (flat if the collectives pass their self-test
    on this communicator, else the stock reducer)."""
    mode = mode or os.environ.get("MOMA_DP", "auto")
    if mode == "auto":
        dev = next(model.parameters()).device
        ctl = agreed_control_group(dev, group) if dist.get_world_size(group) > 1 else None
        try:
            ok = collective_self_test(dev, group)
        except Exception as e:                                   # a backend that cannot run one of the two collectives
            print(f"[moma] flat data-parallel self-test raised {type(e).__name__}: {e}")
            ok = False
        if ctl is not None:
            # one verdict for all: a rank whose self-test RAISED (and never reached the test's own MIN all-reduce) must not end up
            # under another wrap than the rest -- AND-ed on the host-side channel, outside the try
            v = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=ctl)
            ok = bool(v.item() == 1)
        mode = "flat" if ok else "ddp"
        if not ok and dist.get_rank() == 0:
            print("[moma] flat data-parallel self-test FAILED on this communicator: falling back to the stock DDP reducer")
    if mode == "flat":
        return FlatDataParallel(model, group)
    if mode != "ddp":
        raise ValueError(f"unknown data-parallel mode {mode!r} (flat | ddp | auto)")
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=device_ids, gradient_as_bucket_view=True, broadcast_buffers=False,
                                              process_group=group)
    sync = FlatBufferBroadcast(model, group)
    if sync.buffers:
        ddp.register_forward_pre_hook(sync)
    ddp.flat_buffer_broadcast = sync
    return ddp
