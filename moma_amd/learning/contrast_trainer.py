"""Step services of the MoMA loop (reference: learning/contrast_trainer.py).

  momentum_update  (:207-211)  per-tensor mul_/add_ pairs        -> ONE multi-tensor HIP launch (K4)
  _shuffle_bn      (:90-133)   Shuffle-BN key encoding.  Default `per_rank` mode keeps everything on the
                               rank (north star: per-rank queue, no cross-rank gather): a local permutation,
                               teacher forward, un-permute; `all_k` is the shuffled-order keys exactly as the
                               reference's gather-before-unshuffle produces at world size 1.  `gather` mode
                               reproduces the reference's collectives C3-C5 (image all_gather, id broadcast,
                               key all_gather) for exact W>1 parity.
  _compute_loss_accuracy (:189-205), broadcast_memory (:71-81), _global_gather (:83-88).
"""
from __future__ import print_function

import collections

import torch
import torch.distributed as dist

from .. import ops
from .base_trainer import BaseTrainer
from .util import accuracy


class ContrastTrainer(BaseTrainer):
    """trainer for contrastive distillation"""

    # device pointer tables of the (model, model_ema) pairs seen, newest last; bounded (a trainer touches two pairs:
    # backbone and mlp head) so that short-lived models -- tests, sweeps -- do not pile up tables keyed by dead ids
    _ema_tables = collections.OrderedDict()
    _EMA_TABLES_MAX = 8

    def __init__(self, args):
        super().__init__(args)

    # -- K4 ---------------------------------------------------------------------------------------
    @staticmethod
    def momentum_update(model, model_ema, m):
        """model_ema = m * model_ema + (1 - m) * model, parameters only (BN buffers are not averaged).

        Like the reference's zip over parameters() this requires identical architectures; a shape mismatch
        raises RuntimeError -- before touching the teacher, unlike the reference which fails half-way (Q4)."""
        ps = list(model.parameters())          # (data_ptr / numel are read straight off the parameters: a detach() per tensor
        es = list(model_ema.parameters())      #  and call doubles the host time of this call -- ~0.5 ms for EfficientNet-B0)
        key = (id(model), id(model_ema))
        tabs = ContrastTrainer._ema_tables
        tab = tabs.get(key)
        if tab is None or not tab.matches(ps, es):       # (matches() also catches an id reused by a new model)
            tab = ops.EmaTable(ps, es)
            # the kernel writes the EMA weights through raw pointers (no autograd version bump): attention modules among them
            # keep bf16 weight packs keyed on those versions and have to be told
            tab.packed = [mod for mod in model_ema.modules() if hasattr(mod, "invalidate_pack")]
            tabs[key] = tab
            while len(tabs) > ContrastTrainer._EMA_TABLES_MAX:
                tabs.popitem(last=False)
        else:
            tabs.move_to_end(key)
        ops.ema_update_(tab, m)
        for mod in getattr(tab, "packed", ()):
            mod.invalidate_pack()

    # -- collectives ------------------------------------------------------------------------------
    def broadcast_memory(self, contrast):
        """Synchronize memory buffers (C2).  Kept for identical initial queues across ranks."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        if getattr(self.args, "mem", "MoCo") in ["MoCo", "MoCoAtt"]:
            dist.broadcast(contrast.memory, 0)
        else:
            dist.broadcast(contrast.memory_s, 0)
            dist.broadcast(contrast.memory_t, 0)

    @staticmethod
    def _global_gather(x):
        all_x = [torch.ones_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(all_x, x, async_op=False)
        return torch.cat(all_x, dim=0)

    def _shuffle_bn(self, x, model_ema, model_ema_head):
        """-> (k [B,d], all_k [n,d]).  per_rank: n = B, no communication; gather: n = B*W (reference)."""
        mode = getattr(self.args, "shuffle_bn", "per_rank")
        if mode == "gather" and dist.is_available() and dist.is_initialized():
            return self._shuffle_bn_gather(x, model_ema, model_ema_head)
        bsz = x.size(0)
        shuffle_ids = self._host_randperm(bsz, x.device)        # host RNG stream, as the reference (:108)
        reverse_ids = torch.argsort(shuffle_ids)
        with torch.no_grad():
            feat_t, _ = model_ema(x[shuffle_ids], is_feat=True)
            all_k = model_ema_head(feat_t[-1])
        k = all_k[reverse_ids]
        return k, all_k

    _perm_feed = None      # set by helper/step_graph.py while a step is captured: the permutation's static device tensor

    def _host_randperm(self, n, device):
        """torch.randperm from the HOST generator (the reference's stream: seeded runs draw identical permutations), moved
        through pinned memory without blocking -- a pageable H2D copy is a host-device sync every step.  While a step is being
        captured into a HIP graph the permutation is the graph's static input (drawn from the same generator, one per step,
        and copied in before every replay: step_graph.PermFeed)."""
        if self._perm_feed is not None:
            return self._perm_feed.take(n, device)
        if device.type != "cuda":
            return torch.randperm(n).to(device)
        return torch.randperm(n, pin_memory=True).to(device, non_blocking=True)

    def _shuffle_bn_gather(self, x, model_ema, model_ema_head):
        args = self.args
        gp = self.local_group
        bsz = x.size(0)
        node_x = [torch.ones_like(x) for _ in range(dist.get_world_size(gp))]
        dist.all_gather(node_x, x.contiguous(), group=gp, async_op=False)
        node_x = torch.cat(node_x, dim=0)
        shuffle_ids = self._host_randperm(bsz * dist.get_world_size(gp), x.device)
        reverse_ids = torch.argsort(shuffle_ids)
        dist.broadcast(shuffle_ids, 0)
        dist.broadcast(reverse_ids, 0)
        this_ids = shuffle_ids[args.local_rank * bsz:(args.local_rank + 1) * bsz]
        with torch.no_grad():
            feat_t, _ = model_ema(node_x[this_ids], is_feat=True)
            k = model_ema_head(feat_t[-1])
        all_k = self._global_gather(k)
        node_id, ngpus = args.node_rank, args.ngpus_per_node
        node_k = all_k[node_id * ngpus * bsz:(node_id + 1) * ngpus * bsz]
        k = node_k[reverse_ids[args.local_rank * bsz:(args.local_rank + 1) * bsz]]
        return k, all_k

    def _shuffle_bn_attn(self, x, model_ema, model_ema_head, criterion_kd, q):
        """Shuffle-BN key encoding with the attention applied BEFORE the un-shuffle (reference :135-187):
        attn == 'self_mix' runs one module over [q ; k], otherwise atts_q(q) / atts_k(k).  -> (q, k, all_k).
        Per-rank (no collectives) unless shuffle_bn == 'gather'."""
        args = self.args
        bsz = x.size(0)
        gather = getattr(args, "shuffle_bn", "per_rank") == "gather" and dist.is_available() and dist.is_initialized()
        if gather:
            gp = self.local_group
            node_x = [torch.ones_like(x) for _ in range(dist.get_world_size(gp))]
            dist.all_gather(node_x, x.contiguous(), group=gp, async_op=False)
            node_x = torch.cat(node_x, dim=0)
            shuffle_ids = self._host_randperm(bsz * dist.get_world_size(gp), x.device)
            reverse_ids = torch.argsort(shuffle_ids)
            dist.broadcast(shuffle_ids, 0)
            dist.broadcast(reverse_ids, 0)
            lo = args.local_rank * bsz
        else:
            node_x = x
            shuffle_ids = self._host_randperm(bsz, x.device)
            reverse_ids = torch.argsort(shuffle_ids)
            lo = 0
        with torch.no_grad():
            feat_t, _ = model_ema(node_x[shuffle_ids[lo:lo + bsz]], is_feat=True)
            k = model_ema_head(feat_t[-1])
        if args.attn == "self_mix":
            out = criterion_kd.atts(torch.cat([q, k], dim=0))
            q, k = out[:bsz], out[bsz:]
        else:
            q = criterion_kd.atts_q(q)
            k = criterion_kd.atts_k(k)
        if gather:
            all_k = self._global_gather(k)
            node_id, ngpus = args.node_rank, args.ngpus_per_node
            node_k = all_k[node_id * ngpus * bsz:(node_id + 1) * ngpus * bsz]
        else:
            all_k = node_k = k.detach()         # (the reference's keys come out of an all_gather: no gradient, only q has one)
        k = node_k[reverse_ids[lo:lo + bsz]]
        return q, k, all_k

    # -- loss helper ------------------------------------------------------------------------------
    @staticmethod
    def _compute_loss_accuracy(logits, target, criterion):
        """logits: list of [B,K+1] logits; target: labels; criterion: typically nn.CrossEntropyLoss."""
        losses = [criterion(logit, target) for logit in logits]
        accuracies = [accuracy(logit, target)[0] for logit in logits]
        return losses, accuracies

    # -- DP: keep the trainable criterion modules (atts_q, embed_s) in sync across ranks (fixes Q7) --
    # They are not under DDP (their forward is called module by module), so their gradients get ONE flat all-reduce per
    # step, launched from autograd hooks the moment the last of them is accumulated -- these modules sit at the top of the
    # graph, so the collective overlaps the whole backbone backward -- and waited for right before optimizer.step().
    def attach_grad_sync(self, params):
        """Register the hooks once per parameter list (no-op without a process group or at world size 1 -- unless
        `self.grad_sync_single_rank` is set: the one-rank rehearsal of the collective path, bench.py MOMA_BENCH_FORCE_DIST)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() == 1 and not getattr(self, "grad_sync_single_rank", False):
            return
        params = [p for p in params if p.requires_grad]
        if getattr(self, "_gs_params", None) is not None and [id(p) for p in self._gs_params] == [id(p) for p in params]:
            return
        for h in getattr(self, "_gs_handles", []):
            h.remove()
        self._gs_params, self._gs_seen, self._gs_work, self._gs_flat = params, 0, None, None
        # how many of them receive a gradient per backward: learnt from the first step (atts_k / atts_queue / embed_t are
        # in the list but never get one -- SURVEY Q6), until then the blocking fallback in finish_grad_sync() serves
        self._gs_expect = len(params)

        def hook(_p):
            self._gs_seen += 1
            if self._gs_seen == self._gs_expect:                # every criterion gradient of this backward is there
                self._launch_grad_sync()
        self._gs_handles = [p.register_post_accumulate_grad_hook(hook) for p in params]

    def _launch_grad_sync(self):
        grads = [p.grad for p in self._gs_params if p.grad is not None]
        if not grads:
            return
        self._gs_grads = grads
        self._gs_flat = torch.cat([g.reshape(-1) for g in grads])
        self._gs_work = dist.all_reduce(self._gs_flat, async_op=True)
        self.grad_sync_launches = getattr(self, "grad_sync_launches", 0) + 1

    def finish_grad_sync(self):
        """Wait for the step's criterion all-reduce (launching it now if the hooks did not see every gradient, e.g. a
        module that took no part in this step's graph), average, and scatter back into the .grad tensors."""
        if getattr(self, "_gs_params", None) is None:
            return
        if self._gs_work is not None and self._gs_seen != self._gs_expect:
            # the hooks launched the reduce when the EXPECTED number of gradients had arrived, and more arrived after it (another
            # branch of the criterion took part in this backward, a head was un-frozen): the flat buffer in flight misses
            # them.  Drop it -- the .grad tensors are still the local, un-reduced ones -- and reduce everything, blocking.
            self._gs_work.wait()
            self._gs_work, self._gs_flat = None, None
        if self._gs_work is None:
            if self._gs_seen > 0:
                self._gs_expect = self._gs_seen                 # from the next step on the hooks launch it early
            self._launch_grad_sync()
        if self._gs_work is not None:
            self._gs_work.wait()
            flat = self._gs_flat.div_(dist.get_world_size())
            torch._foreach_copy_(self._gs_grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in self._gs_grads]),
                                                                               self._gs_grads)])
        self._gs_seen, self._gs_work, self._gs_flat = 0, None, None

    @staticmethod
    def allreduce_grads(params, single_rank=False, group=None):
        """ONE flat all-reduce (average) over the gradients of `params` on the default / given group -- the stateless form of
        `learning/ddp.py:FlatDataParallel.allreduce_grads` (which also verifies that the ranks reduce the same gradient set;
        the loop calls that one).  -> number of collectives launched."""
        if not (dist.is_available() and dist.is_initialized()):
            return 0
        world = dist.get_world_size(group)
        if world == 1 and not single_rank:
            return 0
        by_kind = {}
        for p in params:
            if p.grad is not None:
                by_kind.setdefault((p.grad.dtype, p.grad.device), []).append(p.grad)
        for grads in by_kind.values():                      # (one group in practice: every gradient here is fp32)
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, group=group)
            flat.div_(world)
            torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
        return len(by_kind)
