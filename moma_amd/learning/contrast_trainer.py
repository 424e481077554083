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

import torch
import torch.distributed as dist

from .. import ops
from .base_trainer import BaseTrainer
from .util import accuracy


class ContrastTrainer(BaseTrainer):
    """trainer for contrastive distillation"""

    _ema_tables = {}

    def __init__(self, args):
        super().__init__(args)

    # -- K4 ---------------------------------------------------------------------------------------
    @staticmethod
    def momentum_update(model, model_ema, m):
        """model_ema = m * model_ema + (1 - m) * model, parameters only (BN buffers are not averaged).

        Like the reference's zip over parameters() this requires identical architectures; a shape mismatch
        raises RuntimeError -- before touching the teacher, unlike the reference which fails half-way (Q4)."""
        ps = [p.detach() for p in model.parameters()]
        es = [p.detach() for p in model_ema.parameters()]
        key = (id(model), id(model_ema))
        tab = ContrastTrainer._ema_tables.get(key)
        if tab is None or not tab.matches(ps, es):
            tab = ops.EmaTable(ps, es)
            ContrastTrainer._ema_tables[key] = tab
        ops.ema_update_(tab, m)

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
        shuffle_ids = torch.randperm(bsz).to(x.device)          # host RNG stream, as the reference (:108)
        reverse_ids = torch.argsort(shuffle_ids)
        with torch.no_grad():
            feat_t, _ = model_ema(x[shuffle_ids], is_feat=True)
            all_k = model_ema_head(feat_t[-1])
        k = all_k[reverse_ids]
        return k, all_k

    def _shuffle_bn_gather(self, x, model_ema, model_ema_head):
        args = self.args
        gp = self.local_group
        bsz = x.size(0)
        node_x = [torch.ones_like(x) for _ in range(dist.get_world_size(gp))]
        dist.all_gather(node_x, x.contiguous(), group=gp, async_op=False)
        node_x = torch.cat(node_x, dim=0)
        shuffle_ids = torch.randperm(bsz * dist.get_world_size(gp)).to(x.device)
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
            shuffle_ids = torch.randperm(bsz * dist.get_world_size(gp)).to(x.device)
            reverse_ids = torch.argsort(shuffle_ids)
            dist.broadcast(shuffle_ids, 0)
            dist.broadcast(reverse_ids, 0)
            lo = args.local_rank * bsz
        else:
            node_x = x
            shuffle_ids = torch.randperm(bsz).to(x.device)
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
            all_k = node_k = k
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
    @staticmethod
    def allreduce_grads(params):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat.div_(dist.get_world_size())
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
