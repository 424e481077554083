"""MoCo-style feature queue of the MoMA step, backed by the HIP library.

Drop-in for the reference's MoMA/mem_moco.py (`BaseMoCo` :6-66, `MoCo` :69-100, `build_mem` :256-272):
same constructor arguments, buffer name (`memory`), attributes (`K`, `T`, `index`) and
`forward(q, k, all_k=None) -> (logits [B,K+1], labels [B])`.  The op chains it replaces:

  _compute_logit  (bmm + mm + transpose + cat + div)   -> moma_infonce_logits        (K2)
  _update_memory  (arange + fmod + index_copy_)        -> moma_enqueue               (K3)
  _update_pointer (host integer)                       -> unchanged, bit-exact contract

plus `forward_fused`, the one-pass replacement of MoCo.forward + CrossEntropyLoss(label 0) + top-1
(helper/loops_moma.py:322,331-335): no [B,K+1] logits, no queue clone, dq produced from the pre-enqueue
queue in the same pass.  All tensors must live on the GPU; there is no CPU path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class BaseMoCo(nn.Module):
    """base class for MoCo-style memory cache (reference MoMA/mem_moco.py:6-66)"""

    def __init__(self, K=65536, T=0.07, precision="fp32"):
        super().__init__()
        self.K = K
        self.T = T
        self.index = 0            # host-side ring pointer; not part of state_dict (as in the reference)
        self.precision = precision

    def _update_pointer(self, bsz):
        self.index = (self.index + bsz) % self.K

    # The reference never checkpoints the queue or its pointer; here the pointer rides along in the
    # state_dict as extra state so that a resumed run continues the ring buffer where it stopped.
    def get_extra_state(self):
        return {"index": int(self.index)}

    def set_extra_state(self, state):
        self.index = int(state.get("index", 0)) % self.K

    def _update_memory(self, k, queue):
        """queue[(index + i) mod K] = k[i]   (reference :17-27)"""
        with torch.no_grad():
            ops.enqueue_(queue, k.detach().contiguous().float(), self.index)

    def _compute_logit(self, q, k, queue):
        """[B,K+1] logits: pos | neg, divided by T   (reference :29-49)"""
        return ops.infonce_logits(q, k, queue, self.T, self.precision)


class MoCo(BaseMoCo):
    """Single-modal MoCo-style cache (reference MoMA/mem_moco.py:69-100).

    queue_dtype=torch.bfloat16 stores the K x d queue in 2-byte rows (half the HBM traffic of K2); with
    precision='bf16' this is numerically identical to fp32 storage because the bf16 MFMA path rounds the
    keys to bf16 at load time anyway.
    """

    def __init__(self, n_dim, K=65536, T=0.07, mem_name="memory", queue_dtype=torch.float32, precision="fp32"):
        super().__init__(K, T, precision)
        # same RNG consumption and normalisation as the reference (:73-75)
        self.register_buffer(mem_name, torch.randn(K, n_dim))
        self.memory = F.normalize(self.memory)
        if queue_dtype != torch.float32:
            self.memory = self.memory.to(queue_dtype)
        self._shadow, self._shadow_key = None, None

    def forward(self, q, k, all_k=None):
        """Reference-compatible: materialised logits and zero labels, then enqueue (:77-100)."""
        bsz = q.size(0)
        k = k.detach()
        # the backward of the logits needs the PRE-enqueue queue -> snapshot, as the reference does (:89)
        queue = self.memory.clone().detach() if (torch.is_grad_enabled() and q.requires_grad) else self.memory
        logits = self._compute_logit(q, k, queue)
        labels = torch.zeros(bsz, dtype=torch.long, device=q.device)
        all_k = all_k if all_k is not None else k
        self._update_memory(all_k, self.memory)
        self._update_pointer(all_k.size(0))
        return logits, labels

    def _bf16_shadow(self):
        """bf16 mirror of an fp32 `memory` for the bf16 policy: the one-pass kernel streams 2-byte rows (half the
        bytes, LDS-DMA without conversion) while `memory` keeps the reference's fp32 storage / state_dict.  The
        bf16 MFMA path rounds the keys to bf16 anyway, so results are identical to reading the fp32 rows.  Rebuilt
        when `memory` was replaced (.cuda(), load_state_dict) or modified in place by anything but _update_memory."""
        mem = self.memory
        key = (mem.data_ptr(), mem._version, tuple(mem.shape))
        if self._shadow is None or self._shadow_key != key or self._shadow.device != mem.device:
            self._shadow = mem.to(torch.bfloat16)
            self._shadow_key = key
        return self._shadow

    def prefetch(self, stream=None):
        """Cache hint before forward_fused: sweep the queue the one-pass kernel will stream (the bf16 mirror under fp32 storage)
        into the Infinity Cache, on a side stream while latency-bound work runs on the main one.  No effect on results."""
        mem = self.memory
        if mem.is_cuda and mem.dtype == torch.float32 and ops.prec_code(self.precision) == ops.PREC_BF16:
            mem = self._bf16_shadow()
        if mem.is_cuda:
            ops.queue_prefetch(mem, stream)

    def qpack(self, B, d, device):
        """A packed-query buffer for the producer of q (Attention.forward(x, qpack=...)): K2 then runs no pre-pack launch.
        None when the one-pass kernel does not take this configuration."""
        if ops.prec_code(self.precision) != ops.PREC_BF16:
            return None
        if getattr(self, "_qpack", None) is None:
            self._qpack = ops.QPack()
        return self._qpack.prepare(B, d, self.T, device)

    def forward_fused(self, q, k, all_k=None, qpack=None):
        """One pass over the queue -> (loss_kd, top-1 accuracy in percent [1]); then enqueue.

        loss_kd == CrossEntropyLoss(logits, zeros) of the reference loop; its gradient w.r.t. q is produced
        by the same kernel from the pre-enqueue queue, so no clone is needed."""
        k = k.detach()
        shadow = None
        if self.memory.dtype == torch.float32 and ops.prec_code(self.precision) == ops.PREC_BF16 and self.memory.is_cuda:
            shadow = self._bf16_shadow()
        rows = (all_k if all_k is not None else k).detach().contiguous().float()
        loss_rows, _lse, top1 = ops.infonce_fused(q, k, self.memory if shadow is None else shadow, self.T, self.precision, qpack,
                                                  enq=self._enqueue_job(rows, shadow))
        self._enqueued(rows.size(0), shadow)
        return loss_rows.mean(), top1.float().mean(0, keepdim=True) * 100.0

    def _enqueue_job(self, rows, shadow):
        """(rows, pointer, fp32 queue | None) for ops.infonce_fused*(enq=...): _update_memory (reference :97-99) on the K2 call's last
        launch -- behind every read of the pre-enqueue queue, as MoCo.forward orders them -- instead of a launch of its own"""
        return (rows, self.index, self.memory if shadow is not None else None)

    def _enqueued(self, n, shadow):
        """host-side bookkeeping of an enqueue the K2 call carried: the mirror is current, the pointer moves (_update_pointer)"""
        if shadow is not None:
            self._shadow_key = (self.memory.data_ptr(), self.memory._version, tuple(self.memory.shape))
        self._update_pointer(n)

    def forward_fused_into(self, q, k, all_k, qpack_buf, out):
        """forward_fused without autograd, results into caller-owned static buffers (`out`: ops.K2Buffers) -- the form a step
        replayed from HIP graphs calls between its forward and its backward graph (helper/step_graph.py; the autograd side
        is ops.StaticK2Loss).  Same kernels, same order: one pass over the pre-enqueue queue, then the enqueue; the pointer
        stays the host integer."""
        shadow = None
        if self.memory.dtype == torch.float32 and ops.prec_code(self.precision) == ops.PREC_BF16 and self.memory.is_cuda:
            shadow = self._bf16_shadow()
        rows = (all_k if all_k is not None else k).detach().contiguous().float()
        ops.infonce_fused_into(q, k, self.memory if shadow is None else shadow, self.T, self.precision, qpack_buf, out,
                               enq=self._enqueue_job(rows, shadow))
        self._enqueued(rows.size(0), shadow)


class MoCoAtt(BaseMoCo):
    """MoCo cache whose forward applies the teacher-student CROSS-attention variants before the logits
    (reference MoMA/mem_moco.py:103-161).  `criterion_kd` is the CMO instance holding the Attention modules
    (K1, HIP); `attn` selects the variant:
      'qk'      one module over the 2B tokens [q ; k]                      'self_qk' separate modules on q and k
      'dual2'   atts_p over [q ; k] -> q, atts_n over [k ; q] -> k, positive logit only
      'all'     one module over [q ; k ; queue]      'dual'  atts_p over [q ; queue], atts_n over [k ; queue]
      default   atts_q(q), atts_k(k), atts_queue(queue)
    The variants that attend over the queue ('all', 'dual', default) run the attention flash-style on the K1 fast path
    (row log-sum-exp kept, P recomputed per tile in the backward): no [H, N, N] array exists, N = K + B / 2B tokens are
    fine (tested at N = 8224); only the exact-fp32 policy materialises the scores like the reference."""

    def __init__(self, n_dim, K=65536, T=0.07, mem_name="memory", queue_dtype=torch.float32, precision="fp32"):
        super().__init__(K, T, precision)
        self.register_buffer(mem_name, torch.randn(K, n_dim))
        self.memory = F.normalize(self.memory)

    def _compute_logit_qk(self, q, k):
        """positive logit only (reference :51-66); [B] after the reference's squeeze"""
        return ((q * k).sum(dim=1) / self.T).contiguous()

    def forward(self, q, k, all_k=None, attn=None, criterion_kd=None):
        logits, labels, k = self.forward_logits(q, k, attn=attn, criterion_kd=criterion_kd)
        all_k = all_k if all_k is not None else k
        self._update_memory(all_k, self.memory)
        self._update_pointer(all_k.size(0))
        return logits, labels

    def forward_logits(self, q, k, attn=None, criterion_kd=None):
        """`forward` up to the enqueue: (logits, labels, k behind its attention module) from a snapshot of the queue (reference
        :111-147).  The step served from HIP graphs captures this part and issues the enqueue -- whose ring pointer is a host
        integer -- between its graphs (helper/step_graph.py)."""
        bsz = q.size(0)
        k = k.detach()
        queue = self.memory.clone().detach()
        cat = lambda *ts: torch.cat([t.float() for t in ts], dim=0)
        if attn == "all":
            out = criterion_kd.atts(cat(q, k, queue))
            q, k, queue = out[:bsz], out[bsz:2 * bsz], out[2 * bsz:]
        elif attn == "qk":
            out = criterion_kd.atts(cat(q, k))
            q, k = out[:bsz], out[bsz:]
        elif attn == "dual":
            out_p = criterion_kd.atts_p(cat(q, queue))
            q, queue = out_p[:bsz], out_p[bsz:]
            out_n = criterion_kd.atts_n(cat(k, queue))
            k, queue = out_n[:bsz], out_n[bsz:]
        elif attn == "dual2":
            q = criterion_kd.atts_p(cat(q, k))[:bsz]
            k = criterion_kd.atts_n(cat(k, q))[:bsz]
        elif attn in ("self_qk", "self_qkv2"):
            q = criterion_kd.atts_q(q)
            k = criterion_kd.atts_k(k)
        else:
            q = criterion_kd.atts_q(q)
            k = criterion_kd.atts_k(k)
            queue = criterion_kd.atts_queue(queue)
        if attn == "dual2":
            logits = self._compute_logit_qk(q, k)
        else:
            logits = self._compute_logit(q.contiguous(), k.contiguous(), queue.contiguous())
        labels = torch.zeros(bsz, dtype=torch.long, device=q.device)
        return logits, labels, k

    def enqueue_keys(self, all_k):
        """_update_memory + _update_pointer (reference :150-152) as a call of its own"""
        self._update_memory(all_k, self.memory)
        self._update_pointer(all_k.size(0))


class _DualQueue(BaseMoCo):
    """Two queues (`memory_s`, `memory_t`) sharing one pointer -- base of MoCoST / MoCoSSTT
    (reference MoMA/mem_moco.py:165-253).  Same RNG consumption and normalisation as the reference."""

    def __init__(self, n_dim, K=65536, T=0.07, queue_dtype=torch.float32, precision="fp32"):
        super().__init__(K, T, precision)
        self.register_buffer("memory_s", torch.randn(K, n_dim))
        self.register_buffer("memory_t", torch.randn(K, n_dim))
        self.memory_s = F.normalize(self.memory_s)
        self.memory_t = F.normalize(self.memory_t)
        if queue_dtype != torch.float32:
            self.memory_s = self.memory_s.to(queue_dtype)
            self.memory_t = self.memory_t.to(queue_dtype)

    def _snap(self, mem, *qs):
        need = torch.is_grad_enabled() and any(q is not None and q.requires_grad for q in qs)
        return mem.clone().detach() if need else mem

    def _enqueue_both(self, k, k_t, all_k, all_k_t):
        all_k = all_k if all_k is not None else k
        all_k_t = all_k_t if all_k_t is not None else k_t
        self._update_memory(all_k, self.memory_s)
        self._update_memory(all_k_t, self.memory_t)
        self._update_pointer(all_k.size(0))


class MoCoST(_DualQueue):
    """student / teacher queues: logits of q against (k, memory_s) and (k_t, memory_t)   (reference :165-201)"""

    def forward(self, q, k, k_t, all_k=None, all_k_t=None):
        k, k_t = k.detach(), k_t.detach()
        queue_s, queue_t = self._snap(self.memory_s, q), self._snap(self.memory_t, q)
        logits_ss = self._compute_logit(q, k, queue_s)
        logits_st = self._compute_logit(q, k_t, queue_t)
        labels = torch.zeros(q.size(0), dtype=torch.long, device=q.device)
        self._enqueue_both(k, k_t, all_k, all_k_t)
        return logits_ss, logits_st, labels

    def forward_fused(self, q, k, k_t, all_k=None, all_k_t=None):
        """-> ((loss_ss, loss_st), (acc_ss, acc_st)): both InfoNCE terms in ONE sweep over the two queues
        (ops.infonce_fused_multi: one pre-pack of q, one launch streaming memory_s then memory_t, one combine)."""
        k, k_t = k.detach(), k_t.detach()
        (l_ss, _, t_ss), (l_st, _, t_st) = ops.infonce_fused_multi(
            [(q, k, self.memory_s), (q, k_t, self.memory_t)], self.T, self.precision)
        self._enqueue_both(k, k_t, all_k, all_k_t)
        acc = lambda t: t.float().mean(0, keepdim=True) * 100.0
        return (l_ss.mean(), l_st.mean()), (acc(t_ss), acc(t_st))


class MoCoSSTT(_DualQueue):
    """four-way variant: optional teacher query q_t adds logits_ts / logits_tt   (reference :205-253)"""

    def forward(self, q, k, q_t=None, k_t=None, all_k=None, all_k_t=None):
        k, k_t = k.detach(), k_t.detach()
        queue_s, queue_t = self._snap(self.memory_s, q, q_t), self._snap(self.memory_t, q, q_t)
        logits_ss = self._compute_logit(q, k, queue_s)
        logits_st = self._compute_logit(q, k_t, queue_t)
        if q_t is not None:
            logits_ts = self._compute_logit(q_t, k, queue_s)
            logits_tt = self._compute_logit(q_t, k_t, queue_t)
        labels = torch.zeros(q.size(0), dtype=torch.long, device=q.device)
        self._enqueue_both(k, k_t, all_k, all_k_t)
        if q_t is not None:
            return logits_ss, logits_st, logits_ts, logits_tt, labels
        return logits_ss, logits_st, labels

    def forward_fused(self, q, k, q_t=None, k_t=None, all_k=None, all_k_t=None):
        """-> (losses, accuracies) in the reference's order (ss, st[, ts, tt]): the 2 / 4 InfoNCE terms of :230-238 in ONE sweep
        over the two queues (two query sets packed by one launch), without materialising any [B,K+1] logits."""
        k, k_t = k.detach(), k_t.detach()
        terms = [(q, k, self.memory_s), (q, k_t, self.memory_t)]
        if q_t is not None:
            terms += [(q_t, k, self.memory_s), (q_t, k_t, self.memory_t)]
        res = ops.infonce_fused_multi(terms, self.T, self.precision)
        self._enqueue_both(k, k_t, all_k, all_k_t)
        return tuple(r[0].mean() for r in res), tuple(r[2].float().mean(0, keepdim=True) * 100.0 for r in res)


def build_mem(opt):
    """Factory on opt.mem (reference MoMA/mem_moco.py:256-272).  Extra, optional opt fields:
    `moma_prec` ('fp32' | 'bf16') and `queue_dtype` ('fp32' | 'bf16')."""
    prec = getattr(opt, "moma_prec", "fp32")
    qdt = {"fp32": torch.float32, "bf16": torch.bfloat16}[getattr(opt, "queue_dtype", "fp32")]
    if opt.mem == "MoCoSSTT":
        return MoCoSSTT(opt.feat_dim, opt.nce_k, opt.nce_t, queue_dtype=qdt, precision=prec)
    if opt.mem == "MoCoST":
        return MoCoST(opt.feat_dim, opt.nce_k, opt.nce_t, queue_dtype=qdt, precision=prec)
    if opt.mem == "MoCoAtt":
        return MoCoAtt(opt.feat_dim, opt.nce_k, opt.nce_t, queue_dtype=qdt, precision=prec)
    return MoCo(opt.feat_dim, opt.nce_k, opt.nce_t, queue_dtype=qdt, precision=prec)   # reference default branch
