"""CPU restatement of the whole MoMA train step -- TEST / BASELINE INFRASTRUCTURE ONLY.

Plain PyTorch fp32 ops in the reference's op order (helper/loops_moma.py:244-372, moma branch :308-335):
per-tensor EMA, materialised [H,N,N] attention, queue clone -> mm -> cat -> div -> CrossEntropy,
index_copy_ enqueue.  Used by
  * tests/test_step_oracle_golden.py : pinned against the 10-step loss / pointer / queue trace captured from
    the reference itself (tests/golden/g5_step_trace.npz)  -> parity status PINNED;
  * tests (GPU)                      : the checker the HIP-backed loop is compared with;
  * bench.py cpu_baseline            : timed on the GPU box's host cores ("kind": "port").
Never imported by moma_amd/.  Backbones are the torch modules of moma_amd.backbones run on the CPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class OracleAttention(nn.Module):
    """MoMA/criterion_moco_att.py:141-167"""

    def __init__(self, dim, num_heads=4, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        x = x.unsqueeze(0)
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(N, C)
        return self.proj(x)


def _head(kind, in_dim, feat_dim):
    """MoMA/criterion_moco_att.py:254-305 (Flatten = index 0, Normalize = last)"""
    flat, norm = nn.Flatten(1), _Normalize()
    if kind == "mlp":
        return nn.Sequential(flat, nn.Linear(in_dim, in_dim), nn.ReLU(inplace=True), nn.Linear(in_dim, feat_dim), norm)
    if kind == "mlp_byol":                                                     # :269-285
        return nn.Sequential(flat, nn.Linear(in_dim, in_dim), nn.BatchNorm1d(in_dim), nn.ReLU(inplace=True),
                             nn.Linear(in_dim, feat_dim), norm)
    if kind == "linear":                                                       # :286-297
        return nn.Sequential(flat, nn.Linear(in_dim, feat_dim), norm)
    return nn.Sequential(flat, norm)


class _Normalize(nn.Module):
    def forward(self, x):
        return F.normalize(x, p=2, dim=1)


class OracleCMO(nn.Module):
    """MoMA/criterion_moco_att.py:236-338, opt.attn == 'self' branch"""

    def __init__(self, head, s_dim, t_dim, feat_dim, num_heads=4, attn="self"):
        super().__init__()
        self.embed_s = _head(head, s_dim, feat_dim)
        self.embed_t = _head(head, t_dim, feat_dim)
        if attn in ("all", "self_mix", "qk"):                                  # :307-333, module layout by opt.attn
            self.atts = OracleAttention(feat_dim, num_heads)
        elif attn in ("dual", "dual2"):
            self.atts_p = OracleAttention(feat_dim, num_heads)
            self.atts_n = OracleAttention(feat_dim, num_heads)
        elif attn in ("self_qk", "self_nomix"):
            self.atts_q = OracleAttention(feat_dim, num_heads)
            self.atts_k = OracleAttention(feat_dim, num_heads)
        else:
            self.atts_q = OracleAttention(feat_dim, num_heads)
            self.atts_k = OracleAttention(feat_dim, num_heads)
            self.atts_queue = OracleAttention(feat_dim, num_heads)


class OracleMoCo(nn.Module):
    """MoMA/mem_moco.py:6-100"""

    def __init__(self, n_dim, K, T):
        super().__init__()
        self.K, self.T, self.index = K, T, 0
        self.register_buffer("memory", F.normalize(torch.randn(K, n_dim)))

    def forward(self, q, k, all_k=None):
        bsz = q.size(0)
        k = k.detach()
        queue = self.memory.clone().detach()                                   # :89
        pos = torch.bmm(q.view(bsz, 1, -1), k.view(bsz, -1, 1)).view(bsz, 1)   # :37-38
        neg = torch.mm(queue, q.transpose(1, 0)).transpose(0, 1)               # :41-42
        out = torch.div(torch.cat((pos, neg), dim=1), self.T).contiguous()     # :44-47
        labels = torch.zeros(bsz, dtype=torch.long)
        all_k = all_k if all_k is not None else k
        with torch.no_grad():                                                  # :23-27
            ids = torch.fmod(torch.arange(all_k.shape[0]) + self.index, self.K).long()
            self.memory.index_copy_(0, ids, all_k)
        self.index = (self.index + all_k.size(0)) % self.K                     # :14-15
        return out, labels


class OracleMoCoAtt(OracleMoCo):
    """MoMA/mem_moco.py:103-161: the memory applies the teacher-student cross-attention variant before the logits."""

    def forward(self, q, k, all_k=None, attn=None, criterion_kd=None):
        bsz = q.size(0)
        k = k.detach()
        queue = self.memory.clone().detach()
        if attn == "all":                                                      # :124-126
            out = criterion_kd.atts(torch.cat([q, k, queue], dim=0))
            q, k, queue = out[:bsz], out[bsz:2 * bsz], out[2 * bsz:]
        elif attn == "qk":                                                     # :127-129
            out = criterion_kd.atts(torch.cat([q, k], dim=0))
            q, k = out[:bsz], out[bsz:]
        elif attn == "dual":                                                   # :130-134
            out_p = criterion_kd.atts_p(torch.cat([q, queue], dim=0))
            q, queue = out_p[:bsz], out_p[bsz:]
            out_n = criterion_kd.atts_n(torch.cat([k, queue], dim=0))
            k, queue = out_n[:bsz], out_n[bsz:]
        elif attn in ("self_qk", "self_qkv2"):                                 # :140-142
            q = criterion_kd.atts_q(q)
            k = criterion_kd.atts_k(k)
        else:                                                                  # :143-146
            q = criterion_kd.atts_q(q)
            k = criterion_kd.atts_k(k)
            queue = criterion_kd.atts_queue(queue)
        pos = torch.bmm(q.view(bsz, 1, -1), k.view(bsz, -1, 1)).view(bsz, 1)
        neg = torch.mm(queue, q.transpose(1, 0)).transpose(0, 1)
        out = torch.div(torch.cat((pos, neg), dim=1), self.T).contiguous()
        labels = torch.zeros(bsz, dtype=torch.long)
        all_k = all_k if all_k is not None else k
        with torch.no_grad():
            ids = torch.fmod(torch.arange(all_k.shape[0]) + self.index, self.K).long()
            self.memory.index_copy_(0, ids, all_k.detach())
        self.index = (self.index + all_k.size(0)) % self.K
        return out, labels


def momentum_update(model, model_ema, m):
    """learning/contrast_trainer.py:207-211"""
    for p1, p2 in zip(model.parameters(), model_ema.parameters()):
        p2.data.mul_(m).add_(p1.detach().data, alpha=(1 - m))


def shuffle_bn(x, model_ema, head):
    """learning/contrast_trainer.py:90-133 at world size 1: all_k is gathered BEFORE the un-shuffle."""
    bsz = x.size(0)
    shuffle_ids = torch.randperm(bsz)
    reverse_ids = torch.argsort(shuffle_ids)
    with torch.no_grad():
        feat_t, _ = model_ema(x[shuffle_ids], is_feat=True)
        k = head(feat_t[-1])
    all_k = k
    return all_k[reverse_ids], all_k


def shuffle_bn_attn(x, model_ema, head, cmo, q, attn):
    """learning/contrast_trainer.py:135-187 at world size 1: keys of the shuffled batch, attention over [q ; k]
    ('self_mix': cmo.atts) or per side (atts_q / atts_k) BEFORE the un-shuffle; all_k stays in shuffled order."""
    bsz = x.size(0)
    shuffle_ids = torch.randperm(bsz)
    reverse_ids = torch.argsort(shuffle_ids)
    with torch.no_grad():
        feat_t, _ = model_ema(x[shuffle_ids], is_feat=True)
        k = head(feat_t[-1])
    if attn == "self_mix":
        out = cmo.atts(torch.cat([q, k], dim=0))
        q, k = out[:bsz], out[bsz:]
    else:
        q = cmo.atts_q(q)
        k = cmo.atts_k(k)
    all_k = k.detach()                 # (:175 all_gather output: carries no gradient -- only q does)
    return q, all_k[reverse_ids], all_k


def distill_kl(y_s, y_t, T):
    """distiller_zoo/KD.py:13-17"""
    p_s = F.log_softmax(y_s / T, dim=1)
    p_t = F.softmax(y_t / T, dim=1)
    return nn.KLDivLoss(reduction="batchmean")(p_s, p_t) * (T ** 2)


def accuracy_top1(output, target):
    """learning/util.py:25-41, k=1"""
    pred = output.topk(1, 1, True, True)[1].t()
    return pred.eq(target.view(1, -1)).float().sum() * (100.0 / target.size(0))


class StepOracle:
    """State of one training run (student, EMA teacher, CMO, queue, SGD) + `step(images, labels)`."""

    def __init__(self, model_s, model_t, cmo: OracleCMO, contrast: OracleMoCo, head="None", alpha=0.999,
                 cls=1.0, div=1.0, beta=1.0, kd_T=4.0, lr=0.05, momentum=0.9, weight_decay=1e-4, ema=True, attn="self"):
        self.model_s, self.model_t, self.cmo, self.contrast = model_s, model_t, cmo, contrast
        self.head, self.alpha, self.cls, self.div, self.beta, self.kd_T, self.ema = head, alpha, cls, div, beta, kd_T, ema
        self.attn = attn            # 'self' = the reference loop; self_mix / self_nomix and a MoCoAtt memory = the widened paths
        trainable = nn.ModuleList([model_s] + [getattr(cmo, n) for n in ("atts", "atts_p", "atts_n", "atts_q", "atts_k",
                                                                         "atts_queue") if hasattr(cmo, n)])   # :339-356
        if head == "mlp":
            trainable.append(cmo.embed_s)
        self.optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)

    def start_epoch(self):
        self.model_s.train()
        self.model_t.train()
        self.cmo.train()
        self.model_t.eval()                                                   # helper/loops_moma.py:224-227

    def step(self, images, labels):
        ms, mt, cmo = self.model_s, self.model_t, self.cmo
        feat_s, logit_s = ms(images, is_feat=True)                            # :268
        with torch.no_grad():
            _, logit_t = mt(images, is_feat=True)                             # :270-272
        loss_cls = F.cross_entropy(logit_s, labels)                           # :278
        loss_div = distill_kl(logit_s, logit_t, self.kd_T)                    # :279
        if self.ema:
            momentum_update(ms, mt, self.alpha)                               # :309
            if self.head == "mlp":
                cmo.embed_t.eval()
                momentum_update(cmo.embed_s, cmo.embed_t, self.alpha)         # :310-312
        for m in mt.modules():                                                # :314-318
            if m.__class__.__name__.find("BatchNorm") != -1:
                m.train()
        f_s = cmo.embed_s(feat_s[-1])                                         # :323-324
        if self.attn in ("self_mix", "self_nomix"):                           # learning/contrast_trainer.py:135-187
            f_s, k, all_k = shuffle_bn_attn(images, mt, cmo.embed_t, cmo, f_s, self.attn)
            logits, labels0 = self.contrast(q=f_s, k=k, all_k=all_k)
        elif isinstance(self.contrast, OracleMoCoAtt):                        # MoMA/mem_moco.py:111-161
            k, all_k = shuffle_bn(images, mt, cmo.embed_t)                    # :320
            logits, labels0 = self.contrast(q=f_s, k=k, all_k=all_k, attn=self.attn, criterion_kd=cmo)
        else:
            k, all_k = shuffle_bn(images, mt, cmo.embed_t)                    # :320
            f_s = cmo.atts_q(f_s)                                             # :326-329
            k = cmo.atts_k(k)
            all_k = cmo.atts_queue(all_k)
            logits, labels0 = self.contrast(q=f_s, k=k, all_k=all_k)          # :331
        loss_kd = F.cross_entropy(logits, labels0)                            # :322,332-335
        loss = self.cls * loss_cls + self.div * loss_div + self.beta * loss_kd  # :350
        acc = accuracy_top1(logit_s, labels)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return float(loss.item()), float(acc.item()), float(loss_kd.item())
