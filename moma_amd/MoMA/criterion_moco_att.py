"""CMO criterion: projection heads + batch-token multi-head attention modules of the MoMA step.

Drop-in for the reference's MoMA/criterion_moco_att.py (`Normalize` :12-18, `Flatten` :21-27,
`Attention` :141-167, `CMO` :236-338): same constructor arguments, sub-module names and state-dict keys
(`atts_q.qkv.weight`, `atts_q.proj.bias`, `embed_s.1.weight`, ...).  `Attention.forward` -- the chain
Linear -> reshape/permute -> q k^T * scale -> softmax -> attn v -> transpose -> Linear -- runs in the HIP
library (K1: moma_mha_fwd_fast / moma_mha_bwd_fast under the bf16 policy, moma_mha_fwd / moma_mha_bwd under exact fp32).  Heads (Flatten / Linear / ReLU / L2-normalise on [B,s_dim])
stay stock torch ops, as the scope table allows.
"""
import torch
from torch import nn
import torch.nn.functional as F

from .. import ops


class Normalize(nn.Module):
    def __init__(self, p=2):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F.normalize(x, p=self.p, dim=1)


class Flatten(nn.Module):
    @staticmethod
    def forward(x):
        return torch.flatten(x, 1)


class Attention(nn.Module):
    """Attention ACROSS THE SAMPLES OF A BATCH: x [N,dim] is one sequence of N tokens.

    Parameters are two nn.Linear modules exactly as in the reference (default kaiming-uniform init, so the
    RNG stream and checkpoints are interchangeable); they are only containers here -- the arithmetic is the
    fused HIP path.  attn_drop / proj_drop must be 0 (the reference always passes 0.)."""

    def __init__(self, dim, num_heads=12, qkv_bias=False, attn_drop=0., proj_drop=0., precision="fp32"):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout inside the batch-token attention is not used by MoMA")
        if dim % num_heads:
            raise ValueError("dim must be divisible by num_heads")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.precision = precision
        self._pack = ops.MhaPack()          # bf16 weight copies of the fast path (not a parameter, not in the state dict)

    def invalidate_pack(self):
        """Call after writing the weights through anything that bypasses autograd's version counters: a raw-pointer kernel,
        or in-place arithmetic on `weight.data` (the reference's own EMA idiom, `p.data.mul_(m).add_(...)`: `.data` carries
        a version counter of its own).  optimizer.step(), load_state_dict(), .to() / .cuda() and ContrastTrainer.momentum_update
        are followed without it."""
        self._pack.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        self._pack.invalidate()
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, qpack=None):
        if x.dim() != 2:
            raise ValueError("Attention expects [N, dim] (batch rows are the tokens)")
        return ops.mha(self._input(x), self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                       self.num_heads, self.precision, self._pack, qpack)

    @staticmethod
    def _input(x):
        # a bf16 x (the output of a head under bf16 autocast) goes to the kernels as it stands: they round x to bf16 at
        # operand staging either way, so the values are identical and the bytes half; everything else is taken as fp32
        return x if x.dtype == torch.bfloat16 else x.float()

    @staticmethod
    def forward_group(modules, xs):
        """[m(x) for m, x in zip(modules, xs)] in one group of launches where possible (no-grad key side of the step:
        atts_k and atts_queue, reference helper/loops_moma.py:327-329)."""
        m0 = modules[0]
        if any(m.num_heads != m0.num_heads or m.precision != m0.precision for m in modules):
            return [m(x) for m, x in zip(modules, xs)]
        return ops.mha_group([(Attention._input(x), m.qkv.weight, m.qkv.bias, m.proj.weight, m.proj.bias, m._pack)
                              for m, x in zip(modules, xs)], m0.num_heads, m0.precision)


def _head(kind, in_dim, feat_dim):
    if kind == "mlp":
        return nn.Sequential(Flatten(), nn.Linear(in_dim, in_dim), nn.ReLU(inplace=True),
                             nn.Linear(in_dim, feat_dim), Normalize(2))
    if kind == "mlp_byol":
        return nn.Sequential(Flatten(), nn.Linear(in_dim, in_dim), nn.BatchNorm1d(in_dim), nn.ReLU(inplace=True),
                             nn.Linear(in_dim, feat_dim), Normalize(2))
    if kind == "linear":
        return nn.Sequential(Flatten(), nn.Linear(in_dim, feat_dim), Normalize(2))
    return nn.Sequential(Flatten(), Normalize(2))


class CMO(nn.Module):
    """Heads `embed_s` / `embed_t` by opt.head and attention modules by opt.attn (reference :251-338).

    Construction order (embed_s, embed_t, then the attention modules) matches the reference so that a
    seeded run draws identical initial weights.  Optional opt fields: `moma_prec`, `num_heads` (default 4,
    the value hard-coded in the reference)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.embed_s = _head(opt.head, opt.s_dim, opt.feat_dim)
        self.embed_t = _head(opt.head, opt.t_dim, opt.feat_dim)
        self.norm1 = nn.LayerNorm
        self.qkv_bias = True
        prec = getattr(opt, "moma_prec", "fp32")
        heads = getattr(opt, "num_heads", 4)

        def att():
            return Attention(opt.feat_dim, num_heads=heads, qkv_bias=self.qkv_bias, attn_drop=0., proj_drop=0.,
                             precision=prec)

        if opt.attn in ("all", "self_mix", "qk"):
            self.atts = att()
        elif opt.attn in ("dual", "dual2"):
            self.atts_p = att()
            self.atts_n = att()
        elif opt.attn in ("self_qk", "self_nomix"):
            self.atts_q = att()
            self.atts_k = att()
        elif opt.attn in ("self_qkv2", "selfv2", "self_viz"):
            # Attention2 (+residual+LayerNorm) / Attention_viz variants: unreachable from the CLI loop
            raise NotImplementedError("attn variant not built: {}".format(opt.attn))
        else:  # opt.attn == 'self'  (reference default branch :334-338)
            self.atts_q = att()
            self.atts_k = att()
            self.atts_queue = att()
