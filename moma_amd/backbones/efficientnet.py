"""EfficientNet-B0 backbone with the MoMA feature-list contract
`model(x, is_feat=True) -> ([reduction_1..4, head, pooled[B,1280,1,1]], logits)`, `get_feat_modules()`.

Own implementation of the published architecture (MBConv + squeeze-excite + swish, TF "same" padding,
stochastic depth).  Parameter names (_conv_stem, _bn0, _blocks.N._expand_conv/_depthwise_conv/_se_reduce/
_se_expand/_project_conv/_bn0.._bn2, _conv_head, _bn1, classifier_.1) match the reference's checkpoints
(models/efficientnet_pytorch/model.py:154-222) so they load unchanged.  The convolutions run on
PyTorch-ROCm / MIOpen: backbones are out of scope as kernels.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# Which PyTorch-ROCm path the two MIOpen-weak ops take (measured on MI355X, bf16 NCHW, B=256):
#   depthwise convolutions: MOMA_DW=hip (default) = the library's LDS-tiled depthwise kernels (dwconv.hip, forward /
#                           backward-data / backward-weight, SAME padding without a padded copy); "aten" = ATen's own
#                           depthwise kernels (k*k global loads per output: >50 % of the step); "miopen" = MIOpen, which
#                           has no tuned gfx950 solver and falls back to naive_conv_* kernels;
#   batch norm (+ SiLU)   : MOMA_BN=hip (default) = the library's fused BatchNorm+activation kernels (bn.hip: 3 HBM
#                           passes forward, 5 backward); "miopen" = nn.BatchNorm2d + F.silu (MIOpen's spatial BN runs at
#                           ~10 % of HBM peak on these shapes); "aten" = ATen's native BN (5 % slower than MIOpen).
_DW_MODE = os.environ.get("MOMA_DW", "hip")
_BN_MODE = os.environ.get("MOMA_BN", "hip")
#   squeeze-excite        : MOMA_SE=hip (default) = per-plane mean and the sigmoid gate (one-pass backward) on the
#                           library's kernels (se.hip); "aten" = adaptive_avg_pool2d / sigmoid / mul.
_SE_MODE = os.environ.get("MOMA_SE", "hip")

# (repeats, kernel, stride, expand, cin, cout, se_ratio) -- EfficientNet-B0 stage table
_B0_STAGES = [
    (1, 3, 1, 1, 32, 16, 0.25),
    (2, 3, 2, 6, 16, 24, 0.25),
    (2, 5, 2, 6, 24, 40, 0.25),
    (3, 3, 2, 6, 40, 80, 0.25),
    (3, 5, 1, 6, 80, 112, 0.25),
    (4, 5, 2, 6, 112, 192, 0.25),
    (1, 3, 1, 6, 192, 320, 0.25),
]
_BN_MOM, _BN_EPS = 0.01, 1e-3


class _DepthwiseNative(torch.autograd.Function):
    """Depthwise conv2d on ATen's own kernels in BOTH directions.  `cudnn.flags(enabled=False)` around the forward
    alone is not enough: autograd picks the backend again at backward time (-> MIOpen's naive_conv_*_bwd, 16 % of
    the train step), so the backward is issued here under the same flag."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, groups):
        w = w.to(x.dtype)
        with torch.backends.cudnn.flags(enabled=False):
            y = F.conv2d(x, w, None, stride, padding, 1, groups)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, groups)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, groups = ctx.cfg
        pad = padding if isinstance(padding, (tuple, list)) else (padding, padding)
        with torch.backends.cudnn.flags(enabled=False):
            gx, gw, _ = torch.ops.aten.convolution_backward(
                gy.contiguous(), x, w, None, list(stride), list(pad), [1, 1], False, [0, 0], groups,
                [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return gx, (gw.float() if gw is not None else None), None, None, None


class _CachedCast(torch.autograd.Function):
    """Identity-with-cast whose forward returns an up-to-date low-precision copy kept elsewhere (one multi-tensor cast
    per model forward instead of one tiny cast kernel per parameter); the backward is the ordinary cast back."""

    @staticmethod
    def forward(ctx, master, cached):
        ctx.dtype = master.dtype
        return cached.view_as(cached)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype), None


# MOMA_WCACHE=1 (default): under bf16 autocast on the GPU a model forward first refreshes bf16 copies of all its
# convolution weights / biases with ONE multi-tensor copy (autocast otherwise launches a cast kernel per parameter and
# forward: ~340 launches of ~4 us per train step) and the convolutions take the copies.
_WCACHE = os.environ.get("MOMA_WCACHE", "1") == "1"


class SamePadConv2d(nn.Conv2d):
    """Conv2d with TensorFlow 'SAME' padding computed from the input size at call time."""

    def __init__(self, cin, cout, kernel_size, stride=1, groups=1, bias=True):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=0, groups=groups, bias=bias)
        self._wc = self._bc = None                  # bf16 copies, valid for the model forward that refreshed them
        self._wc_live = False

    def _params(self, x):
        """(weight, bias) for this call: the refreshed bf16 copies when the enclosing model forward provided them."""
        if self._wc_live and x.is_cuda and x.dtype == torch.bfloat16:
            track = torch.is_grad_enabled()
            w = _CachedCast.apply(self.weight, self._wc) if (track and self.weight.requires_grad) else self._wc
            b = None
            if self.bias is not None:
                b = _CachedCast.apply(self.bias, self._bc) if (track and self.bias.requires_grad) else self._bc
            return w, b
        return self.weight, self.bias

    def forward(self, x):
        ih, iw = x.shape[-2:]
        kh, kw = self.kernel_size
        sh, sw = self.stride
        ph = max((math.ceil(ih / sh) - 1) * sh + kh - ih, 0)
        pw = max((math.ceil(iw / sw) - 1) * sw + kw - iw, 0)
        depthwise = self.groups > 1 and self.groups == self.in_channels == self.out_channels and self.bias is None
        if depthwise and x.is_cuda and _DW_MODE == "hip" and kh == kw and sh == sw and self.dilation == (1, 1):
            from .. import ops
            if ops.dwconv_supported(kh, sh):
                if torch.is_autocast_enabled():
                    x = x.to(torch.get_autocast_dtype("cuda"))
                if x.dtype in (torch.float32, torch.bfloat16):
                    # (the depthwise kernels take the fp32 master weights directly: no cast at all)
                    return ops.dwconv(x, self.weight, sh, ph // 2, pw // 2, math.ceil(ih / sh), math.ceil(iw / sw))
        pad = (ph // 2, pw // 2)
        if ph % 2 or pw % 2:
            x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
            pad = 0
        if depthwise and x.is_cuda and _DW_MODE in ("hip", "aten"):
            if torch.is_autocast_enabled():
                x = x.to(torch.get_autocast_dtype("cuda"))
            return _DepthwiseNative.apply(x, self.weight, self.stride, pad, self.groups)
        if self._wc_live and torch.is_autocast_enabled() and x.is_cuda and x.dtype != torch.bfloat16 and \
                torch.get_autocast_dtype("cuda") == torch.bfloat16:
            x = x.to(torch.bfloat16)
        w, b = self._params(x)
        return F.conv2d(x, w, b, self.stride, pad, self.dilation, self.groups)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers / state-dict keys) with an optional fused activation.
    On the GPU the pair runs on the library's kernels (`ops.bn_act`); elsewhere it is BN followed by the activation."""

    _nbt_pending = 0       # forward calls not yet added to the `num_batches_tracked` buffer (a 1-element GPU add per
                           # BN call otherwise: 144 launches per step); flushed whenever the buffer is read out

    def _flush_nbt(self):
        if self._nbt_pending and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._nbt_pending)
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_nbt()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, act=None, want_mean=False):
        """want_mean: also return the per-plane mean [N,C,1,1] of the result (fused into the apply pass on the GPU)."""
        if _BN_MODE == "hip" and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.momentum is not None:
            from .. import ops
            use_batch = self.training or self.running_mean is None
            if self.training and self.track_running_stats and self.num_batches_tracked is not None:
                self._nbt_pending += 1
            return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, use_batch,
                              self.momentum, self.eps, act, want_mean)
        if _BN_MODE == "aten" and x.is_cuda:
            with torch.backends.cudnn.flags(enabled=False):
                y = super().forward(x)
        else:
            y = super().forward(x)
        y = F.silu(y) if act == "silu" else (F.relu(y) if act == "relu" else y)
        return (y, F.adaptive_avg_pool2d(y, 1)) if want_mean else y


def _drop_connect(x, p, training):
    if not training or not p:
        return x
    keep = 1.0 - p
    mask = torch.floor(keep + torch.rand([x.shape[0], 1, 1, 1], dtype=x.dtype, device=x.device))
    return x / keep * mask


class MBConvBlock(nn.Module):
    def __init__(self, kernel, stride, expand, cin, cout, se_ratio):
        super().__init__()
        mid = cin * expand
        self.expand, self.stride, self.cin, self.cout = expand, stride, cin, cout
        if expand != 1:
            self._expand_conv = SamePadConv2d(cin, mid, 1, bias=False)
            self._bn0 = BatchNorm2d(mid, momentum=_BN_MOM, eps=_BN_EPS)
        self._depthwise_conv = SamePadConv2d(mid, mid, kernel, stride=stride, groups=mid, bias=False)
        self._bn1 = BatchNorm2d(mid, momentum=_BN_MOM, eps=_BN_EPS)
        sq = max(1, int(cin * se_ratio))
        self._se_reduce = SamePadConv2d(mid, sq, 1)
        self._se_expand = SamePadConv2d(sq, mid, 1)
        self._project_conv = SamePadConv2d(mid, cout, 1, bias=False)
        self._bn2 = BatchNorm2d(cout, momentum=_BN_MOM, eps=_BN_EPS)

    def forward(self, x, drop_connect_rate=None):
        inp = x
        if self.expand != 1:
            x = self._bn0(self._expand_conv(x), act="silu")
        x, pooled = self._bn1(self._depthwise_conv(x), act="silu", want_mean=True)      # squeeze fused into the BN pass
        s = self._se_expand(F.silu(self._se_reduce(pooled)))
        if _SE_MODE == "hip" and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
            from .. import ops
            x = ops.se_gate(x, s)
        else:
            x = torch.sigmoid(s) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            x = _drop_connect(x, drop_connect_rate, self.training) + inp
        return x


class EfficientNet(nn.Module):
    def __init__(self, num_classes=1000, dropout_rate=0.2, drop_connect_rate=0.2, in_channels=3):
        super().__init__()
        self.drop_connect_rate = drop_connect_rate
        self._conv_stem = SamePadConv2d(in_channels, 32, 3, stride=2, bias=False)
        self._bn0 = BatchNorm2d(32, momentum=_BN_MOM, eps=_BN_EPS)
        blocks = []
        for rep, k, s, e, cin, cout, se in _B0_STAGES:
            blocks.append(MBConvBlock(k, s, e, cin, cout, se))
            blocks += [MBConvBlock(k, 1, e, cout, cout, se) for _ in range(rep - 1)]
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = SamePadConv2d(320, 1280, 1, bias=False)
        self._bn1 = BatchNorm2d(1280, momentum=_BN_MOM, eps=_BN_EPS)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self.classifier_ = nn.Sequential(nn.Dropout(dropout_rate), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)

    def extract_endpoints(self, x):
        feats = []
        x = self._bn0(self._conv_stem(x), act="silu")
        prev = x
        nb = len(self._blocks)
        for i, blk in enumerate(self._blocks):
            rate = self.drop_connect_rate * float(i) / nb if self.drop_connect_rate else self.drop_connect_rate
            x = blk(x, drop_connect_rate=rate)
            if prev.size(2) > x.size(2):
                feats.append(prev)
            prev = x
        feats.append(self._bn1(self._conv_head(x), act="silu"))
        return feats

    def get_feat_modules(self):
        return nn.ModuleList([self._conv_stem, self._bn0, self._blocks, self._conv_head, self._bn1, self.classifier_])

    def _refresh_weight_cache(self, x):
        """One multi-tensor fp32 -> bf16 copy of every non-depthwise conv weight / bias (see _WCACHE)."""
        live = (_WCACHE and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        convs = getattr(self, "_wcache_convs", None)
        if convs is None:
            convs = [m for m in self.modules() if isinstance(m, SamePadConv2d) and not (m.groups > 1 and _DW_MODE == "hip")]
            object.__setattr__(self, "_wcache_convs", convs)
        if not live:
            for m in convs:
                m._wc_live = False
            return
        src, dst = [], []
        for m in convs:
            if m._wc is None or m._wc.device != m.weight.device:
                m._wc = torch.empty_like(m.weight, dtype=torch.bfloat16)
                m._bc = torch.empty_like(m.bias, dtype=torch.bfloat16) if m.bias is not None else None
            src.append(m.weight.detach()); dst.append(m._wc)
            if m.bias is not None:
                src.append(m.bias.detach()); dst.append(m._bc)
            m._wc_live = True
        with torch.no_grad():
            torch._foreach_copy_(dst, src)

    def forward(self, x, is_feat=False):
        self._refresh_weight_cache(x)
        out = self.extract_endpoints(x)
        pooled = self._avg_pooling(out[-1])
        out.append(pooled)
        logits = self.classifier_(pooled.flatten(start_dim=1))
        for m in self._wcache_convs:                 # the copies are only valid inside this forward
            m._wc_live = False
        return (out, logits) if is_feat else logits


def efficientnet_b0(num_classes=1000, **kw):
    return EfficientNet(num_classes=num_classes, **kw)
