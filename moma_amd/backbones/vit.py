"""Minimal Vision Transformers with the distillation feature contract.

The reference ships ViT definitions that need `timm` and expose neither `forward(x, is_feat=True)` nor
`get_feat_modules()` (no runnable reference backbone for BASELINE configs 3 and 5), so this is the
build's own definition: pre-norm blocks, class token, learned position embedding (bicubic-resized when the input
size differs from the construction size), `model(x, is_feat=True) -> ([patch tokens, block outputs.., cls feature],
logits)`; the MoMA loop uses the last entry ([B, embed_dim]).  Parameter names follow the common timm layout
(patch_embed.proj, cls_token, pos_embed, blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm, head) so
a locally saved checkpoint loads by path.  Token attention inside the backbone is torch SDPA (backbones are out of
the hand-written scope); the KD-term batch-token attention is the K1 kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        q, k, v = self.qkv(x).view(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v)
        return self.proj(y.transpose(1, 2).reshape(b, n, c))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads, ratio):
        super().__init__()
        self.norm1, self.attn = nn.LayerNorm(dim, eps=1e-6), _Attn(dim, heads)
        self.norm2, self.mlp = nn.LayerNorm(dim, eps=1e-6), _Mlp(dim, int(dim * ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch=16, dim=384, depth=12, heads=6, ratio=4.0, num_classes=1000):
        super().__init__()
        self.patch, self.grid, self.embed_dim = patch, img_size // patch, dim
        self.patch_embed = _PatchEmbed(patch, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + self.grid ** 2, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads, ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def _pos(self, gh, gw):
        if gh == self.grid and gw == self.grid:
            return self.pos_embed
        cls, grid = self.pos_embed[:, :1], self.pos_embed[:, 1:]
        grid = grid.reshape(1, self.grid, self.grid, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid.float(), size=(gh, gw), mode="bicubic", align_corners=False).to(cls.dtype)
        return torch.cat([cls, grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)

    def get_feat_modules(self):
        return nn.ModuleList([self.patch_embed, self.blocks, self.norm, self.head])

    def forward(self, x, is_feat=False):
        gh, gw = x.shape[-2] // self.patch, x.shape[-1] // self.patch
        t = self.patch_embed(x)
        feats = [t]
        t = torch.cat([self.cls_token.expand(t.shape[0], -1, -1).to(t.dtype), t], dim=1) + self._pos(gh, gw).to(t.dtype)
        every = max(1, len(self.blocks) // 4)
        for i, blk in enumerate(self.blocks):
            t = blk(t)
            if (i + 1) % every == 0:
                feats.append(t)
        cls = self.norm(t)[:, 0]
        feats.append(cls)
        logits = self.head(cls)
        return (feats, logits) if is_feat else logits


def _vit(dim, depth, heads, img):
    def make(num_classes=1000, **_):
        return VisionTransformer(img_size=img, dim=dim, depth=depth, heads=heads, num_classes=num_classes)
    return make


vit_tiny_patch16_224 = _vit(192, 12, 3, 224)
vit_small_patch16_224 = _vit(384, 12, 6, 224)
vit_base_patch16_224 = _vit(768, 12, 12, 224)
vit_tiny_patch16_384 = _vit(192, 12, 3, 384)
vit_base_patch16_384 = _vit(768, 12, 12, 384)
