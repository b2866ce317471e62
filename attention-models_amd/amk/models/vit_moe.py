"""ViT with SwitchHead attention and a top-k MoE FFN in every layer (reference:
models/vit_moe.py:10-107), BASELINE.json configs[3].  One ``n_experts`` feeds both the
SwitchHead experts and the MoE FFN, as in the reference (models/vit_moe.py:26-27)."""
import torch
import torch.nn as nn

from .layers import LayerNorm
from .attention import SwitchHeadAttention
from .moe import MoELayer
from .vitvqgan import Patchify


class EncoderLayer(nn.Module):
    def __init__(self, dim, n_heads, d_head, n_experts, sel_experts, dropout):
        super().__init__()
        self.self_attn = SwitchHeadAttention(dim, n_heads, d_head, num_experts=n_experts, sel_experts=sel_experts,
                                             dropout=dropout)
        self.moe = MoELayer(input_dim=dim, output_dim=dim, num_experts=n_experts, sel_experts=sel_experts)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)

    def forward(self, x, context_mask=None):
        x = self.self_attn(x=self.norm1(x), context_mask=context_mask) + x
        return self.moe(self.norm2(x)) + x


class Encoder(nn.Module):
    def __init__(self, dim, n_heads, d_head, depth, n_experts, sel_experts, dropout):
        super().__init__()
        self.layers = nn.ModuleList(EncoderLayer(dim, n_heads, d_head, n_experts, sel_experts, dropout)
                                    for _ in range(depth))

    def forward(self, x, context_mask=None):
        # the layers' arithmetic (EncoderLayer.forward) with every residual add inside the kernel of the LayerNorm that
        # follows it: (x + branch, LN(x + branch)) from one launch, and one launch for both in the backward
        layers = self.layers
        if len(layers) == 0:
            return x
        n = layers[0].norm1(x)
        for i, layer in enumerate(layers):
            a = layer.self_attn(x=n, context_mask=context_mask)
            x, n = layer.norm2(a, residual=x)
            m = layer.moe(n)
            if i + 1 < len(layers):
                x, n = layers[i + 1].norm1(m, residual=x)
            else:
                x = m + x
        return x


class ViTMoE(nn.Module):
    def __init__(self, dim=1024, image_size=256, patch_size=32, n_heads=16, d_head=64, depth=6, n_experts=32,
                 sel_experts=2, dropout=0.0, num_classes=1000):
        super().__init__()
        self.dim = dim
        self.patch_size = patch_size
        self.patch_dim = 3 * patch_size * patch_size
        self.to_patch_embedding = nn.Sequential(Patchify(patch_size), LayerNorm(self.patch_dim),
                                                nn.Linear(self.patch_dim, dim), LayerNorm(dim))
        self.class_token = nn.Parameter(torch.randn(1, 1, dim))
        n_patches = (image_size // patch_size) ** 2
        self.pos_enc = nn.Parameter(torch.randn(1, n_patches + 1, dim))
        self.encoder = Encoder(dim, n_heads, d_head, depth, n_experts, sel_experts, dropout)
        self.norm = LayerNorm(dim)
        self.class_embed = nn.Linear(dim, num_classes)

    def forward(self, x):
        tok = self.to_patch_embedding(x)
        cls = self.class_token.expand(tok.shape[0], -1, -1)
        tok = torch.cat([cls, tok], dim=1) + self.pos_enc
        return self.class_embed(self.norm(self.encoder(tok))[:, 0])
