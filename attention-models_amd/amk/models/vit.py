"""ViT classifier (reference: models/vit.py:24-69), BASELINE.json configs[1].

Reference quirk kept on purpose: ``ViT`` passes ``dropout`` into the
encoder's ``mult`` slot, so with dropout 0.0 every layer's FFN is Linear(dim, 0) / Linear(0, dim)
and contributes exactly 0; the GELU MLP it builds (``encoder.feed_forward``) is attached to the
encoder but never called.  Both exist here with the same parameter names so checkpoints and
parameter counts (33,629,160 at the config-2 size) match.
"""
import torch
import torch.nn as nn

from .layers import LayerNorm
from .transformer import Encoder
from .vitvqgan import Patchify


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class ViT(nn.Module):
    def __init__(self, dim, image_size=256, patch_size=16, n_heads=12, d_head=64, depth=12, mlp_dim=3072,
                 dropout=0.0, num_classes=None):
        super().__init__()
        self.dim = dim
        self.patch_size = patch_size
        self.patch_dim = 3 * patch_size * patch_size
        self.to_patch_embedding = nn.Sequential(Patchify(patch_size), LayerNorm(self.patch_dim),
                                                nn.Linear(self.patch_dim, dim), LayerNorm(dim))
        self.final_fc = nn.Linear(dim, num_classes)
        self.class_token = nn.Parameter(torch.randn(dim))
        n_patches = (image_size // patch_size) ** 2
        self.pos_enc = nn.Parameter(torch.randn(1, n_patches + 1, dim))
        self.encoder = Encoder(dim, n_heads, d_head, depth, dropout)      # dropout lands in `mult` (quirk)
        self.encoder.feed_forward = FeedForward(dim, mlp_dim)              # built, never called (quirk)

    def forward(self, x):
        tok = self.to_patch_embedding(x)
        cls = self.class_token.expand(tok.shape[0], 1, -1)
        tok = torch.cat([cls, tok], dim=1) + self.pos_enc
        return self.final_fc(self.encoder(tok)[:, 0])
