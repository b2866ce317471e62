"""ViT-VQGAN with the codebook lookup and the attention cores on libamk.so.

Class / parameter names follow /root/reference/models/vitvqgan.py so that reference
checkpoints load (``load_state_dict(strict=False)`` as the reference's callers do):
``encoder.to_patch_embedding.{1,2,3}``, ``encoder.pos_enc``, ``encoder.encoder.layers.N.*``,
``pre_quant``, ``codebook.embedding.weight``, ``post_quant``, ``decoder.*``.

Build decision: the reference's FeedForward (vitvqgan.py:20-34)
cannot be constructed at HEAD -- it calls ``super().__init__(in_features=..., hidden_features=...,
out_features=..., bias=...)`` on ``object``.  Those keyword names and the
``(int(hidden*2/3)+7)//8*8`` rounding are the signature of an xformers-style SwiGLU base, so
the FFN here is SwiGLU: ``w12 = Linear(in, 2*hidden)``, ``w3 = Linear(hidden, out)``,
``w3(silu(a) * b)``.  Attention and VQ parity do not depend on it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .attention import SoftmaxAttention
from .layers import LayerNorm, Linear


def l2_norm(x):
    return F.normalize(x, p=2, dim=-1)


class SwiGLU(nn.Module):
    def __init__(self, in_features, hidden_features, out_features, bias=True):
        super().__init__()
        self.w12 = Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        if not x.is_cuda:
            a, b = self.w12(x).chunk(2, dim=-1)
            return self.w3(F.silu(a) * b)
        # the gate in the w12 GEMM's epilogue, its derivative in the w3 input gradient's (amk_gemm_f32)
        return ops.swiglu_ffn(x, self.w12.weight, self.w12.bias, self.w3.weight, self.w3.bias)


class FeedForward(SwiGLU):
    def __init__(self, in_features, hidden_features=None, bias=True):
        hidden = (int(hidden_features * 2 / 3) + 7) // 8 * 8
        super().__init__(in_features, hidden, in_features, bias=bias)


class EncoderLayer(nn.Module):
    """Pre-LN block: x + attn(LN(x)); x + ffn(LN(x)) (reference lines 38-61)."""

    def __init__(self, dim, n_heads, d_head, mlp_dim, dropout):
        super().__init__()
        self.self_attn = SoftmaxAttention(dim, n_heads, d_head, dropout)
        self.feed_forward = FeedForward(dim, mlp_dim)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)

    def forward(self, x):
        x = x + self.self_attn(self.norm1(x))
        return x + self.feed_forward(self.norm2(x))

    def forward_fused(self, h, pending):
        """The same block on a residual stream kept as (h, pending): the stream's value is h + pending
        and each add is done inside the LayerNorm kernel that consumes it (amk_add_layernorm_fwd).
        Returns the new (h, pending)."""
        if pending is None:
            y = self.norm1(h, branch=True)
        else:
            h, y = self.norm1(pending, h, branch=True)
        h, y = self.norm2(self.self_attn(y), h, branch=True)
        return h, self.feed_forward(y)


class TransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, depth, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList(EncoderLayer(dim, n_heads, d_head, mlp_dim, dropout) for _ in range(depth))

    def forward(self, x):
        h, pending = x, None
        for blk in self.layers:
            h, pending = blk.forward_fused(h, pending)
        return h if pending is None else h + pending


class Patchify(nn.Module):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' without einops."""

    def __init__(self, patch):
        super().__init__()
        self.patch = patch

    def forward(self, img):
        B, Cc, Hh, Ww = img.shape
        p = self.patch
        t = img.view(B, Cc, Hh // p, p, Ww // p, p)
        return t.permute(0, 2, 4, 3, 5, 1).reshape(B, (Hh // p) * (Ww // p), p * p * Cc)


def unpatchify(tokens, patch, grid):
    """'b (h w) (p1 p2 c) -> b c (h p1) (w p2)'."""
    B = tokens.shape[0]
    t = tokens.view(B, grid, grid, patch, patch, -1)
    return t.permute(0, 5, 1, 3, 2, 4).reshape(B, -1, grid * patch, grid * patch)


class ViTEncoder(nn.Module):
    def __init__(self, dim, img_size, patch_size, n_heads, d_head, depth, mlp_dim, dropout):
        super().__init__()
        self.dim = dim
        self.patch_size = patch_size
        self.img_size = img_size
        patch_dim = 3 * patch_size * patch_size
        n_patches = (img_size // patch_size) ** 2
        self.to_patch_embedding = nn.Sequential(
            Patchify(patch_size), LayerNorm(patch_dim), Linear(patch_dim, dim), LayerNorm(dim)
        )
        self.pos_enc = nn.Parameter(torch.randn(1, n_patches, dim))
        self.pre_norm = LayerNorm(dim)
        self.encoder = TransformerBlock(dim, n_heads, d_head, depth, mlp_dim, dropout)

    def forward(self, img):
        tok = self.to_patch_embedding(img)
        return self.encoder(self.pre_norm(self.pos_enc + tok))


class ViTDecoder(nn.Module):
    def __init__(self, dim, img_size, patch_size, n_heads, d_head, depth, mlp_dim, dropout):
        super().__init__()
        self.patch_size = patch_size
        self.img_size = img_size
        patch_dim = 3 * patch_size * patch_size
        n_patches = (img_size // patch_size) ** 2
        self.pos_enc = nn.Parameter(torch.randn(1, n_patches, dim))
        self.pre_norm = LayerNorm(dim)
        self.decoder = TransformerBlock(dim, n_heads, d_head, depth, mlp_dim, dropout)
        self.fc = Linear(dim, patch_dim)

    def forward(self, tok):
        tok = self.decoder(self.pre_norm(tok + self.pos_enc))
        return unpatchify(self.fc(tok), self.patch_size, self.img_size // self.patch_size)


class Codebook(nn.Module):
    """VQ nearest-neighbour lookup on l2-normalised vectors (reference lines 140-176).

    forward(z) -> (z_q with straight-through gradient, int64 indices, loss) where
    loss = beta*mean((z_q.detach()-z)^2) + mean((z_q-z.detach())^2) on the normalised z.
    """

    def __init__(self, codebook_size=8192, codebook_dim=32, beta=0.25):
        super().__init__()
        self.codebook_size = codebook_size
        self.codebook_dim = codebook_dim
        self.beta = beta
        self.embedding = nn.Embedding(codebook_size, codebook_dim)
        self.embedding.weight.data.normal_()

    def forward(self, z):
        return ops.vq_lookup(z, self.embedding.weight, self.beta)

    def indices_to_embeddings(self, indices):
        return ops.vq_gather(indices, self.embedding.weight)


class ViTVQGAN(nn.Module):
    def __init__(self, vit_params, codebook_params):
        super().__init__()
        self.encoder = ViTEncoder(**vit_params)
        self.pre_quant = Linear(vit_params["dim"], codebook_params["codebook_dim"])
        self.codebook = Codebook(**codebook_params)
        self.post_quant = Linear(codebook_params["codebook_dim"], vit_params["dim"])
        self.decoder = ViTDecoder(**vit_params)

    def forward(self, imgs):
        z = self.pre_quant(self.encoder(imgs))
        z_q, _indices, loss = self.codebook(z)
        return self.decoder(self.post_quant(z_q)), loss

    def decode_indices(self, indices):
        return self.decoder(self.post_quant(self.codebook.indices_to_embeddings(indices)))

    def encode_imgs(self, imgs):
        z = self.pre_quant(self.encoder(imgs))
        return self.codebook(z)[1]  # int64 (B, T)

    @property
    def num_patches(self):
        return (self.encoder.img_size // self.encoder.patch_size) ** 2
