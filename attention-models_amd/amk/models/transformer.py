"""Pre-LN encoder / decoder stacks around SoftmaxAttention (scaffolding of SURVEY.md section 8
a15; reference: models/transformer.py:11-135).  Plain PyTorch except for the attention cores.
The seq2seq ``Transformer`` toy of the reference file is out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .attention import SoftmaxAttention


class LayerNorm(nn.Module):
    """Learnable gain, fixed zero shift kept as a BUFFER named ``beta`` (it is in the state_dict
    but never trained), as the reference has it."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x, residual=None):
        """LN(x), or with ``residual``: (x + residual, LN(x + residual)) from one kernel."""
        if residual is None:
            if x.is_cuda:
                return ops.layer_norm(x, self.gamma, self.beta)   # amk_add_layernorm_fwd / _bwd
            return F.layer_norm(x, x.shape[-1:], self.gamma, self.beta)
        if x.is_cuda:
            return ops.add_layer_norm(x, residual, self.gamma, self.beta)
        h = x + residual
        return h, F.layer_norm(h, h.shape[-1:], self.gamma, self.beta)


class GEGLU(nn.Module):
    def forward(self, x):
        if x.is_cuda and x.shape[-1] % 8 == 0 and x.numel() > 0:
            return ops.geglu(x)                     # amk_geglu_fwd / _bwd: one pass each
        val, gate = x.chunk(2, dim=-1)
        return gate * F.gelu(val)


class FeedForward(nn.Module):
    """Linear(dim, 2*inner, no bias) -> GEGLU -> LayerNorm(inner) -> Linear(inner, dim, no bias),
    inner = int(dim * mult * 2 / 3)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        inner = int(dim * mult * 2 / 3)
        self.ff = nn.Sequential(nn.Linear(dim, inner * 2, bias=False), GEGLU(), LayerNorm(inner),
                                nn.Linear(inner, dim, bias=False))

    def forward(self, x):
        return self.ff(x)


class EncoderLayer(nn.Module):
    def __init__(self, dim, n_heads=8, d_head=64, mult=4, dropout=0.0):
        super().__init__()
        self.self_attn = SoftmaxAttention(dim, n_heads, d_head, dropout)
        self.feed_forward = FeedForward(dim, mult=mult)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)

    def forward(self, x, context_mask=None):
        # x + attn(LN(x)), then x + ffn(LN(x)): the first add happens inside the second LayerNorm kernel
        x, y = self.norm2(x, self.self_attn(x=self.norm1(x), context_mask=context_mask))
        return self.feed_forward(y) + x


class Encoder(nn.Module):
    def __init__(self, dim, n_heads=8, d_head=64, depth=6, mult=4, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList(EncoderLayer(dim, n_heads, d_head, mult, dropout) for _ in range(depth))

    def forward(self, x, context_mask=None):
        for layer in self.layers:
            x = layer(x, context_mask=context_mask)
        return x


class DecoderLayer(nn.Module):
    """self-attention (causal mask) -> cross-attention over ``context`` (key-padding mask) -> FFN."""

    def __init__(self, dim, n_heads=8, d_head=64, mult=4, dropout=0.0):
        super().__init__()
        self.self_attn = SoftmaxAttention(dim, n_heads, d_head, dropout)
        self.cross_attn = SoftmaxAttention(dim, n_heads, d_head, dropout)
        self.feed_forward = FeedForward(dim, mult)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)

    def forward(self, dec_inp, context, context_mask=None, causal_mask=None):
        # the adds after self- and cross-attention happen inside the LayerNorm kernels that follow them
        x, y = self.norm2(dec_inp, self.self_attn(x=self.norm1(dec_inp), causal_mask=causal_mask))
        x, y = self.norm3(x, self.cross_attn(x=y, context=context, context_mask=context_mask))
        return self.feed_forward(y) + x


class Decoder(nn.Module):
    def __init__(self, dim, n_heads=8, d_head=64, depth=6, mult=4, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList(DecoderLayer(dim, n_heads, d_head, mult, dropout) for _ in range(depth))

    def forward(self, dec_in, context, context_mask=None, causal_mask=None):
        out = dec_in
        for layer in self.layers:
            out = layer(out, context, context_mask=context_mask, causal_mask=causal_mask)
        return out
