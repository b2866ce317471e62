"""Drop-in attention blocks: the reference's class names, constructor arguments, forward
signatures and state_dict keys, with the score / softmax / PV arithmetic in libamk.so.

SoftmaxAttention mirrors /root/reference/models/softmax_attention.py:22-82.
"""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class SoftmaxAttention(nn.Module):
    """Multi-head softmax attention (self- or cross-, via ``context``).

    Parameters (checkpoint-compatible with the reference): ``q.0.weight (h*d, dim)``,
    ``kv.0.weight (2*h*d, dim)`` with K rows first then V, ``W_o.{weight,bias}``.
    Dropout is applied to the projection outputs and to the final output, never to the
    attention probabilities (reference lines 30-44, 81).
    """

    def __init__(self, dim, num_heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.dim_head = dim_head
        inner = num_heads * dim_head
        # Sequential containers only to keep the reference's "q.0.weight"/"kv.0.weight" keys.
        self.q = nn.Sequential(nn.Linear(dim, inner, bias=False))
        self.kv = nn.Sequential(nn.Linear(dim, 2 * inner, bias=False))
        self.W_o = nn.Linear(inner, dim)
        self.dropout_p = float(dropout)
        self.scale = dim_head ** -0.5

    def _drop(self, t):
        return F.dropout(t, self.dropout_p, self.training) if self.dropout_p > 0.0 else t

    def forward(self, x, context=None, causal_mask=None, context_mask=None):
        src = x if context is None else context
        q = self._drop(self.q(x))        # (B, I, h*d), consumed in place by the kernel
        kv = self._drop(self.kv(src))    # (B, J, 2*h*d): '(kv h d)' columns
        o = ops.attention_fused_kv(
            q, kv, self.num_heads, self.dim_head, self.scale,
            key_mask=context_mask, causal_mask=causal_mask,
        )                                # (B, I, h*d) == 'b h t d -> b t (h d)'
        return self._drop(self.W_o(o))
