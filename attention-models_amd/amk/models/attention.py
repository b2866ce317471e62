"""Drop-in attention blocks: the reference's class names, constructor arguments, forward
signatures and state_dict keys, with the score / softmax / PV arithmetic in libamk.so.

SoftmaxAttention mirrors /root/reference/models/softmax_attention.py:22-82.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .layers import Linear
from .moe import StackedExpertsMixin, linear_like_init_


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
        self.q = nn.Sequential(Linear(dim, inner, bias=False))
        self.kv = nn.Sequential(Linear(dim, 2 * inner, bias=False))
        self.W_o = Linear(inner, dim)
        self.dropout_p = float(dropout)
        self.scale = dim_head ** -0.5

    def _drop(self, t):
        return F.dropout(t, self.dropout_p, self.training) if self.dropout_p > 0.0 else t

    def forward(self, x, context=None, causal_mask=None, context_mask=None):
        src = x if context is None else context
        if context is None and x.is_cuda:
            q, kv = ops.linear2(x, self.q[0].weight, self.kv[0].weight)  # self-attention: both projections in one launch
            q, kv = self._drop(q), self._drop(kv)
        else:
            q = self._drop(self.q(x))        # (B, I, h*d), consumed in place by the kernel
            kv = self._drop(self.kv(src))    # (B, J, 2*h*d): '(kv h d)' columns
        o = ops.attention_fused_kv(
            q, kv, self.num_heads, self.dim_head, self.scale,
            key_mask=context_mask, causal_mask=causal_mask,
        )                                # (B, I, h*d) == 'b h t d -> b t (h d)'
        return self._drop(self.W_o(o))


class SwitchHeadAttention(StackedExpertsMixin, nn.Module):
    """SwitchHead attention (reference: models/switchhead_attention.py:18-116).

    Dense Q and K projections; the V projection and the output projection are mixtures of E
    experts routed PER (token, head): ``moe_v`` weights its top-k experts with
    sigmoid(W_s logits); ``moe_out`` routes with the W_d logits but sums its experts
    UN-weighted, so W_d receives no gradient (reference :80-87).
    The experts are shared by all heads; heads are summed at the end.  No output bias.

    state_dict keys: q.0.weight, k.0.weight, W_s.0.weight, W_d.0.weight,
    experts_v.{e}.weight (d, dim), experts_out.{e}.weight (dim, d).
    """

    _stacked = {"experts_v_weight": ("experts_v", "weight"), "experts_out_weight": ("experts_out", "weight")}

    def __init__(self, dim, num_heads=8, dim_head=64, num_experts=5, sel_experts=2, dropout=0.0):
        super().__init__()
        self.dim, self.num_heads, self.dim_head = dim, num_heads, dim_head
        self.num_experts, self.sel_experts = num_experts, sel_experts
        inner = num_heads * dim_head
        self.q = nn.Sequential(nn.Linear(dim, inner, bias=False))
        self.k = nn.Sequential(nn.Linear(dim, inner, bias=False))
        self.W_s = nn.Sequential(nn.Linear(dim, num_heads * num_experts, bias=False))
        self.experts_v_weight = nn.Parameter(torch.empty(num_experts, dim_head, dim))
        self.W_d = nn.Sequential(nn.Linear(dim, num_heads * num_experts, bias=False))
        self.experts_out_weight = nn.Parameter(torch.empty(num_experts, dim, dim_head))
        linear_like_init_(self.experts_v_weight)
        linear_like_init_(self.experts_out_weight)
        self.dropout_p = float(dropout)
        self.scale = dim_head ** -0.5
        self.inf = -1e9
        self.last_selected_v = None
        self.last_selected_out = None

    def _drop(self, t):
        return F.dropout(t, self.dropout_p, self.training) if self.dropout_p > 0.0 else t

    def forward(self, x, context=None, causal_mask=None, context_mask=None):
        B, I, _ = x.shape
        src = x if context is None else context
        J = src.shape[1]
        if J != I:
            # the reference indexes moe_out rows by the CONTEXT positions and returns (B, J, dim)
            # for J != I; no caller uses it (models/vit_moe.py:34 is self).
            raise NotImplementedError("SwitchHeadAttention with a context of another length is a reference quirk, not supported")
        h, d, E, k = self.num_heads, self.dim_head, self.num_experts, self.sel_experts
        if context is None and not (self.dropout_p > 0.0 and self.training):
            # self-attention: q, k and the V experts' gate projection read the same rows -> one GEMM over the
            # stacked weights (three launches and three passes over x otherwise, each way).  W_d only selects
            # experts (top-k indices), so it stays without a gradient, as in the reference -- and out of the stacked
            # GEMM, whose backward would otherwise spend a sixth of its two products on columns of zeros.
            W = torch.cat([self.q[0].weight, self.k[0].weight, self.W_s[0].weight], 0)
            q2, k2, gate_s = F.linear(x, W).split([h * d, h * d, h * E], dim=-1)
            with torch.no_grad():
                gate_d = F.linear(x, self.W_d[0].weight)
            q = q2.view(B, I, h, d).permute(0, 2, 1, 3)       # strided views of the GEMM output, no copies
            kk = k2.view(B, J, h, d).permute(0, 2, 1, 3)
        else:
            q = self._drop(self.q(x)).view(B, I, h, d).permute(0, 2, 1, 3)
            kk = self._drop(self.k(src)).view(B, J, h, d).permute(0, 2, 1, 3)
            gate_s, gate_d = self.W_s(src), self.W_d(src)
        src2 = src.reshape(B * J, self.dim)
        # where a token's h*k pairs cover at least half of the experts, an expert's product is formed once per distinct
        # (token, expert) and the head / slot sums are dense products over per-expert sums (ops._SharedRowExperts)
        distinct = ops.distinct_experts_ok(self.dim, d, h * k, E, src2, self.experts_v_weight, self.experts_out_weight)
        # moe_v: unit = (b, t, head); every unit reads the token's full input row
        if distinct:
            v, sel_v = ops.shared_row_experts(src2, gate_s.reshape(B * J * h, E), self.experts_v_weight, k, h)
        else:
            v, sel_v = ops.routed_linear(src2, gate_s.reshape(B * J * h, E), self.experts_v_weight, None,
                                         k, x_div=h * k, weighted=True, outer=1)
        v = v.view(B, J, h, d).permute(0, 2, 1, 3)
        o = ops.attention(q, kk, v, self.scale, key_mask=context_mask, causal_mask=causal_mask)  # (B,h,I,d)
        o2 = o.permute(0, 2, 1, 3).reshape(B * I * h, d)  # 'b i h d' rows, contiguous by construction
        # moe_out: routed by W_d(gate_inputs = src), experts summed un-weighted, then summed over heads
        if distinct:
            out, sel_o = ops.summed_experts(o2, gate_d.reshape(B * J * h, E), self.experts_out_weight, k, h)
        else:
            out, sel_o = ops.routed_linear(o2, gate_d.reshape(B * J * h, E), self.experts_out_weight, None,
                                           k, x_div=k, weighted=False, outer=h)
        self.last_selected_v = sel_v.view(B, J, h, k)
        self.last_selected_out = sel_o.view(B, J, h, k)
        return out.view(B, I, self.dim)


class AgentAttention(nn.Module):
    """Agent attention (reference: models/agent_attention.py:21-80).

    ``pool_size = int(agent_num ** 0.5)`` agent tokens per head are the adaptive average pool of
    q over the sequence; the reference pools the (t, h) plane to (pool, pool) and therefore only
    runs when ``num_heads == pool_size`` -- enforced here at construction.
    ``context_mask`` is accepted and ignored, as in the reference; there is no cross mode.

    state_dict keys: qkv.weight, W_o.{weight,bias}, bias1, bias2, dwc.1.{weight,bias}.
    bias1 / bias2 are scalars added to entire softmax rows: they cannot change the output and
    get (exactly) zero gradient; they are kept only for checkpoint compatibility.
    """

    def __init__(self, dim, num_heads=8, dim_head=64, agent_num=47, dropout=0.0):
        super().__init__()
        self.dim, self.num_heads, self.dim_head = dim, num_heads, dim_head
        self.pool_size = int(agent_num ** 0.5)
        if self.pool_size != num_heads:
            raise ValueError(
                f"AgentAttention needs num_heads == int(agent_num ** 0.5) (got {num_heads} vs {self.pool_size}): "
                "the reference's einsum fails otherwise (models/agent_attention.py:56-60)")
        inner = num_heads * dim_head
        self.qkv = Linear(dim, 3 * inner, bias=False)   # layers.Linear: nn.Linear's parameters, ops.linear's kernels
        self.scale = dim_head ** -0.5
        self.W_o = Linear(inner, dim)
        self.dropout_p = float(dropout)
        self.bias1 = nn.Parameter(torch.zeros(1, 1, 1, 1))
        self.bias2 = nn.Parameter(torch.zeros(1, 1, 1, 1))
        # index 1 of a Sequential in the reference (index 0 / 2 are einops Rearrange layers)
        self.dwc = nn.Sequential(nn.Identity(), nn.Conv2d(dim_head, dim_head, 3, padding=1, groups=dim_head), nn.Identity())

    def forward(self, x, context_mask=None):
        conv = self.dwc[1]
        o = ops.agent_attention(self.qkv(x), conv.weight, conv.bias, self.num_heads, self.dim_head,
                                self.pool_size, self.scale)
        o = self.W_o(o)
        return F.dropout(o, self.dropout_p, self.training) if self.dropout_p > 0.0 else o
