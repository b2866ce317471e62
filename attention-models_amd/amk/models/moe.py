"""MoELayer with the reference's API (models/moe.py:14-38) on the grouped-GEMM kernels.

The reference keeps one nn.Linear per expert and walks them in a Python loop.  Here the
expert weights are ONE (E, out, in) parameter (a single contiguous HBM slab the grouped GEMM
streams once) and the state_dict is translated to / from the reference's
``experts.{e}.weight`` / ``experts.{e}.bias`` keys, so reference checkpoints load unchanged.
"""
import math

import torch
import torch.nn as nn

from .. import ops


class StackedExpertsMixin:
    """state_dict translation between stacked (E, ...) parameters and per-expert keys.
    ``_stacked`` maps parameter name -> (module list name, leaf name)."""

    _stacked = {}

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        for pname, (mod, leaf) in self._stacked.items():
            t = destination.pop(prefix + pname, None)
            if t is None:
                continue
            for e in range(t.shape[0]):
                destination[f"{prefix}{mod}.{e}.{leaf}"] = t[e]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for pname, (mod, leaf) in self._stacked.items():
            n = getattr(self, pname).shape[0]
            keys = [f"{prefix}{mod}.{e}.{leaf}" for e in range(n)]
            if all(kk in state_dict for kk in keys):
                state_dict[prefix + pname] = torch.stack([state_dict.pop(kk) for kk in keys])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


def linear_like_init_(weight, bias=None):
    """nn.Linear's default init applied per expert slice of a stacked (E, out, in) tensor."""
    fan_in = weight.shape[-1]
    bound = 1.0 / math.sqrt(fan_in)
    with torch.no_grad():
        weight.uniform_(-bound, bound)  # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        if bias is not None:
            bias.uniform_(-bound, bound)


class MoELayer(StackedExpertsMixin, nn.Module):
    """out[b,t] = sum over the token's top-k experts (ascending expert id) of
    sigmoid(gate logit) * (x W_e^T + b_e).  Weights are sigmoid of the selected logits, not a
    softmax, and are not renormalised (reference lines 27-29)."""

    _stacked = {"experts_weight": ("experts", "weight"), "experts_bias": ("experts", "bias")}

    def __init__(self, input_dim, output_dim, num_experts, sel_experts):
        super().__init__()
        if output_dim != input_dim:
            # the reference allocates results as (b, t, input_dim): other widths fail there too
            raise ValueError("MoELayer requires output_dim == input_dim (reference models/moe.py:31)")
        self.sel_experts = sel_experts
        self.num_experts = num_experts
        self.gate = nn.Linear(input_dim, num_experts)
        self.experts_weight = nn.Parameter(torch.empty(num_experts, output_dim, input_dim))
        self.experts_bias = nn.Parameter(torch.empty(num_experts, output_dim))
        linear_like_init_(self.experts_weight, self.experts_bias)
        self.last_selected_experts = None

    def forward(self, inputs):
        B, T, D = inputs.shape
        x2 = inputs.reshape(B * T, D)
        logits = self.gate(inputs).reshape(B * T, self.num_experts)
        out, ids = ops.routed_linear(x2, logits, self.experts_weight, self.experts_bias,
                                     self.sel_experts, x_div=self.sel_experts, weighted=True, outer=1)
        self.last_selected_experts = ids.view(B, T, self.sel_experts)
        return out.view(B, T, -1)
