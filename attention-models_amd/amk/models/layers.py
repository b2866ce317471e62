"""nn.Linear / nn.LayerNorm with the same parameters and state_dict keys, routed through libamk.so when the input is on
the GPU: Linear through ops.linear -- the own exact-f32 MFMA GEMMs of csrc/gemm_f32.hip (amk_gemm_f32: forward, weight +
bias gradient in one pass) or the vendor GEMM, as ops.DENSE_MODE decides per shape; bf16 kernels under autocast --,
LayerNorm through amk_add_layernorm_* (residual add and normalisation in one kernel)."""
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Linear(nn.Linear):
    def forward(self, x):
        if not x.is_cuda:
            return F.linear(x, self.weight, self.bias)
        return ops.linear(x, self.weight, self.bias)   # (ops.DENSE_MODE / ops.GEMM_MODE pick the GEMM)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm over the last axis.  ``forward(x)`` = LN(x); ``forward(x, residual)`` returns
    ``(x + residual, LN(x + residual))`` from one kernel.  ``branch=True``: the normalised output only feeds Linear
    layers (the pre-LN blocks) -- under bf16 autocast it is then written in bf16 by the same kernel."""

    def forward(self, x, residual=None, branch=False):
        fusable = x.is_cuda and self.elementwise_affine and self.bias is not None and len(self.normalized_shape) == 1
        if residual is None:
            if not fusable:
                return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
            return ops.layer_norm(x, self.weight, self.bias, self.eps, branch=branch)
        if not fusable:
            h = x + residual
            return h, F.layer_norm(h, self.normalized_shape, self.weight, self.bias, self.eps)
        return ops.add_layer_norm(x, residual, self.weight, self.bias, self.eps, branch=branch)
