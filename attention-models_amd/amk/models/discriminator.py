"""PatchGAN discriminator used by the ViT-VQGAN train step (reference:
models/utils/discriminator.py:6-54, constructed as NLayerDiscriminator(3, 64, 3) in
trainers/vitgqgan.py:64).  Not a kernel target: convolutions / BatchNorm stay on MIOpen.
It exists here because the benchmark step contains it; state_dict
keys (``model.N.*``) match the reference layer order.

The gradient penalty of that step (trainers/vitgqgan.py:115-131) differentiates the
discriminator's input gradient a second time.  ATen expresses the second derivative of a
convolution's data gradient with respect to the weight as a *forward* convolution with batch and
channel axes swapped -- for the first layer that is a convolution with a 128 x 128 kernel, which
MIOpen runs at 5.4 ms (the four such calls cost 9 ms of a 121 ms step).  ``Conv2d`` below states
the same derivatives as the primitives they are: a data-gradient node whose backward is one
forward convolution plus one ordinary weight-gradient call.  Same arithmetic, vendor kernels
throughout, shapes MIOpen has tuned kernels for.
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

_INPUT_GRAD_ONLY = False


@contextlib.contextmanager
def input_grad_only():
    """Inside: the backward of ``Conv2d`` skips weight / bias gradients.  For
    ``torch.autograd.grad(outputs, [input], create_graph=True)`` (the gradient penalty), where they
    are not asked for -- a custom Function cannot see that by itself and would compute them."""
    global _INPUT_GRAD_ONLY
    old, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRAD_ONLY = old


class _ConvDataGrad(torch.autograd.Function):
    """gx = conv_transpose-like data gradient of y = conv(x, W) for a given gy; differentiable in gy and W."""

    @staticmethod
    def forward(ctx, gy, weight, x_shape, stride, padding):
        ctx.save_for_backward(gy, weight)
        ctx.cfg = (stride, padding)
        return conv2d_input(x_shape, weight, gy, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, ggx):
        gy, weight = ctx.saved_tensors
        stride, padding = ctx.cfg
        ggx = ggx.contiguous()
        ggy = F.conv2d(ggx, weight, None, stride, padding) if ctx.needs_input_grad[0] else None
        gw = conv2d_weight(ggx, weight.shape, gy, stride=stride, padding=padding) if ctx.needs_input_grad[1] else None
        return ggy, gw, None, None, None


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, bias is not None)
        return F.conv2d(x, weight, bias, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = _ConvDataGrad.apply(gy, weight, x.shape, stride, padding) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if not _INPUT_GRAD_ONLY:
            if ctx.needs_input_grad[1]:
                gw = conv2d_weight(x, weight.shape, gy, stride=stride, padding=padding)
            if has_bias and ctx.needs_input_grad[2]:
                gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict) with the derivative structure described above."""

    def forward(self, x):
        if self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != "zeros" or isinstance(self.padding, str):
            return super().forward(x)
        if torch.is_autocast_enabled():
            if self.in_channels < 8 and x.is_cuda:
                # the image layer (3 input channels) stays in f32: MIOpen's bf16 weight-gradient kernels for such a layer
                # returned NaN from finite operands at small image sizes (tools/train_sanity.py bf16: step 3, only
                # model.0.weight.grad non-finite); its cost is negligible (memory-bound, 0.2 % of the step's FLOP)
                with torch.autocast("cuda", enabled=False):
                    return _Conv.apply(x.float(), self.weight, self.bias, self.stride, self.padding)
            return super().forward(x)  # mixed precision: ATen's own convolution autograd (dtype handling included)
        return _Conv.apply(x, self.weight, self.bias, self.stride, self.padding)


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=64, n_layers=3):
        super().__init__()
        widths = [ndf * min(2 ** i, 8) for i in range(n_layers + 1)]   # 64, 128, 256, 512
        layers = [Conv2d(input_nc, widths[0], 4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        for i in range(1, n_layers + 1):
            stride = 2 if i < n_layers else 1
            layers += [
                Conv2d(widths[i - 1], widths[i], 4, stride=stride, padding=1, bias=False),
                nn.BatchNorm2d(widths[i]),
                nn.LeakyReLU(0.2, True),
            ]
        layers.append(Conv2d(widths[-1], 1, 4, stride=1, padding=1))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)
