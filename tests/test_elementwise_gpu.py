"""Fused SwiGLU gate (amk_swiglu_fwd / _bwd through the C ABI) against plain PyTorch fp32 on the CPU
(the op sequence the oracle's FFN uses: chunk -> silu -> mul), values and gradients."""
import pytest
import torch
import torch.nn.functional as F

from oracle.fixture_recipe import seeded
from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 16, 2 * 96), (1, 1, 8), (3, 1024, 2 * 1368), (5, 7, 2 * 20)])
def test_swiglu_matches_reference(device, shape):
    from amk import ops

    ab = seeded(shape, 11 + shape[-1], 2.0)
    cot = seeded(shape[:-1] + (shape[-1] // 2,), 12 + shape[-1])
    ref_in = ab.clone().requires_grad_(True)
    a, b = ref_in.chunk(2, dim=-1)
    ref = F.silu(a) * b
    (g_ref,) = torch.autograd.grad((ref * cot).sum(), [ref_in])
    x = ab.to(device).requires_grad_(True)
    out = ops.swiglu(x)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close(out, ref, 2e-6, "swiglu")
    (g,) = torch.autograd.grad((out * cot.to(device)).sum(), [x])
    assert_close(g, g_ref, 2e-6, "swiglu grad")


def test_swiglu_extremes(device):
    """Large |a|: silu saturates to a or 0 without NaN/inf."""
    from amk import ops

    ab = torch.tensor([[-200.0, -30.0, 0.0, 90.0, 1.0, 2.0, 3.0, 4.0]])
    out = ops.swiglu(ab.to(device)).cpu()
    a, b = ab.chunk(2, dim=-1)
    assert torch.isfinite(out).all()
    assert_close(out, F.silu(a) * b, 1e-6, "extremes")


@pytest.mark.parametrize("shape", [(2, 16, 2 * 96), (1, 1, 8), (2, 300, 2 * 4096), (5, 7, 2 * 20)])
def test_geglu_matches_reference(device, shape):
    """amk_geglu_fwd / _bwd against the reference's GEGLU (models/transformer.py:22-27): gate * gelu(val)."""
    from amk import ops

    ab = seeded(shape, 21 + shape[-1], 2.0)
    cot = seeded(shape[:-1] + (shape[-1] // 2,), 22 + shape[-1])
    ref_in = ab.clone().requires_grad_(True)
    val, gate = ref_in.chunk(2, dim=-1)
    ref = gate * F.gelu(val)
    (g_ref,) = torch.autograd.grad((ref * cot).sum(), [ref_in])
    x = ab.to(device).requires_grad_(True)
    out = ops.geglu(x)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close(out, ref, 2e-6, "geglu")
    (g,) = torch.autograd.grad((out * cot.to(device)).sum(), [x])
    assert_close(g, g_ref, 2e-6, "geglu grad")
    ext = torch.tensor([[-200.0, -30.0, 0.0, 90.0, 1.0, 2.0, 3.0, 4.0]])
    o = ops.geglu(ext.to(device)).cpu()
    assert torch.isfinite(o).all()
    assert_close(o, ext[:, 4:] * F.gelu(ext[:, :4]), 1e-6, "geglu extremes")
