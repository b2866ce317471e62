"""The bf16-in / bf16-out element-wise kernels of the mixed-precision mode (csrc/mixed_bf16.hip: SwiGLU gate, residual
add + LayerNorm) against plain PyTorch f32 computations on the same (bf16-rounded) inputs.  Floating-point kernels
outside the reference's module list (SURVEY.md section 8f rank 1 under the reference's shipped bf16 autocast,
cfg/vitvqgan.yaml:73); tolerance: bf16 output rounding, 2^-8 relative to the largest element, i.e. 4e-3 (8e-3 for
gradients that pass through two roundings); f32 outputs (the residual stream, its gradient, statistics) 2e-5."""
import pytest
import torch
import torch.nn.functional as F

from oracle.fixture_recipe import seeded
from util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,H", [(7, 8), (1000, 1368), (130, 64)])
def test_swiglu_bf16(device, M, H):
    from amk import ops

    ab = seeded((M, 2 * H), 1).bfloat16()
    cot = seeded((M, H), 2).bfloat16()
    r = ab.float().requires_grad_(True)
    ref = F.silu(r[:, :H]) * r[:, H:]
    (gref,) = torch.autograd.grad(ref, r, cot.float())
    x = ab.to(device).requires_grad_(True)
    out = ops.swiglu(x)
    assert out.dtype == torch.bfloat16
    (g,) = torch.autograd.grad(out, x, cot.to(device))
    assert g.dtype == torch.bfloat16
    assert rel_err(out.float(), ref) < 4e-3
    assert rel_err(g.float(), gref) < 8e-3


@pytest.mark.parametrize("M,D", [(5, 4), (1000, 256), (130, 1024), (33, 768)])
@pytest.mark.parametrize("x_bf16,residual", [(True, True), (False, True), (False, False), (True, False)])
def test_add_layernorm_mixed(device, M, D, x_bf16, residual):
    from amk import ops

    x = seeded((M, D), 11 + D)
    if x_bf16:
        x = x.bfloat16()
    r = seeded((M, D), 12 + D)
    w = seeded((D,), 13 + D) * 0.5 + 1.0
    b = seeded((D,), 14 + D)
    cy, ch = seeded((M, D), 15 + D).bfloat16(), seeded((M, D), 16 + D)
    xr, rr, wr, br = x.float().requires_grad_(True), r.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    h_ref = xr + rr if residual else xr
    y_ref = F.layer_norm(h_ref, (D,), wr, br, 1e-5)
    loss_ref = (y_ref * cy.float()).sum() + (h_ref * ch).sum()
    g_ref = torch.autograd.grad(loss_ref, [xr, rr, wr, br], allow_unused=True)

    xd, rd, wd, bd = (t.to(device).requires_grad_(True) for t in (x, r, w, b))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if residual:
            h, y = ops.add_layer_norm(xd, rd, wd, bd, 1e-5, branch=True)
        else:
            y = ops.layer_norm(xd, wd, bd, 1e-5, branch=True)
            h = xd
    assert y.dtype == torch.bfloat16
    loss = (y.float() * cy.to(device).float()).sum() + (h.float() * ch.to(device)).sum()
    gs = torch.autograd.grad(loss, [xd] + ([rd] if residual else []) + [wd, bd])
    assert rel_err(y.float(), y_ref) < 4e-3
    if residual:
        assert h.dtype == torch.float32 and rel_err(h, h_ref) < 2e-5
    assert gs[0].dtype == x.dtype
    assert rel_err(gs[0].float(), g_ref[0]) < (8e-3 if x_bf16 else 2e-5)
    if residual:
        assert gs[1].dtype == torch.float32 and rel_err(gs[1], g_ref[1]) < 2e-5
    assert rel_err(gs[-2], g_ref[2]) < 2e-5 and rel_err(gs[-1], g_ref[3]) < 2e-5
