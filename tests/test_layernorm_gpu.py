"""amk_add_layernorm_fwd / _bwd and amk_colsum against torch's own fp32 LayerNorm / sums on the CPU.

These are floating-point kernels outside the reference's module list (SURVEY.md section 8f rank 1:
the pre-LN / residual epilogue), so the checker is a plain PyTorch fp32 computation of the same op;
tolerance 2e-5 relative (the north star asks 1e-4).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle.fixture_recipe import seeded
from util import assert_close

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.mark.parametrize("M,D", [(1, 4), (7, 256), (130, 32), (33, 768), (64, 1024), (5, 4096), (1000, 192), (3, 260)])
@pytest.mark.parametrize("residual", [False, True])
def test_layernorm_matches_torch(device, M, D, residual):
    from amk import ops

    x = seeded((M, D), 11 + D)
    r = seeded((M, D), 12 + D)
    w = seeded((D,), 13 + D) * 0.5 + 1.0
    b = seeded((D,), 14 + D)
    cy = seeded((M, D), 15 + D)
    ch = seeded((M, D), 16 + D)
    xr, rr, wr, br = (t.clone().requires_grad_(True) for t in (x, r, w, b))
    h_ref = xr + rr if residual else xr
    y_ref = F.layer_norm(h_ref, (D,), wr, br, 1e-5)
    loss_ref = (y_ref * cy).sum() + ((h_ref * ch).sum() if residual else 0.0)
    g_ref = torch.autograd.grad(loss_ref, [xr, rr, wr, br], allow_unused=True)

    xd, rd, wd, bd = (t.to(device).requires_grad_(True) for t in (x, r, w, b))
    if residual:
        h, y = ops.add_layer_norm(xd, rd, wd, bd, 1e-5)
        assert_close(h, h_ref, TOL, "h")
        loss = (y * cy.to(device)).sum() + (h * ch.to(device)).sum()
    else:
        y = ops.layer_norm(xd, wd, bd, 1e-5)
        loss = (y * cy.to(device)).sum()
    assert_close(y, y_ref, TOL, "y")
    g = torch.autograd.grad(loss, [xd, rd, wd, bd], allow_unused=True)
    assert_close(g[0], g_ref[0], TOL, "dx")
    if residual:
        assert_close(g[1], g_ref[1], TOL, "dres")
    assert_close(g[2], g_ref[2], TOL, "dgamma")
    assert_close(g[3], g_ref[3], TOL, "dbeta")


def test_add_layernorm_unused_outputs(device):
    """Only h used downstream (dy is None) and only y used (dh is None)."""
    from amk import ops

    x = seeded((9, 64), 1).to(device).requires_grad_(True)
    r = seeded((9, 64), 2).to(device).requires_grad_(True)
    w = torch.ones(64, device=device, requires_grad=True)
    b = torch.zeros(64, device=device, requires_grad=True)
    h, y = ops.add_layer_norm(x, r, w, b)
    (gx,) = torch.autograd.grad(h.sum(), [x], retain_graph=True)
    assert torch.equal(gx, torch.ones_like(gx))
    c = seeded((9, 64), 3)
    gx2, gr2 = torch.autograd.grad((y * c.to(device)).sum(), [x, r])
    assert torch.equal(gx2, gr2)
    xr = (x.detach().cpu() + r.detach().cpu()).requires_grad_(True)
    (gref,) = torch.autograd.grad((F.layer_norm(xr, (64,)) * c).sum(), [xr])
    assert_close(gx2, gref, TOL, "dx via y only")


@pytest.mark.parametrize("M,N", [(1, 4), (77, 256), (32768, 256), (1000, 2736), (513, 4096), (64, 30)])
def test_colsum_and_bias_linear(device, M, N):
    from amk import ops

    x = seeded((M, N), 3 + N)
    got = ops.colsum(x.to(device))
    want = x.double().sum(0).float()
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())) * max(1.0, M ** 0.5 / 8)

    K = 24
    a = seeded((min(M, 200), K), 5)
    wgt = seeded((N, K), 6)
    bias = seeded((N,), 7)
    cot = seeded((a.shape[0], N), 8)
    ar, wr, br = (t.clone().requires_grad_(True) for t in (a, wgt, bias))
    g_ref = torch.autograd.grad((F.linear(ar, wr, br) * cot).sum(), [ar, wr, br])
    ad, wd, bd = (t.to(device).requires_grad_(True) for t in (a, wgt, bias))
    out = ops.linear(ad, wd, bd)
    assert_close(out, F.linear(a, wgt, bias), TOL, "linear out")
    g = torch.autograd.grad((out * cot.to(device)).sum(), [ad, wd, bd])
    for name, u, v in zip(("dx", "dw", "db"), g, g_ref):
        assert_close(u, v, TOL, name)


def test_fused_block_equals_unfused(device):
    """TransformerBlock's (h, pending) residual stream = the plain x + f(LN(x)) chain of EncoderLayer.forward."""
    from amk.models.vitvqgan import TransformerBlock

    torch.manual_seed(0)
    blk = TransformerBlock(128, 2, 64, depth=3, mlp_dim=256).to(device)
    x = torch.randn(2, 70, 128, device=device, requires_grad=True)
    cot = torch.randn(2, 70, 128, device=device)
    out_f = blk(x)
    g_f = torch.autograd.grad((out_f * cot).sum(), [x] + list(blk.parameters()))
    y = x
    for layer in blk.layers:
        y = layer(y)
    g_u = torch.autograd.grad((y * cot).sum(), [x] + list(blk.parameters()))
    assert_close(out_f, y, TOL, "fused vs unfused out")
    for a, b in zip(g_f, g_u):
        assert_close(a, b, 5e-5, "fused vs unfused grad")
