"""amk_gemm_f32 (csrc/gemm_f32.hip) against plain PyTorch fp32 / fp64 on the CPU: the three products (NT, NN, TN), the
folded element-wise passes (LayerNorm on the input tile, bias, residual, SwiGLU forward / backward, bias gradient)
and the segmented forms (two projections per launch, a contraction in two pieces, two weight gradients per launch),
at ragged sizes: rows not a multiple of the 128-row tile, contraction tails shorter than the 32-deep step, columns
not a multiple of 128.

A floating-point kernel outside the reference's module list (the nn.Linear layers of
models/softmax_attention.py:30-42,80 and models/vitvqgan.py:20-61 with the element-wise passes around them), so the
checker is PyTorch on the CPU.  Tolerance 2e-5 relative to the largest element (the north star asks 1e-4) against an
fp64 product of the same fp32 inputs.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle.fixture_recipe import seeded
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5

SHAPES = [(1, 4, 4), (5, 8, 36), (128, 128, 32), (130, 64, 40), (257, 300, 24), (1000, 192, 256), (333, 260, 1368),
          (96, 516, 64), (2048, 32, 256), (2048, 256, 32)]


def _ln(x, gamma, beta):
    return F.layer_norm(x.double(), (x.shape[1],), gamma.double(), beta.double(), 1e-5)


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "bias", "resid", "ln", "ln_bias_resid"])
def test_nt(device, M, N, K, mode):
    from amk import dense

    a, w, b, r = seeded((M, K), 1 + K), seeded((N, K), 2 + N) * K ** -0.5, seeded((N,), 3), seeded((M, N), 4)
    gam, bet = seeded((K,), 5) * 0.3 + 1.0, seeded((K,), 6) * 0.3
    use_b, use_r, use_ln = "bias" in mode, "resid" in mode, "ln" in mode
    ad = a.to(device)
    ln = None
    src = a.double()
    if use_ln:
        mean, rstd = dense.row_stats(ad)
        assert rel_err(mean, a.double().mean(1)) < 1e-5 or a.double().mean(1).abs().max() < 1e-6
        assert rel_err(rstd, 1.0 / torch.sqrt(a.double().var(1, unbiased=False) + 1e-5)) < TOL
        ln = (mean, rstd, gam.to(device), bet.to(device))
        src = _ln(a, gam, bet)
    ref = src @ w.double().t() + (b.double() if use_b else 0.0) + (r.double() if use_r else 0.0)
    out = dense.gemm_nt(ad, w.to(device), b.to(device) if use_b else None, resid=r.to(device) if use_r else None, ln=ln)
    assert out.shape == (M, N)
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("M,N1,N2,K", [(300, 128, 256, 64), (1000, 512, 1024, 256), (64, 256, 100, 40)])
@pytest.mark.parametrize("use_ln", [False, True])
def test_nt_two_projections(device, M, N1, N2, K, use_ln):
    from amk import dense

    a, w1, w2 = seeded((M, K), 1), seeded((N1, K), 2) * K ** -0.5, seeded((N2, K), 3) * K ** -0.5
    b2 = seeded((N2,), 4)
    gam, bet = seeded((K,), 5) * 0.3 + 1.0, seeded((K,), 6) * 0.3
    ad = a.to(device)
    ln, src = None, a.double()
    if use_ln:
        ln = (*dense.row_stats(ad), gam.to(device), bet.to(device))
        src = _ln(a, gam, bet)
    c1, c2 = dense.gemm_nt(ad, w1.to(device), None, w2=w2.to(device), bias2=b2.to(device), ln=ln)
    assert rel_err(c1, src @ w1.double().t()) < TOL
    assert rel_err(c2, src @ w2.double().t() + b2.double()) < TOL


@pytest.mark.parametrize("M,H,K", [(100, 64, 32), (300, 100, 40), (1000, 1368, 256), (129, 36, 256)])
@pytest.mark.parametrize("use_ln,keep", [(False, True), (True, True), (True, False)])
def test_nt_swiglu(device, M, H, K, use_ln, keep):
    from amk import dense

    a, w12, b12 = seeded((M, K), 1), seeded((2 * H, K), 2) * K ** -0.5, seeded((2 * H,), 3)
    gam, bet = seeded((K,), 5) * 0.3 + 1.0, seeded((K,), 6) * 0.3
    ad = a.to(device)
    ln, src = None, a.double()
    if use_ln:
        ln = (*dense.row_stats(ad), gam.to(device), bet.to(device))
        src = _ln(a, gam, bet)
    ab_ref = src @ w12.double().t() + b12.double()
    g_ref = F.silu(ab_ref[:, :H]) * ab_ref[:, H:]
    g, ab = dense.gemm_nt_swiglu(ad, w12.to(device), b12.to(device), ln=ln, keep_ab=keep)
    assert rel_err(g, g_ref) < TOL
    if keep:
        assert rel_err(ab, ab_ref) < TOL
    else:
        assert ab is None


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_nn(device, M, N, K):
    """dX = dY W: dY (M, K), W (K, N)."""
    from amk import dense

    if N % 4:
        pytest.skip("NN wants output widths that are multiples of 4")
    dy, w = seeded((M, K), 1 + K), seeded((K, N), 2 + N) * K ** -0.5
    out = dense.gemm_nn(dy.to(device), w.to(device))
    assert rel_err(out, dy.double() @ w.double()) < TOL


@pytest.mark.parametrize("M,K1,K2,N", [(300, 64, 128, 256), (1000, 512, 1024, 256), (70, 40, 24, 36), (257, 32, 8, 128)])
def test_nn_two_segments(device, M, K1, K2, N):
    from amk import dense

    a1, a2 = seeded((M, K1), 1), seeded((M, K2), 2)
    w1, w2 = seeded((K1, N), 3) * K1 ** -0.5, seeded((K2, N), 4) * K2 ** -0.5
    out = dense.gemm_nn(a1.to(device), w1.to(device), a2=a2.to(device), w2=w2.to(device))
    assert rel_err(out, a1.double() @ w1.double() + a2.double() @ w2.double()) < TOL


@pytest.mark.parametrize("M,H,K", [(100, 64, 32), (300, 100, 40), (1000, 1368, 256)])
def test_nn_swiglu_backward(device, M, H, K):
    """dGate = dOut W3 (W3 (K, H) as stored: out x in); (dA | dB) from the forward's (a | b)."""
    from amk import dense

    d_out, w3, ab = seeded((M, K), 1), seeded((K, H), 2) * K ** -0.5, seeded((M, 2 * H), 3)
    abr = ab.double().requires_grad_(True)
    g = F.silu(abr[:, :H]) * abr[:, H:]
    (ref,) = torch.autograd.grad(g, abr, d_out.double() @ w3.double())
    out = dense.gemm_nn(d_out.to(device), w3.to(device), swiglu_ab=ab.to(device))
    assert rel_err(out, ref) < TOL


@pytest.mark.parametrize("M,N,K", SHAPES + [(5000, 256, 512), (40000, 128, 128)])
@pytest.mark.parametrize("use_ln", [False, True])
def test_tn(device, M, N, K, use_ln):
    """dW = dY^T X' (N, K), db = column sums of dY."""
    from amk import dense

    if N % 4:
        pytest.skip("TN wants gradient widths that are multiples of 4")
    dy, x = seeded((M, N), 1 + N), seeded((M, K), 2 + K)
    gam, bet = seeded((K,), 5) * 0.3 + 1.0, seeded((K,), 6) * 0.3
    xd = x.to(device)
    ln, src = None, x.double()
    if use_ln:
        ln = (*dense.row_stats(xd), gam.to(device), bet.to(device))
        src = _ln(x, gam, bet)
    dw, dw2, db = dense.gemm_tn(dy.to(device), xd, ln=ln, want_bias=True)
    assert dw2 is None
    assert rel_err(dw, dy.double().t() @ src) < TOL
    assert rel_err(db, dy.double().sum(0)) < TOL
    again = dense.gemm_tn(dy.to(device), xd, ln=ln, want_bias=True)
    assert torch.equal(dw, again[0]) and torch.equal(db, again[2]), "the chunked sum must be bitwise reproducible"


@pytest.mark.parametrize("M,N1,N2,K", [(3000, 128, 256, 64), (5000, 512, 1024, 256), (700, 256, 100, 40)])
def test_tn_two_gradients(device, M, N1, N2, K):
    from amk import dense

    y1, y2, x = seeded((M, N1), 1), seeded((M, N2), 2), seeded((M, K), 3)
    gam, bet = seeded((K,), 5) * 0.3 + 1.0, seeded((K,), 6) * 0.3
    xd = x.to(device)
    ln = (*dense.row_stats(xd), gam.to(device), bet.to(device))
    src = _ln(x, gam, bet)
    dw1, dw2, db = dense.gemm_tn(y1.to(device), xd, y2=y2.to(device), ln=ln, want_bias=True)
    assert rel_err(dw1, y1.double().t() @ src) < TOL
    assert rel_err(dw2, y2.double().t() @ src) < TOL
    assert rel_err(db, torch.cat([y1.double().sum(0), y2.double().sum(0)])) < TOL


def test_refuses_cpu_and_bad_shapes(device):
    from amk import dense

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dense.gemm_nt(torch.randn(4, 4), torch.randn(4, 4))
    with pytest.raises(RuntimeError, match="multiples of 4"):
        dense.gemm_nt(torch.randn(4, 6, device=device), torch.randn(4, 6, device=device))


def test_auto_mode_keeps_wide_contractions_on_the_library(device, monkeypatch):
    """AMK_DENSE=auto: the own kernels take inputs of at most ops.DENSE_AUTO_MAX_K features (the ViT-VQGAN layers); the
    ViT classifier's K = 1024 layers stay on the vendor kernels, which are faster there."""
    from amk import ops

    monkeypatch.setattr(ops, "DENSE_MODE", "auto")
    w = torch.empty(1024, 256, device=device)
    assert ops._dense_ok(torch.empty(64, 256, device=device), w)
    assert not ops._dense_ok(torch.empty(64, 1024, device=device), torch.empty(1024, 1024, device=device))
    monkeypatch.setattr(ops, "DENSE_MODE", "amk")
    assert ops._dense_ok(torch.empty(64, 1024, device=device), torch.empty(1024, 1024, device=device))
    monkeypatch.setattr(ops, "DENSE_MODE", "lib")
    assert not ops._dense_ok(torch.empty(64, 256, device=device), w)
