"""amk_gemm_tn_bf16 (csrc/gemm_bf16.hip): the weight / bias gradient of nn.Linear in the mixed-precision mode against
an fp64 product of the same bf16 inputs on the CPU.  The kernel multiplies bf16 values exactly and accumulates in f32,
so the only error is the f32 summation: tolerance 2e-5 relative to the largest element (the vendor GEMM it replaces
rounds the result to bf16, 4e-3).  Reference: the backward of nn.Linear in models/softmax_attention.py:30-42,80 and
models/vitvqgan.py:20-34 under cfg/vitvqgan.yaml:73 (mixed_precision: bf16).
"""
import pytest
import torch

from oracle.fixture_recipe import seeded
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5

SHAPES = [(1, 8, 8), (5, 8, 40), (31, 16, 24), (33, 136, 8), (257, 304, 24), (1000, 192, 256), (3333, 264, 1368),
          (4096, 2736, 256), (40000, 128, 128), (20000, 256, 512)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("want_bias", [False, True])
def test_tn_bf16(device, M, N, K, want_bias):
    from amk import dense

    y, x = seeded((M, N), 1 + N).bfloat16(), seeded((M, K), 2 + K).bfloat16()
    dw, db = dense.gemm_tn_bf16(y.to(device), x.to(device), want_bias=want_bias)
    assert dw.dtype == torch.float32 and dw.shape == (N, K)
    assert rel_err(dw, y.double().t() @ x.double()) < TOL
    if want_bias:
        assert rel_err(db, y.double().sum(0)) < TOL
    else:
        assert db is None
    again = dense.gemm_tn_bf16(y.to(device), x.to(device), want_bias=want_bias)
    assert torch.equal(dw, again[0]), "the chunked sum must be bitwise reproducible"


def test_tn_bf16_strided_rows(device):
    """y and x as column slices of wider matrices (the q | kv halves of one projection output)."""
    from amk import dense

    wide = seeded((3000, 768), 3).bfloat16().to(device)
    x = seeded((3000, 256), 4).bfloat16().to(device)
    y = wide[:, 256:768]
    dw, db = dense.gemm_tn_bf16(y, x, want_bias=True)
    assert rel_err(dw, y.double().cpu().t() @ x.double().cpu()) < TOL
    assert rel_err(db, y.double().cpu().sum(0)) < TOL


def test_tn_bf16_refuses(device):
    from amk import dense

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dense.gemm_tn_bf16(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="multiples of 8"):
        dense.gemm_tn_bf16(torch.zeros(8, 12, dtype=torch.bfloat16, device=device), torch.zeros(8, 8, dtype=torch.bfloat16, device=device))
    with pytest.raises(RuntimeError, match="bf16"):
        dense.gemm_tn_bf16(torch.zeros(8, 8, device=device), torch.zeros(8, 8, device=device))


@pytest.mark.parametrize("shape,N,bias", [((4, 100, 256), 512, True), ((2, 64, 264), 1368, False), ((3000, 256), 256, True)])
def test_linear_under_autocast(device, shape, N, bias):
    """ops.linear under bf16 autocast: output and input gradient are the library's (bitwise the same calls as
    F.linear under autocast); weight and bias gradient are the exact products of the bf16 operands, which the
    library's bf16-rounded gradients approach to 2^-8."""
    import torch.nn.functional as F

    from amk import ops

    K = shape[-1]
    x = seeded(shape, 1).to(device).requires_grad_(True)
    w = (seeded((N, K), 2) * K ** -0.5).to(device).requires_grad_(True)
    b = seeded((N,), 3).to(device).requires_grad_(True) if bias else None
    g = seeded(shape[:-1] + (N,), 4).to(device)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.linear(x, w, b)
        y_ref = F.linear(x, w, b)
    assert y.dtype == torch.bfloat16 and torch.equal(y, y_ref)
    ins = (x, w) + ((b,) if bias else ())
    got = torch.autograd.grad(y, ins, g.bfloat16())
    ref = torch.autograd.grad(y_ref, ins, g.bfloat16())
    assert got[0].dtype == torch.float32 and torch.equal(got[0], ref[0])
    g16, x16 = g.bfloat16().double().reshape(-1, N).cpu(), x.detach().bfloat16().double().reshape(-1, K).cpu()
    assert got[1].dtype == torch.float32 and rel_err(got[1], g16.t() @ x16) < TOL
    assert rel_err(ref[1], g16.t() @ x16) < 1e-2
    if bias:
        assert got[2].dtype == torch.float32 and rel_err(got[2], g16.sum(0)) < TOL
