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


FWD_SHAPES = [(1, 8, 8), (5, 16, 40), (130, 136, 24), (257, 304, 72), (1000, 192, 256), (333, 264, 1368), (4096, 2736, 256),
              (2048, 32, 512), (128, 128, 32)]
TOL16 = 2 ** -8   # one bf16 rounding of the output (2^-9 relative to each element, here relative to the largest)


@pytest.mark.parametrize("M,N,K", FWD_SHAPES)
@pytest.mark.parametrize("bias", [False, True])
def test_nt_bf16(device, M, N, K, bias):
    from amk import dense

    a, w = seeded((M, K), 1 + K).bfloat16(), (seeded((N, K), 2 + N) * K ** -0.5).bfloat16()
    b = seeded((N,), 3) if bias else None
    ref = a.double() @ w.double().t() + (b.double() if bias else 0.0)
    out = dense.gemm_nt_bf16(a.to(device), w.to(device), b.to(device) if bias else None)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    assert rel_err(out.float(), ref) < TOL16
    # exactly the correctly rounded result wherever the fp32 sum is not at a rounding boundary
    exact = (out.cpu() == ref.float().bfloat16()).float().mean().item()
    assert exact > 0.99


@pytest.mark.parametrize("M,N,K", FWD_SHAPES)
def test_nn_bf16(device, M, N, K):
    from amk import dense

    a, w = seeded((M, K), 1 + K).bfloat16(), (seeded((K, N), 2 + N) * K ** -0.5).bfloat16()
    ref = a.double() @ w.double()
    out = dense.gemm_nn_bf16(a.to(device), w.to(device))
    assert rel_err(out.float(), ref) < TOL16
    assert (out.cpu() == ref.float().bfloat16()).float().mean().item() > 0.99


@pytest.mark.parametrize("M,H,K", [(100, 64, 32), (300, 104, 40), (1000, 1368, 256), (129, 40, 256), (4096, 8, 64)])
@pytest.mark.parametrize("keep", [True, False])
def test_nt_swiglu_bf16(device, M, H, K, keep):
    import torch.nn.functional as F

    from amk import dense

    a, w12, b12 = seeded((M, K), 1).bfloat16(), (seeded((2 * H, K), 2) * K ** -0.5).bfloat16(), seeded((2 * H,), 3)
    ab_ref = a.double() @ w12.double().t() + b12.double()
    g_ref = F.silu(ab_ref[:, :H]) * ab_ref[:, H:]
    g, ab = dense.gemm_nt_swiglu_bf16(a.to(device), w12.to(device), b12.to(device), keep_ab=keep)
    assert g.shape == (M, H) and rel_err(g.float(), g_ref) < TOL16
    if keep:
        assert rel_err(ab.float(), ab_ref) < TOL16
    else:
        assert ab is None


def test_swiglu_ffn_under_autocast(device):
    """ops.swiglu_ffn under bf16 autocast against the separate autocast ops (F.linear, silu * value, F.linear): bf16-level
    agreement of the output and every gradient; weight / bias gradients in fp32."""
    import torch.nn.functional as F

    from amk import ops

    D, H = 256, 1368
    x = seeded((4, 200, D), 1).to(device).requires_grad_(True)
    w12 = (seeded((2 * H, D), 2) * D ** -0.5).to(device).requires_grad_(True)
    b12 = (seeded((2 * H,), 3) * 0.1).to(device).requires_grad_(True)
    w3 = (seeded((D, H), 4) * H ** -0.5).to(device).requires_grad_(True)
    b3 = (seeded((D,), 5) * 0.1).to(device).requires_grad_(True)
    g = seeded((4, 200, D), 6).to(device).bfloat16()
    ins = (x, w12, b12, w3, b3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.swiglu_ffn(x, w12, b12, w3, b3)
        ab = F.linear(x, w12, b12)
        a, b = ab.chunk(2, dim=-1)
        y_ref = F.linear(F.silu(a) * b, w3, b3)
    assert y.dtype == torch.bfloat16
    assert rel_err(y.float(), y_ref.float()) < 2e-2
    got, ref = torch.autograd.grad(y, ins, g), torch.autograd.grad(y_ref, ins, g)
    # fp64 gradients of the fp32 module as the yardstick for both
    xd, w12d, b12d, w3d, b3d = (t.detach().double().requires_grad_(True) for t in ins)
    abd = F.linear(xd, w12d, b12d)
    ad, bd = abd.chunk(2, dim=-1)
    exact = torch.autograd.grad(F.linear(F.silu(ad) * bd, w3d, b3d), (xd, w12d, b12d, w3d, b3d), g.double())
    for name, u, v, e in zip(("dx", "dw12", "db12", "dw3", "db3"), got, ref, exact):
        assert u.dtype == torch.float32 and u.shape == v.shape
        assert rel_err(u, e) < 2e-2, name
        assert rel_err(u, e) < 1.5 * rel_err(v.float(), e) + 1e-3, f"{name}: no worse than the library path"


@pytest.mark.parametrize("M,H,K", [(100, 64, 32), (300, 104, 40), (1000, 1368, 256), (129, 40, 256)])
def test_nn_swiglu_backward_bf16(device, M, H, K):
    """Bitwise the pair amk_gemm_bf16(op 1) + amk_swiglu_bf16_bwd (dG rounded to bf16 in between), and close to fp64."""
    import ctypes

    import torch.nn.functional as F

    from amk import dense, lib

    dy, w3, ab = seeded((M, K), 1).bfloat16().to(device), (seeded((K, H), 2) * K ** -0.5).bfloat16().to(device), seeded((M, 2 * H), 3).bfloat16().to(device)
    out = dense.gemm_nn_swiglu_bwd_bf16(dy, w3, ab)
    dg = dense.gemm_nn_bf16(dy, w3)
    two = torch.empty_like(ab)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.check(lib.load().amk_swiglu_bf16_bwd(P(ab), P(dg), M, H, P(two), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "swiglu bwd")
    assert rel_err(out.float(), two.float()) < 2 ** -7   # (rcp / exp2 against the other kernel's expf: last-bit differences)
    abr = ab.double().cpu().requires_grad_(True)
    g = F.silu(abr[:, :H]) * abr[:, H:]
    (ref,) = torch.autograd.grad(g, abr, dy.double().cpu() @ w3.double().cpu())
    assert rel_err(out.float(), ref) < 2e-2


def test_swiglu_ffn_under_autocast_no_grad(device):
    import torch.nn.functional as F

    from amk import ops

    D, H = 256, 1368
    x = seeded((2, 300, D), 1).to(device)
    w12, b12 = (seeded((2 * H, D), 2) * D ** -0.5).to(device), (seeded((2 * H,), 3) * 0.1).to(device)
    w3, b3 = (seeded((D, H), 4) * H ** -0.5).to(device), (seeded((D,), 5) * 0.1).to(device)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.swiglu_ffn(x, w12, b12, w3, b3)
        a, b = F.linear(x, w12, b12).chunk(2, dim=-1)
        y_ref = F.linear(F.silu(a) * b, w3, b3)
    assert y.dtype == torch.bfloat16 and y.shape == y_ref.shape
    assert rel_err(y.float(), y_ref.float()) < 2e-2
