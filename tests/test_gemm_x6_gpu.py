"""The split-bf16 GEMM (csrc/gemm_x6.hip, amk_gemm_x6_nt; ops.GEMM_MODE = "bf16x6") against float64 and
against the exact-f32 library GEMM, then the whole ViT-VQGAN (forward, loss, indices, every gradient) with
every nn.Linear on it against the reference-generated golden fixture at the exact-f32 path's tolerances."""
import pytest
import torch

from oracle.fixture_recipe import seeded
from util import assert_close, load_golden, rel_err, weights_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 256), (100, 70, 52), (1, 1, 4), (333, 2736, 256),
                                   (512, 256, 1368), (4096, 1024, 256), (77, 200, 36)])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_x6_nt_error_is_f32_level(device, M, N, K, bias):
    from amk import ops

    a, b = seeded((M, K), 1), seeded((N, K), 2)
    bv = seeded((N,), 3) if bias else None
    want = a.double() @ b.double().t() + (bv.double() if bias else 0.0)
    got = ops.gemm_x6_nt(a.to(device), b.to(device), bv.to(device) if bias else None)
    lib = torch.nn.functional.linear(a.to(device), b.to(device), bv.to(device) if bias else None)
    e_x6, e_lib = rel_err(got, want), rel_err(lib, want)
    assert e_x6 <= 2e-6, (e_x6, e_lib)                    # max |c - c64| / max |c64|
    assert e_x6 <= 3.0 * e_lib + 5e-7, (e_x6, e_lib)      # the level of the exact-f32 library GEMM


def test_gemm_x6_strided_rows(device):
    """Row strides other than K (the q / kv views of a fused projection, a transposed weight copy)."""
    from amk import ops

    big = seeded((64, 200), 4).to(device)
    a = big[:, 40:104]            # (64, 64) view, row stride 200, 16-byte aligned start
    b = seeded((48, 64), 5).to(device)
    assert_close(ops.gemm_x6_nt(a, b), a.double().cpu() @ b.double().cpu().t(), 2e-6, "strided A")


@pytest.fixture
def x6_mode():
    from amk import ops

    old, ops.GEMM_MODE = ops.GEMM_MODE, "bf16x6"
    yield
    ops.GEMM_MODE = old


def test_linear_autograd_x6(device, x6_mode):
    from amk import ops

    x = seeded((3, 50, 96), 1).to(device).requires_grad_(True)
    w = seeded((40, 96), 2, 0.1).to(device).requires_grad_(True)
    b = seeded((40,), 3).to(device).requires_grad_(True)
    cot = seeded((3, 50, 40), 4).to(device)
    y = ops.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad((y * cot).sum(), [x, w, b])
    x64, w64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    wx, ww, wb = torch.autograd.grad((y64 * cot.double().cpu()).sum(), [x64, w64, b64])
    for name, a_, b_ in (("y", y, y64), ("dx", gx, wx), ("dw", gw, ww), ("db", gb, wb)):
        assert_close(a_, b_, 3e-6, name)


def test_vitvqgan_small_golden_on_x6_gemms(device, x6_mode):
    """tests/test_vq_gpu.py::test_vitvqgan_small_golden with every projection / FFN / quant GEMM on the
    split-bf16 kernel: same fixture (generated from the reference), same tolerances, indices bit-exact."""
    import numpy as np

    from amk.models import ViTVQGAN

    import json
    import os

    from util import GOLDEN

    fx = load_golden("vitvqgan_small")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["vitvqgan_small"]
    m = ViTVQGAN(meta["cfg"], meta["codebook"])
    m.load_state_dict(weights_of(fx))
    m = m.to(device)
    imgs = torch.from_numpy(fx["imgs"]).to(device)
    rec, loss = m(imgs)
    idx = m.encode_imgs(imgs)
    bad = np.nonzero(idx.cpu().numpy().reshape(-1) != fx["idx"].reshape(-1))[0]
    assert all(fx["margin"].reshape(-1)[i] < 1e-6 for i in bad)
    assert_close(rec, fx["rec"], 5e-5, "reconstruction")
    assert_close(loss, fx["loss"], 5e-5, "codebook loss")
    total = torch.nn.functional.mse_loss(rec, imgs) + torch.nn.functional.l1_loss(rec, imgs) + loss
    params = dict(m.named_parameters())
    names = sorted(params)
    gs = torch.autograd.grad(total, [params[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        if "g:" + n in fx:
            assert_close(g, fx["g:" + n], 2e-4, f"grad {n}")
