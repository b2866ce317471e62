"""Mixed precision: the reference's shipped configuration trains under accelerate's bf16 autocast
(cfg/vitvqgan.yaml:73, trainers/vitgqgan.py:149,168).  The kernels of libamk.so are f32; inside a torch.autocast region
the ops take their inputs as f32 (bf16 activations from the autocast Linear layers are upcast), run as usual and return
f32, so the models work unchanged with the library GEMMs in bf16.  Outside autocast a bf16 tensor is still refused."""
import pytest
import torch

from util import seeded

pytestmark = pytest.mark.gpu


def test_vitvqgan_under_bf16_autocast(device):
    from amk.models import ViTVQGAN

    torch.manual_seed(0)
    vit = dict(dim=128, img_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=256, dropout=0.0)
    model = ViTVQGAN(vit, dict(codebook_size=512, codebook_dim=32)).to(device)
    img = torch.rand(4, 3, 64, 64, device=device)
    rec32, loss32 = model(img)
    (rec32.square().mean() + loss32).backward()
    g32 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        rec16, loss16 = model(img)
        total = rec16.float().square().mean() + loss16
    total.backward()
    assert torch.isfinite(rec16).all() and torch.isfinite(loss16)
    # bf16 GEMMs around exact attention / VQ: the reconstruction stays within bf16 rounding of the f32 pass
    # (codes can flip where two codebook rows are nearly equidistant, which moves single patches)
    err = float((rec16.detach().float() - rec32.detach()).abs().mean() / rec32.detach().abs().mean())
    assert err < 0.08, err
    for n, p in model.named_parameters():
        if n in g32:
            assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), n


def test_moe_and_switchhead_under_bf16_autocast(device):
    from amk.models import MoELayer, SwitchHeadAttention

    x = seeded((2, 65, 256), 1).to(device).requires_grad_(True)
    for m in (MoELayer(256, 256, 8, 2).to(device), SwitchHeadAttention(256, 4, 64, num_experts=4, sel_experts=2).to(device)):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x)
        assert out.shape == x.shape and torch.isfinite(out).all()
        out.float().sum().backward()
        assert torch.isfinite(x.grad).all()


def test_bf16_tensors_are_refused_outside_autocast(device):
    from amk import ops

    q = torch.randn(1, 2, 64, 64, device=device, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="fp32"):
        ops.attention(q, q, q, 0.125)
