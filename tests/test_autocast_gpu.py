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


def test_train_step_under_bf16_autocast(device):
    """One GAN step with both phases' forwards under bf16 autocast (the reference's accelerator.autocast blocks):
    losses finite and within mixed-precision distance of the f32 step, parameters updated and still f32."""
    import copy

    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    torch.manual_seed(0)
    vit = dict(dim=128, img_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=256, dropout=0.0)
    model = ViTVQGAN(vit, dict(codebook_size=512, codebook_dim=32)).to(device)
    discr = NLayerDiscriminator(3, 16, 3).to(device)
    model2, discr2 = copy.deepcopy(model), copy.deepcopy(discr)
    img = torch.rand(4, 3, 64, 64, device=device)
    eta = torch.rand(4, 1, 1, 1, device=device)
    ref = VQGANTrainStep(model, discr).step(img, eta=eta)
    before = {n: p.detach().clone() for n, p in model2.named_parameters()}
    got = VQGANTrainStep(model2, discr2, autocast=torch.bfloat16).step(img, eta=eta)
    for key in ("d_loss", "g_loss", "l1", "l2", "codebook_loss"):
        a, b = float(got[key]), float(ref[key])
        assert a == a and abs(a - b) <= 0.05 * max(abs(b), 0.1), (key, a, b)
    moved = 0
    for n, p in model2.named_parameters():
        assert p.dtype == torch.float32 and torch.isfinite(p).all()
        moved += int(not torch.equal(p, before[n]))
    assert moved > 0


def test_bf16_training_stays_finite_on_a_small_model(device):
    """Twenty GAN steps under bf16 autocast on a 64 px toy: every parameter and loss stays finite and the
    reconstruction loss goes down.  (MIOpen's bf16 weight-gradient kernel for the 3-channel image layer returned NaN from
    finite operands here at step 3; that layer now stays in f32, amk/models/discriminator.py.)"""
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    torch.manual_seed(0)
    cfg = dict(dim=128, img_size=64, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=256, dropout=0.0)
    model = ViTVQGAN(cfg, dict(codebook_size=512, codebook_dim=32)).to(device)
    discr = NLayerDiscriminator(3, 32, 3).to(device)
    tr = VQGANTrainStep(model, discr, lr=1e-3, warmup_steps=10, decay_steps=400, autocast=torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    imgs = torch.nn.functional.interpolate(torch.rand(8, 3, 8, 8, generator=g), size=64, mode="bilinear").to(device)
    first = None
    for step in range(20):
        logs = tr.step(imgs)
        assert all(bool(torch.isfinite(v)) for v in logs.values()), (step, {k: float(v) for k, v in logs.items()})
        first = first if first is not None else float(logs["l2"])
    assert all(bool(torch.isfinite(p).all()) for p in list(model.parameters()) + list(discr.parameters()))
    assert float(logs["l2"]) < first


def test_bf16_resume_and_shared_forward(device, tmp_path):
    """Under bf16 autocast: resume_from_checkpoint refreshes the bf16 parameter copies (they equal a cast of the loaded
    weights and are the ones the GEMMs read), and the shared-forward form of the step stays finite."""
    from amk import ops
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.train import VQGANTrainStep

    cfg = dict(dim=128, img_size=32, patch_size=8, n_heads=2, d_head=64, depth=1, mlp_dim=256, dropout=0.0)
    imgs = torch.rand(4, 3, 32, 32, device=device)

    def make(seed, **kw):
        torch.manual_seed(seed)
        return VQGANTrainStep(ViTVQGAN(cfg, dict(codebook_size=64, codebook_dim=32)).to(device),
                              NLayerDiscriminator(3, 8, 3).to(device), warmup_steps=1, autocast=torch.bfloat16, **kw)

    a = make(0)
    for _ in range(2):
        a.step(imgs)
    path = str(tmp_path / "ckpt.pt")
    a.save_ckpt(path, config={"note": "test"})
    b = make(1)
    b.resume_from_checkpoint(path)
    assert b.global_step == a.global_step
    for (n, p), q in zip(b.model.named_parameters(), a.model.parameters()):
        assert torch.equal(p, q), n
        assert ops._w16(p) is p._amk_bf16 and torch.equal(p._amk_bf16, p.detach().to(torch.bfloat16)), n
    logs = b.step(imgs)
    assert all(bool(torch.isfinite(v)) for v in logs.values())
    c = make(2, share_forward=True)
    for _ in range(3):
        logs = c.step(imgs)
    assert all(bool(torch.isfinite(v)) for v in logs.values())
    assert all(bool(torch.isfinite(p).all()) for p in c.model.parameters())


def test_muse_decoder_with_padded_prompt_under_bf16_autocast(device):
    """models/muse.py:88-96 with a key-padding mask on the text positions (a padded prompt), under the reference's autocast
    setting: the cross-attention (1024-token-like queries x padded text keys) runs on the MASKED bf16-MFMA kernels -- not on
    the f32 kernels with upcast inputs, as until round 3 -- and stays within bf16 rounding of the f32 pass."""
    from amk import ops
    from amk.models.muse import BidirectionalDecoder

    torch.manual_seed(0)
    dec = BidirectionalDecoder(dim=128, codebook_size=64, n_heads=2, d_head=64, depth=2, mult=2, dropout=0.0, num_patches=96).to(device)
    ids = torch.randint(0, 65, (3, 96), device=device)
    ctx = torch.randn(3, 20, 128, device=device)
    keep = torch.ones(3, 20, dtype=torch.bool, device=device)
    keep[0, 7:] = False
    keep[1, 13:] = False
    cot = torch.randn(3, 96, 64, device=device)
    out32 = dec(ids, context=ctx, context_mask=keep)
    g32 = torch.autograd.grad((out32 * cot).sum(), [p for p in dec.parameters() if p.requires_grad], allow_unused=True)
    ops.KERNEL_EVENTS = {}
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = dec(ids, context=ctx, context_mask=keep)
        g16 = torch.autograd.grad((out16.float() * cot).sum(), [p for p in dec.parameters() if p.requires_grad], allow_unused=True)
        torch.cuda.synchronize()
        names = set(ops.KERNEL_EVENTS)
    finally:
        ops.KERNEL_EVENTS = None
    assert "attn_bf16_fwd_kernel<masked>" in names and "attn_bf16_bwd_kernel<masked>" in names, names   # the cross-attention
    assert "attn_bf16_fwd_kernel" in names                                                            # the (unmasked) self-attention
    assert not any(n.startswith("attn_fwd") or n.startswith("attn_bwd") for n in names), names        # no f32 attention kernel ran
    err = float((out16.detach().float() - out32.detach()).abs().max() / out32.detach().abs().max())
    assert err < 3e-2, err
    for a, b in zip(g16, g32):
        if b is not None:
            assert a is not None and torch.isfinite(a).all()
            assert float((a.float() - b).abs().max()) <= 6e-2 * float(b.abs().max()) + 1e-6
