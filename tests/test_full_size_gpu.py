"""BASELINE.json configs[1] (ViT, dim 1024, depth 6), configs[3] (ViTMoE, E 32) and configs[4] (Muse decoder,
dim 1024, depth 22, 479 M parameters) AT THEIR OWN SIZES on the GPU.  The oracle cannot run these in seconds
(the reference needs 3 s for one ViTMoE step at batch 2 on 8 cores), so the checks are the size-independent
ones: finite loss, non-zero finite gradients for every parameter the reference gives one, bitwise
repeatability on the reproducible attention backward, and -- for the shape only config 4 has, cross
attention with 1024 queries over 77 keys and 16 heads -- slices of the kernels' output against the oracle."""
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded
from util import assert_close

pytestmark = pytest.mark.gpu


def _grad_summary(model):
    tot, missing = 0.0, []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            missing.append(n)
            continue
        assert torch.isfinite(p.grad).all(), n
        tot += float(p.grad.double().pow(2).sum())
    return tot ** 0.5, missing


def _step_twice_bitwise(model, loss_fn):
    """Two forward + backward passes from the same state on the reproducible attention backward."""
    from amk import ops

    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        out = []
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            loss = loss_fn()
            loss.backward()
            out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
    (l0, g0), (l1, g1) = out
    assert torch.equal(l0, l1)
    return l0, g0, g1


def test_vit_config1_full_size(device):
    """configs[1]: ViT dim 1024, patch 32, img 256, 16 heads, depth 6, 1000 classes (33.6 M parameters)."""
    from amk.models import ViT

    torch.manual_seed(0)
    model = ViT(dim=1024, image_size=256, patch_size=32, n_heads=16, d_head=64, depth=6, mlp_dim=2048, dropout=0.0,
                num_classes=1000).to(device)
    assert sum(p.numel() for p in model.parameters()) == 33629160
    g = torch.Generator().manual_seed(1)
    imgs = torch.randn(16, 3, 256, 256, generator=g).to(device)
    labels = torch.randint(0, 1000, (16,), generator=g).to(device)
    loss, g0, g1 = _step_twice_bitwise(model, lambda: torch.nn.functional.cross_entropy(model(imgs), labels))
    assert torch.isfinite(loss) and 5.0 < float(loss) < 9.0       # ~ln(1000) at random init
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    norm, missing = _grad_summary(model)
    assert norm > 0
    # the reference's ViT never reads encoder.feed_forward: the only parameters without a gradient
    assert all(n.startswith("encoder.feed_forward") for n in missing), missing


def test_vit_moe_config3_full_size(device):
    """configs[3]: ViTMoE dim 1024, patch 32, depth 6, 32 experts top-2, SwitchHead h 8 (240.6 M parameters)."""
    from amk.models import ViTMoE

    torch.manual_seed(0)
    model = ViTMoE(dim=1024, image_size=256, patch_size=32, n_heads=8, d_head=64, depth=6, n_experts=32, sel_experts=2,
                   dropout=0.0, num_classes=1000).to(device)
    assert abs(sum(p.numel() for p in model.parameters()) - 240.6e6) < 0.1e6
    g = torch.Generator().manual_seed(2)
    imgs = torch.randn(8, 3, 256, 256, generator=g).to(device)
    labels = torch.randint(0, 1000, (8,), generator=g).to(device)
    loss, g0, g1 = _step_twice_bitwise(model, lambda: torch.nn.functional.cross_entropy(model(imgs), labels))
    assert torch.isfinite(loss)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    norm, missing = _grad_summary(model)
    assert norm > 0
    assert all("W_d" in n for n in missing), missing   # moe_out never uses its gate weights


def test_muse_decoder_config4_full_size(device):
    """configs[4]: Muse decoder D 1024, 16 heads, depth 22, mult 6, vocabulary 8192, 1024 image tokens x 77 text
    positions (479 M parameters), training forward + backward over frozen ViT-VQGAN codes."""
    from amk.models import MUSE, ViTVQGAN

    torch.manual_seed(0)
    vq = ViTVQGAN(dict(dim=256, img_size=256, patch_size=8, n_heads=8, d_head=64, depth=6, mlp_dim=2048, dropout=0.0),
                  dict(codebook_size=8192, codebook_dim=32))
    muse = MUSE(dim=1024, vq=vq, n_heads=16, d_head=64, depth=22, mult=6).to(device)
    nparam = sum(p.numel() for p in muse.decoder.parameters())
    assert 470e6 < nparam < 490e6
    g = torch.Generator().manual_seed(3)
    text = torch.randn(2, 77, 768, generator=g).to(device)
    imgs = torch.rand(2, 3, 256, 256, generator=g).to(device)

    def loss_fn():
        torch.manual_seed(11)   # fill_mask draws the masked positions
        return muse(text, imgs)

    loss, g0, g1 = _step_twice_bitwise(muse, loss_fn)
    assert torch.isfinite(loss) and 7.0 < float(loss) < 12.0   # ~ln(8192) at random init
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    assert all(not n.startswith("vq.") for n in g0)            # the tokenizer is frozen
    norm, _ = _grad_summary(muse.decoder)
    assert norm > 0


def test_cross_attention_config4_shape_slices_vs_oracle(device):
    """The attention shape only configs[4] has: I = 1024 queries, J = 77 keys (padded text), 16 heads.  Three
    (batch, head) slices of output and gradients against the oracle, with the key-padding mask of a short prompt."""
    from amk import ops

    B, H, I, J, D = 2, 16, 1024, 77, 64
    q, k, v, cot = seeded((B, H, I, D), 1), seeded((B, H, J, D), 2), seeded((B, H, J, D), 3), seeded((B, H, I, D), 4)
    km = torch.ones(B, J, dtype=torch.bool)
    km[1, 30:] = False
    qd, kd, vd = (t.to(device).requires_grad_(True) for t in (q, k, v))
    o = ops.attention(qd, kd, vd, D ** -0.5, key_mask=km.to(device))
    gq, gk, gv = torch.autograd.grad((o * cot.to(device)).sum(), [qd, kd, vd])
    for b, h in ((0, 0), (1, 7), (1, 15)):
        qs, ks, vs = (t[b:b + 1, h:h + 1].clone().requires_grad_(True) for t in (q, k, v))
        want = ref_cpu.attention_core(qs, ks, vs, D ** -0.5, km[b:b + 1], None)
        wq, wk, wv = torch.autograd.grad((want * cot[b:b + 1, h:h + 1]).sum(), [qs, ks, vs])
        assert_close(o[b:b + 1, h:h + 1], want, 2e-5, f"o[{b},{h}]")
        assert_close(gq[b:b + 1, h:h + 1], wq, 2e-5, f"dq[{b},{h}]")
        assert_close(gk[b:b + 1, h:h + 1], wk, 2e-5, f"dk[{b},{h}]")
        assert_close(gv[b:b + 1, h:h + 1], wv, 2e-5, f"dv[{b},{h}]")


def test_vitvqgan_config2_full_size_train_step(device):
    """BASELINE.json configs[2] -- the benchmarked workload -- as ONE whole GAN train step at its own size (dim 256,
    patch 8, 1024 tokens, depth 6 + 6, codebook 8192 x 32, PatchGAN discriminator with gradient penalty), batch 8:
    finite losses; every generator and discriminator parameter receives a finite non-zero gradient (read from Adam's
    first moment, the gradients themselves are zeroed inside the fused update); and, on the reproducible paths
    (attention backward with ordered dq, codebook gradient by ordered sums), two steps from identical state give
    bitwise identical generator gradients / losses up to what the vendor's convolution weight gradients allow --
    the generator phase's inputs come through the discriminator, so the comparison is bitwise for the losses of the
    discriminator phase and the generator's forward, and 1e-5 relative for parameters after the step."""
    import copy

    import bench
    from amk import ops
    from amk.models import ViTVQGAN
    from amk.models.discriminator import NLayerDiscriminator
    from amk.optim import FlatAdam
    from amk.train import VQGANTrainStep

    torch.manual_seed(0)
    model = ViTVQGAN(bench.VIT, bench.CODEBOOK).to(device)
    discr = NLayerDiscriminator(3, 64, 3).to(device)
    model2, discr2 = copy.deepcopy(model), copy.deepcopy(discr)
    imgs = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(device)
    eta = torch.rand(8, 1, 1, 1, generator=torch.Generator().manual_seed(4)).to(device)
    old = ops.DETERMINISTIC_ATTENTION_BACKWARD
    ops.DETERMINISTIC_ATTENTION_BACKWARD = True
    try:
        runs = []
        for m, d in ((model, discr), (model2, discr2)):
            tr = VQGANTrainStep(m, d, warmup_steps=1)
            logs = tr.step(imgs, eta=eta)
            runs.append((tr, {k: v.detach().clone() for k, v in logs.items()}))
    finally:
        ops.DETERMINISTIC_ATTENTION_BACKWARD = old
    (tr, logs), (tr2, logs2) = runs
    for k, v in logs.items():
        assert torch.isfinite(v).all(), k
    assert set(logs) == {"g_loss", "l1", "l2", "codebook_loss", "loss", "d_loss"}
    # every parameter got a gradient: Adam's first moment after one step is (1 - beta1) * clipped gradient
    for opt, mod, what in ((tr.g_optim, model, "generator"), (tr.d_optim, discr, "discriminator")):
        assert isinstance(opt, FlatAdam)
        for n, p in mod.named_parameters():
            st = opt.state_of(p)
            assert st["step"] == 1, f"{what} {n}: no optimizer step (no gradient)"
            m1 = st["exp_avg"]
            assert torch.isfinite(m1).all() and float(m1.abs().max()) > 0.0, f"{what} {n}: zero or non-finite gradient"
            assert torch.isfinite(p).all(), f"{what} {n}"
    # repeatability
    assert torch.equal(logs["d_loss"], logs2["d_loss"]), "discriminator phase: same state, same losses"
    for k in ("l1", "l2", "codebook_loss"):
        assert torch.equal(logs[k], logs2[k]), k
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-5 * max(1.0, float(p.detach().abs().max())) + 2.1e-4, n  # (+ 2 lr: an Adam sign flip)


def test_bf16_gemms_at_the_configs2_layer_shape(device):
    """The mixed-precision GEMMs (csrc/gemm_bf16.hip) at the full configs[2] layer shape (batch 32: M = 32 768 rows, the FFN's
    256 -> 2 x 1368 -> 256): size-independent properties and a float64 check of sampled rows / columns.
      * w12 + SwiGLU: the gate equals silu(a) * b of the (a | b) the same launch wrote (bf16-exactly up to one rounding);
      * dY W3 + gate backward equals the two-launch form bit for bit where the products agree;
      * dW / db: linear in dY (dW(2 dY) = 2 dW(dY) exactly), reproducible, and right on sampled entries."""
    import torch.nn.functional as F

    from amk import dense

    M, Dm, H = 32768, 256, 1368
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, Dm, generator=g).bfloat16().to(device)
    w12 = (torch.randn(2 * H, Dm, generator=g) * Dm ** -0.5).bfloat16().to(device)
    b12 = (torch.randn(2 * H, generator=g) * 0.1).to(device)
    w3 = (torch.randn(Dm, H, generator=g) * H ** -0.5).bfloat16().to(device)
    dy = torch.randn(M, Dm, generator=g).bfloat16().to(device)
    gate, ab = dense.gemm_nt_swiglu_bf16(x, w12, b12)
    assert gate.shape == (M, H) and ab.shape == (M, 2 * H)
    assert torch.isfinite(gate.float()).all() and torch.isfinite(ab.float()).all()
    rows = torch.tensor([0, 1, 127, 128, 4095, 16384, 32767], device=device)
    ref_ab = x[rows].double() @ w12.double().t() + b12.double()
    assert float((ab[rows].double() - ref_ab).abs().max()) <= 2 ** -7 * float(ref_ab.abs().max())
    ref_g = F.silu(ref_ab[:, :H]) * ref_ab[:, H:]
    assert float((gate[rows].double() - ref_g).abs().max()) <= 2 ** -7 * float(ref_g.abs().max())
    gate_only, none = dense.gemm_nt_swiglu_bf16(x, w12, b12, keep_ab=False)
    assert none is None and torch.equal(gate_only, gate)
    dab = dense.gemm_nn_swiglu_bwd_bf16(dy, w3, ab)
    assert dab.shape == (M, 2 * H) and torch.isfinite(dab.float()).all()
    abr = ab[rows].double().requires_grad_(True)
    (ref_dab,) = torch.autograd.grad(F.silu(abr[:, :H]) * abr[:, H:], abr, dy[rows].double() @ w3.double())
    assert float((dab[rows].double() - ref_dab).abs().max()) <= 2e-2 * float(ref_dab.abs().max())
    dw, db = dense.gemm_tn_bf16(dab, x, want_bias=True)
    dw2, db2 = dense.gemm_tn_bf16((2.0 * dab.float()).bfloat16(), x, want_bias=True)   # (doubling is exact in bf16)
    assert torch.equal(dw2, 2.0 * dw) and torch.equal(db2, 2.0 * db)
    again = dense.gemm_tn_bf16(dab, x, want_bias=True)
    assert torch.equal(again[0], dw) and torch.equal(again[1], db)
    cols = torch.tensor([0, 1, 1367, 1368, 2735], device=device)
    ref_dw = dab[:, cols].double().t() @ x.double()
    assert float((dw[cols].double() - ref_dw).abs().max()) <= 2e-5 * float(ref_dw.abs().max())
    assert float((db[cols].double() - dab[:, cols].double().sum(0)).abs().max()) <= 2e-5 * float(db.abs().max())
