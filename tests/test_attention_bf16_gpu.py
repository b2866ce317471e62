"""The bf16-MFMA attention core (csrc/attn_bf16.hip) and SoftmaxAttention under torch.autocast(bfloat16) -- the
reference's shipped training precision (cfg/vitvqgan.yaml:73).

Checker: fixtures generated from the reference's SoftmaxAttention under torch.autocast("cpu", dtype=bfloat16)
(tests/golden/softmax_attention_bf16.npz, oracle/gen_golden.py:gen_softmax_attention_bf16), stored next to the same
module's f32 results.  The fixture says what tolerance the mode supports: the reference's own autocast output differs
from its own f32 output by 6e-3 .. 9e-3 (max |a - b| / max |b|; golden_meta.json).  The bar here: within 2e-2 of the
autocast fixture, and no further from the f32 result than 1.5x what the reference's own autocast is.
The core alone (no projections) is also checked against an f32 computation on the bf16-rounded operands at 5e-3."""
import json
import os

import pytest
import torch

from oracle.fixture_recipe import seeded
from util import GOLDEN, load_golden, rel_err, weights_of

pytestmark = pytest.mark.gpu


def _core_ref(q, k, v, scale):
    """f32 attention on (B,H,T,D) tensors."""
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v)


@pytest.mark.parametrize("B,H,I,J", [(1, 1, 32, 64), (2, 3, 100, 77), (1, 2, 256, 300), (2, 8, 1024, 1024)])
def test_bf16_core_forward_backward(device, B, H, I, J):
    from amk import ops

    D = 64
    q2 = seeded((B, I, H * D), 1).bfloat16()
    kv2 = seeded((B, J, 2 * H * D), 2).bfloat16()
    cot = seeded((B, I, H * D), 3).bfloat16()
    qr, kvr = q2.float().requires_grad_(True), kv2.float().requires_grad_(True)
    q = qr.view(B, I, H, D).permute(0, 2, 1, 3)
    kv = kvr.view(B, J, 2, H, D)
    o_ref = _core_ref(q, kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3), D ** -0.5)
    o_ref2 = o_ref.permute(0, 2, 1, 3).reshape(B, I, H * D)
    gq, gkv = torch.autograd.grad((o_ref2 * cot.float()).sum(), [qr, kvr])

    qd, kvd = q2.to(device).requires_grad_(True), kv2.to(device).requires_grad_(True)
    o = ops.attention_fused_kv(qd, kvd, H, D, D ** -0.5)
    assert o.dtype == torch.bfloat16
    dq, dkv = torch.autograd.grad((o.float() * cot.to(device).float()).sum(), [qd, kvd])
    assert rel_err(o.float(), o_ref2) < 5e-3          # bf16 rounding of P and of the output
    assert rel_err(dq.float(), gq) < 1e-2
    assert rel_err(dkv.float(), gkv) < 1e-2
    # reproducible: no atomics anywhere
    dq2, dkv2 = torch.autograd.grad((ops.attention_fused_kv(qd, kvd, H, D, D ** -0.5).float() * cot.to(device).float()).sum(), [qd, kvd])
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2)


def test_bf16_lazy_reference_moves(device):
    """Scores that climb by a step per 64-key tile: the unmasked bf16 forward's lazy softmax reference has to move in every
    tile (the rescale branch), ragged sizes; against f32 on the bf16-rounded operands."""
    from amk import ops

    B, H, I, J, D = 2, 2, 70, 333, 64
    u = torch.nn.functional.normalize(seeded((D,), 75), dim=0)
    q = seeded((B, I, H, D), 71) + u * torch.linspace(0.0, 4.0, I).view(1, I, 1, 1)
    k = seeded((B, J, H, D), 72) * 0.3 + ((torch.arange(J) // 64).float() / (J // 64)).view(1, J, 1, 1) * u * 60.0
    v = seeded((B, J, H, D), 73)
    q2 = q.reshape(B, I, H * D).bfloat16()
    kv2 = torch.stack([k, v], dim=2).reshape(B, J, 2 * H * D).bfloat16()
    qr, kvr = q2.float(), kv2.float().view(B, J, 2, H, D)
    o_ref = _core_ref(qr.view(B, I, H, D).permute(0, 2, 1, 3), kvr[:, :, 0].permute(0, 2, 1, 3), kvr[:, :, 1].permute(0, 2, 1, 3), D ** -0.5)
    o = ops.attention_fused_kv(q2.to(device), kv2.to(device), H, D, D ** -0.5)
    assert rel_err(o.float(), o_ref.permute(0, 2, 1, 3).reshape(B, I, H * D)) < 5e-3


def _masked_ref(q, k, v, scale, key_mask, causal_mask):
    """models/softmax_attention.py:62-76 in f32 on (B,H,T,D) tensors: masked_fill(-1e9), context_mask True = keep,
    causal_mask True = masked."""
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], -1e9)
    if causal_mask is not None:
        s = s.masked_fill(causal_mask[None, None], -1e9)
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v)


@pytest.mark.parametrize("B,H,I,J,kind", [(2, 2, 64, 64, "key"), (2, 3, 100, 77, "key"), (1, 2, 130, 130, "causal"),
                                           (2, 2, 96, 300, "both"), (1, 1, 40, 70, "dead"), (2, 8, 1024, 77, "key"),
                                           (1, 4, 512, 512, "causal")])
def test_bf16_core_with_masks(device, B, H, I, J, kind):
    """The <MASKED> variants of the bf16 kernels against an f32 computation on the bf16-rounded operands: key-padding
    mask (the Muse text prompt: I 1024 x J 77), causal mask, both, and rows whose every key is masked (uniform weights,
    as masked_fill(-1e9) gives them).  The launches are the bf16 kernels (asserted by name), not the f32 ones on upcasts."""
    from amk import ops

    D = 64
    g = torch.Generator().manual_seed(B * 1000 + I + J)
    q2 = seeded((B, I, H * D), 1).bfloat16()
    kv2 = seeded((B, J, 2 * H * D), 2).bfloat16()
    cot = seeded((B, I, H * D), 3).bfloat16()
    key_mask = causal = None
    if kind in ("key", "both", "dead"):
        key_mask = torch.rand(B, J, generator=g) > 0.3
        key_mask[:, 0] = True
        if kind == "dead":
            key_mask[0] = False                       # every key of batch 0 masked: uniform softmax over all J
    if kind in ("causal", "both"):
        causal = torch.ones(I, J, dtype=torch.bool).triu(1)
    qr, kvr = q2.float().requires_grad_(True), kv2.float().requires_grad_(True)
    q = qr.view(B, I, H, D).permute(0, 2, 1, 3)
    kv = kvr.view(B, J, 2, H, D)
    o_ref = _masked_ref(q, kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3), D ** -0.5, key_mask, causal)
    o_ref2 = o_ref.permute(0, 2, 1, 3).reshape(B, I, H * D)
    gq, gkv = torch.autograd.grad((o_ref2 * cot.float()).sum(), [qr, kvr])

    qd, kvd = q2.to(device).requires_grad_(True), kv2.to(device).requires_grad_(True)
    km = key_mask.to(device) if key_mask is not None else None
    cm = causal.to(device) if causal is not None else None
    ops.KERNEL_EVENTS = {}
    try:
        o = ops.attention_fused_kv(qd, kvd, H, D, D ** -0.5, key_mask=km, causal_mask=cm)
        dq, dkv = torch.autograd.grad((o.float() * cot.to(device).float()).sum(), [qd, kvd])
        torch.cuda.synchronize()
        assert set(ops.KERNEL_EVENTS) == {"attn_bf16_fwd_kernel<masked>", "attn_bf16_bwd_kernel<masked>"}, set(ops.KERNEL_EVENTS)
    finally:
        ops.KERNEL_EVENTS = None
    assert o.dtype == torch.bfloat16
    assert rel_err(o.float(), o_ref2) < 5e-3
    assert rel_err(dq.float(), gq) < 1e-2
    assert rel_err(dkv.float(), gkv) < 1e-2
    if kind == "dead":   # the dead batch: exactly the mean of v over all keys, and no gradient through the scores
        vmean = kv2.float().view(B, J, 2, H, D)[0, :, 1].mean(0).reshape(1, H * D)
        assert rel_err(o[0].float().cpu(), vmean.expand(I, -1)) < 5e-3
        assert float(dq[0].float().abs().max()) == 0.0


@pytest.mark.parametrize("variant", ["self", "self_keymask", "cross_ctxmask"])
def test_module_under_autocast_matches_reference_autocast(device, variant):
    from amk import ops
    from amk.models import SoftmaxAttention

    fx = load_golden("softmax_attention_bf16")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["softmax_attention_bf16"]
    ref_err = meta["reference_autocast_vs_reference_f32"][variant]
    dim, h, d = (int(v) for v in fx["dims"])
    m = SoftmaxAttention(dim, h, d).to(device)
    m.load_state_dict({k: v.to(device) for k, v in weights_of(fx).items()}, strict=True)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    kw = {}
    wrt = [x]
    if variant == "self_keymask":
        kw["context_mask"] = torch.from_numpy(fx["keymask"]).to(device)
    if variant == "cross_ctxmask":
        kw["context"] = torch.from_numpy(fx["context"]).to(device).requires_grad_(True)
        kw["context_mask"] = torch.from_numpy(fx["ctxmask"]).to(device)
        wrt.append(kw["context"])
    ops.KERNEL_EVENTS = {}
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x, **kw)
        gs = torch.autograd.grad((out.float() * torch.from_numpy(fx["cot"]).to(device)).sum(), wrt + [p for _, p in sorted(m.named_parameters())])
        torch.cuda.synchronize()
        names = {n for n in ops.KERNEL_EVENTS if n.startswith("attn")}
    finally:
        ops.KERNEL_EVENTS = None
    # the core ran on the bf16-MFMA kernels, masked calls included (no upcast to the f32 kernels)
    sfx = "<masked>" if "context_mask" in kw else ""
    assert names == {"attn_bf16_fwd_kernel" + sfx, "attn_bf16_bwd_kernel" + sfx}, names
    want, want32 = torch.from_numpy(fx[f"{variant}:out"]), torch.from_numpy(fx[f"{variant}:out_f32"])
    assert rel_err(out.float(), want) < 2e-2
    assert rel_err(out.float(), want32) < 1.5 * ref_err["err_out"]
    assert rel_err(gs[0], torch.from_numpy(fx[f"{variant}:gx"])) < 2e-2
    assert rel_err(gs[0], torch.from_numpy(fx[f"{variant}:gx_f32"])) < 1.5 * ref_err["err_gx"]
    off = 2 if "context" in kw else 1
    for (n, _), g in zip(sorted(m.named_parameters()), gs[off:]):
        assert rel_err(g, torch.from_numpy(fx[f"{variant}:g:{n}"])) < 2e-2, n
        assert rel_err(g, torch.from_numpy(fx[f"{variant}:g32:{n}"])) < 1.5 * ref_err["err_gparams"], n
