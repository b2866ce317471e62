"""HIP attention kernels (through the C ABI) against the CPU oracle and the golden vectors.

Tolerance: the north star asks for fp32 outputs within 1e-4 relative (max |a-b| / max |b|)
of the reference CPU path; the kernels use exact-f32 MFMA, so 2e-5 is asserted here.
"""
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded, seeded_params
from util import rel_err, assert_close, load_golden, weights_of

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(params=["fused", "fused-kept-128", "fused-recompute-128", "fused-recompute-256", "deterministic",
                        "deterministic-128", "two-kernel", "fused-x6fwd"], autouse=True)
def backward_path(request):
    """Every test runs with every backward path of the attention core (amk_attn_bwd `stages`): the fused
    pass reading the scores the forward kept (the default) or recomputing them, with 128 or 256 keys per
    workgroup; the same pass with reproducible dq (direct stores / per-key-block partials); the two
    reproducible recompute kernels; and the split-bf16 forward (amk_attn_fwd_x6) in front of the fused backward."""
    from amk import ops

    old = (ops.DETERMINISTIC_ATTENTION_BACKWARD, ops.ATTENTION_FORWARD, ops.ATTENTION_KEEP_SCORES,
           ops.ATTENTION_BACKWARD_KEYS, ops.ATTENTION_BACKWARD_TWO_KERNEL)
    ops.DETERMINISTIC_ATTENTION_BACKWARD = request.param.startswith("deterministic")
    ops.ATTENTION_BACKWARD_TWO_KERNEL = request.param == "two-kernel"
    ops.ATTENTION_FORWARD = "bf16x6" if request.param.endswith("x6fwd") else "f32"
    ops.ATTENTION_KEEP_SCORES = "recompute" not in request.param
    ops.ATTENTION_BACKWARD_KEYS = 128 if request.param.endswith("128") else (256 if request.param.endswith("256") else 0)
    yield request.param
    (ops.DETERMINISTIC_ATTENTION_BACKWARD, ops.ATTENTION_FORWARD, ops.ATTENTION_KEEP_SCORES,
     ops.ATTENTION_BACKWARD_KEYS, ops.ATTENTION_BACKWARD_TWO_KERNEL) = old


def _core_case(device, B, H, I, J, key_mask=None, causal=None, seed=0, layout="bthd", D=64):
    from amk import ops

    q = seeded((B, H, I, D), seed + 1)
    k = seeded((B, H, J, D), seed + 2)
    v = seeded((B, H, J, D), seed + 3)
    cot = seeded((B, H, I, D), seed + 4)
    scale = D ** -0.5
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref = ref_cpu.attention_core(qc, kc, vc, scale, key_mask, causal)
    g_ref = torch.autograd.grad((o_ref * cot).sum(), [qc, kc, vc])

    def to_dev(t):
        t = t.to(device)
        if layout == "bthd":  # (B,T,H,D) storage viewed as (B,H,T,D): the projection layout
            t = t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        return t.requires_grad_(True)

    qd, kd, vd = to_dev(q), to_dev(k), to_dev(v)
    km = key_mask.to(device) if key_mask is not None else None
    cm = causal.to(device) if causal is not None else None
    o = ops.attention(qd, kd, vd, scale, key_mask=km, causal_mask=cm)
    g = torch.autograd.grad((o * cot.to(device)).sum(), [qd, kd, vd])
    assert_close(o, o_ref, TOL, "o")
    for name, a, b in zip(("dq", "dk", "dv"), g, g_ref):
        if J == 1 and name != "dv":
            # one key: P = 1 whatever q and k are, so dq = dk = 0 exactly in real arithmetic; both sides
            # return rounding noise of (dP - delta), which a relative comparison cannot judge
            assert float(a.abs().max()) <= 1e-6 and float(b.abs().max()) <= 1e-6, name
            continue
        assert_close(a, b, TOL, name)


@pytest.mark.parametrize(
    "B,H,I,J",
    [(1, 1, 32, 32), (2, 3, 128, 128), (1, 2, 65, 65), (2, 2, 40, 77), (1, 1, 1, 1), (1, 2, 200, 130), (1, 1, 1024, 1024)],
)
@pytest.mark.parametrize("layout", ["bthd", "bhtd"])
def test_core_unmasked(device, B, H, I, J, layout):
    _core_case(device, B, H, I, J, seed=I * 7 + J, layout=layout)


def test_core_key_mask(device):
    B, H, I, J = 2, 2, 70, 77
    km = torch.ones(B, J, dtype=torch.bool)
    km[0, -17:] = False
    km[1, ::3] = False
    _core_case(device, B, H, I, J, key_mask=km, seed=5)


def test_core_causal(device):
    B, H, T = 2, 2, 100
    causal = torch.ones(T, T).triu(1).bool()
    _core_case(device, B, H, T, T, causal=causal, seed=6)


def test_core_both_masks_and_dead_row(device):
    """A fully masked query row gives a uniform softmax over ALL keys (fill is -1e9, not -inf)
    and passes no gradient to q/k (reference semantics, SURVEY.md section 7 'hard parts')."""
    B, H, T = 1, 2, 96
    causal = torch.ones(T, T).triu(1).bool()
    causal[5, :] = True
    causal[70, :] = True
    km = torch.ones(B, T, dtype=torch.bool)
    km[0, 10:20] = False
    _core_case(device, B, H, T, T, key_mask=km, causal=causal, seed=7)


def test_core_all_keys_masked(device):
    B, H, I, J = 1, 1, 33, 45
    km = torch.zeros(B, J, dtype=torch.bool)
    _core_case(device, B, H, I, J, key_mask=km, seed=8)


def test_core_softmax_spike(device):
    """Forces the online-softmax rescale: one key dominates late in the sequence."""
    from amk import ops

    B, H, I, J, D = 1, 1, 64, 320, 64
    q = seeded((B, H, I, D), 91)
    k = seeded((B, H, J, D), 92)
    v = seeded((B, H, J, D), 93)
    k[0, 0, 300] = q[0, 0, 7] * 4.0   # score ~ 4*|q|^2/8 >> the rest, in the last tile
    k[0, 0, 3] = q[0, 0, 9] * 3.0     # and an early spike for another row
    o_ref = ref_cpu.attention_core(q, k, v, D ** -0.5)
    o = ops.attention(q.to(device), k.to(device), v.to(device), D ** -0.5)
    assert_close(o, o_ref, TOL, "o")


@pytest.mark.parametrize("D", [64, 32, 128])
def test_lazy_reference_moves_in_every_tile(device, backward_path, D):
    """The unmasked forwards keep a LAZY softmax reference (it moves only when a row's tile maximum passes it by 2^8,
    csrc/attn_fwd.hip attn_fwd_plain_kernel): scores that climb steeply along the keys force the move -- the rescale
    branch and the rewrite of the reference held in the MFMA accumulators -- in every tile, for some rows by much more
    than the threshold, for others by less (no move: p up to 2^8).  Ragged sizes, so the peeled last tile is exercised
    too; outputs and all gradients against the oracle, whose statistics are the true maxima."""
    from amk import ops

    B, H, I, J = 2, 2, 70, 333
    q = seeded((B, H, I, D), 71)
    k = seeded((B, H, J, D), 72) * 0.3
    v = seeded((B, H, J, D), 73)
    cot = seeded((B, H, I, D), 74)
    u = torch.nn.functional.normalize(seeded((D,), 75), dim=0)
    # a STAIRCASE, one step per 64-key tile: the scores climb from tile to tile (the reference has to move) but stay
    # diffuse inside a tile, so the softmax does not collapse onto one key (a one-hot row makes dP - delta cancel
    # catastrophically in any f32 backward, the reference's included)
    ramp = (torch.arange(J) // 64).float().view(1, 1, J, 1) / (J // 64)
    k = k + ramp * u * 70.0                              # q . k grows with the key tile along u ...
    q = q + u * torch.linspace(0.0, 6.0, I).view(1, 1, I, 1)   # ... by an amount that differs from row to row
    scale = D ** -0.5
    # (the oracle in float64 here: at |s| ~ 100 an f32 oracle's own rounding of s is as large as the kernel's)
    qc, kc, vc = (t.double().requires_grad_(True) for t in (q, k, v))
    o_ref = ref_cpu.attention_core(qc, kc, vc, scale)
    g_ref = [g.float() for g in torch.autograd.grad((o_ref * cot.double()).sum(), [qc, kc, vc])]
    o_ref = o_ref.detach().float()
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale * 1.4426950408889634
    climb = float((s[..., -1] - s[..., 0]).abs().max())
    assert climb > 64.0, climb                               # many times the 2^8 threshold across the sequence
    qd, kd, vd = (t.to(device).requires_grad_(True) for t in (q, k, v))
    o = ops.attention(qd, kd, vd, scale)
    g = torch.autograd.grad((o * cot.to(device)).sum(), [qd, kd, vd])
    assert_close(o, o_ref, TOL, "o")
    # gradients: scores of magnitude ~100 in the log2 domain carry an f32 rounding of ~1e-5 each, which the backward's
    # P (dP - delta) and the long k rows (|k| ~ 90) amplify: 1-4e-4 against float64 for this library's kernels before and
    # after the lazy reference alike (AMK_ATTN_FWD_PLAIN=0) -- the input's conditioning, not the forward under test
    for name, a, b in zip(("dq", "dk", "dv"), g, g_ref):
        assert rel_err(a, b) < 1e-3, (name, rel_err(a, b))


@pytest.mark.parametrize(
    "variant", ["self", "self_keymask", "self_causal", "self_both", "self_deadrow", "cross", "cross_ctxmask"]
)
def test_module_matches_reference_golden(device, variant):
    """amk.models.SoftmaxAttention with the reference's own weights vs the reference's outputs."""
    from amk.models import SoftmaxAttention

    fx = load_golden("softmax_attention")
    dim, h, d = (int(v) for v in fx["dims"])
    m = SoftmaxAttention(dim, h, d)
    missing = m.load_state_dict(weights_of(fx), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m = m.to(device)
    x = torch.from_numpy(fx["x"]).to(device).requires_grad_(True)
    ctx = torch.from_numpy(fx["context"]).to(device).requires_grad_(True)
    cot = torch.from_numpy(fx["cot"]).to(device)
    t = lambda name: torch.from_numpy(fx[name]).to(device)
    kw = {
        "self": {},
        "self_keymask": dict(context_mask=t("keymask")),
        "self_causal": dict(causal_mask=t("causal")),
        "self_both": dict(causal_mask=t("causal"), context_mask=t("keymask")),
        "self_deadrow": dict(causal_mask=t("dead")),
        "cross": dict(context=ctx),
        "cross_ctxmask": dict(context=ctx, context_mask=t("ctxmask")),
    }[variant]
    out = m(x, **kw)
    assert_close(out, fx[f"{variant}:out"], TOL, "out")
    params = dict(m.named_parameters())
    names = sorted(params)
    wrt = [x] + ([ctx] if "context" in kw else []) + [params[n] for n in names]
    gs = torch.autograd.grad((out * cot).sum(), wrt)
    assert_close(gs[0], fx[f"{variant}:gx"], TOL, "grad x")
    off = 1
    if "context" in kw:
        assert_close(gs[1], fx[f"{variant}:gctx"], TOL, "grad context")
        off = 2
    for n, g in zip(names, gs[off:]):
        if f"{variant}:g:{n}" in fx:  # parameter gradients are stored for two of the variants
            assert_close(g, fx[f"{variant}:g:{n}"], TOL, f"grad {n}")


def test_module_config1(device):
    """BASELINE.json configs[0]: SoftmaxAttention dim 512, h 16, d 64, (B 2, T 128)."""
    from amk.models import SoftmaxAttention

    fx = load_golden("softmax_attention_c1")
    dim, h, d, B, T = (int(v) for v in fx["dims"])
    s_w, s_x, s_c = (int(v) for v in fx["seeds"])
    m = SoftmaxAttention(dim, h, d)
    shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
    m.load_state_dict(seeded_params(shapes, s_w))
    m = m.to(device)
    x = seeded((B, T, dim), s_x).to(device).requires_grad_(True)
    cot = seeded((B, T, dim), s_c).to(device)
    out = m(x)
    (gx,) = torch.autograd.grad((out * cot).sum(), [x])
    assert tuple(out.shape) == (B, T, dim)          # README.md:108 shape contract
    assert_close(out[:, ::4], fx["out_s4"], TOL, "out")
    assert_close(gx[:, ::4], fx["gx_s4"], TOL, "grad x")


def test_full_size_properties(device):
    """C3 layer size (B 8, h 8, T 1024): size-independent checks instead of a CPU oracle run.
    (1) rows of P sum to 1: with v = ones the output is exactly ones up to rounding;
    (2) linearity in v; (3) determinism (bitwise equal across two launches)."""
    from amk import ops

    B, H, T, D = 8, 8, 1024, 64
    g = torch.Generator(device="cpu").manual_seed(1234)
    q = torch.randn(B, H, T, D, generator=g).to(device)
    k = torch.randn(B, H, T, D, generator=g).to(device)
    v1 = torch.randn(B, H, T, D, generator=g).to(device)
    v2 = torch.randn(B, H, T, D, generator=g).to(device)
    s = D ** -0.5
    ones = ops.attention(q, k, torch.ones_like(v1), s)
    assert float((ones - 1).abs().max()) < 1e-5
    a = ops.attention(q, k, v1, s)
    b = ops.attention(q, k, v2, s)
    ab = ops.attention(q, k, v1 + 2 * v2, s)
    assert_close(ab, a + 2 * b, 1e-5, "linearity in v")
    assert torch.equal(a, ops.attention(q, k, v1, s))
    # backward determinism: the two-kernel path is bitwise reproducible; the fused path is in dk, dv
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v1))
    g1 = torch.autograd.grad(ops.attention(qg, kg, vg, s).square().sum(), [qg, kg, vg])
    g2 = torch.autograd.grad(ops.attention(qg, kg, vg, s).square().sum(), [qg, kg, vg])
    assert torch.equal(g1[1], g2[1]) and torch.equal(g1[2], g2[2])
    if ops.DETERMINISTIC_ATTENTION_BACKWARD:
        assert torch.equal(g1[0], g2[0])
    else:
        assert_close(g1[0], g2[0], 1e-5, "dq run-to-run")
    # one (batch, head) slice against the oracle
    o_ref = ref_cpu.attention_core(q[3:4, 5:6].cpu(), k[3:4, 5:6].cpu(), v1[3:4, 5:6].cpu(), s)
    assert_close(a[3:4, 5:6], o_ref, TOL, "slice vs oracle")


@pytest.mark.parametrize("B,H,I,J", [(2, 2, 200, 130), (1, 2, 1024, 1024), (2, 1, 33, 300)])
def test_kept_scores_equal_recomputed_scores(device, backward_path, B, H, I, J):
    """The scores amk_attn_fwd_keep leaves are bit for bit what the fused backward would recompute: dk and
    dv (no atomics) must be IDENTICAL between amk_attn_bwd_kept and amk_attn_bwd, for both workgroup sizes;
    dq may differ in the order of its atomic adds only."""
    if backward_path != "fused":
        pytest.skip("one pass is enough: the test drives every variant itself")
    from amk import ops

    D = 64
    mk = lambda seed, T: seeded((B, T, H, D), seed).to(device).permute(0, 2, 1, 3)
    q, k, v, d_o = mk(1, I), mk(2, J), mk(3, J), mk(4, I)
    km = torch.ones(B, J, dtype=torch.uint8)
    km[0, 5::7] = 0
    km = km.to(device)
    scale = D ** -0.5
    q, k, v, o, stats, scores = ops._attn_forward(q, k, v, km, None, scale, keep_scores=True)
    assert scores is not None
    outs = {}
    for keys in (16, 32):  # AMK_ATTN_BWD_KEYS128 / KEYS256
        for kept in (False, True):
            dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
            ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, km, None, scale, stages=9 | keys,
                               scores=scores if kept else None)
            outs[(keys, kept)] = (dq, dk, dv)
    ref = outs[(16, False)]
    for key, (dq, dk, dv) in outs.items():
        assert torch.equal(dk, outs[(key[0], False)][1]) and torch.equal(dv, outs[(key[0], False)][2]), key
        assert_close(dq, ref[0], 2e-6, f"dq {key}")
        assert_close(dk, ref[1], 2e-6, f"dk {key}")
        assert_close(dv, ref[2], 2e-6, f"dv {key}")


@pytest.mark.parametrize("D", [32, 128])
@pytest.mark.parametrize("B,H,I,J,masks", [(2, 3, 128, 128, ""), (1, 2, 65, 77, "k"), (2, 2, 100, 100, "c"),
                                           (1, 2, 200, 130, "kc"), (1, 1, 1, 1, ""), (1, 2, 300, 40, "k"),
                                           (1, 2, 300, 700, ""), (2, 1, 333, 520, "k"), (1, 2, 260, 300, "c")])
def test_core_other_head_dims(device, backward_path, D, B, H, I, J, masks):
    """dim_head 32 and 128 (the reference takes any dim_head, models/softmax_attention.py:23) against the oracle, every
    mask combination, ragged sizes, several key blocks per (batch, head).  Paths: "fused" = the one-pass backward of
    csrc/attn_bwd_fused_gen.hip (round 4) -- reading the scores the unmasked forward kept, or recomputing them
    ("fused-recompute-*", and every masked call) --; "deterministic" and "two-kernel" = the two reproducible recompute
    kernels of csrc/attn_generic.hip."""
    if backward_path not in ("fused", "fused-recompute-256", "deterministic", "two-kernel"):
        pytest.skip("head dims other than 64: one-pass (kept / recomputed scores) or the reproducible recompute pair")
    km = None
    if "k" in masks:
        km = torch.ones(B, J, dtype=torch.bool)
        km[0, -J // 4:] = False
        km[-1, ::3] = False
    causal = torch.ones(I, J).triu(1).bool() if "c" in masks else None
    for layout in ("bthd", "bhtd"):
        _core_case(device, B, H, I, J, key_mask=km, causal=causal, seed=D + I, layout=layout, D=D)


@pytest.mark.parametrize("dim_head", [32, 128])
def test_module_other_head_dims(device, backward_path, dim_head):
    """SoftmaxAttention(dim, heads, dim_head != 64): self and cross attention against the oracle module."""
    if backward_path != "fused":
        pytest.skip("one pass")
    from amk.models import SoftmaxAttention

    torch.manual_seed(0)
    m = SoftmaxAttention(96, num_heads=3, dim_head=dim_head)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = seeded((2, 50, 96), 1)
    ctx = seeded((2, 33, 96), 2)
    cot = seeded((2, 50, 96), 3)
    km = torch.ones(2, 33, dtype=torch.bool)
    km[1, 20:] = False
    m = m.to(device)
    for context, mask in ((None, None), (ctx, km)):
        xr = x.clone().requires_grad_(True)
        want = ref_cpu.softmax_attention(xr, w, 3, dim_head, context=context, context_mask=mask)
        (gx_ref,) = torch.autograd.grad((want * cot).sum(), [xr])
        xd = x.to(device).requires_grad_(True)
        got = m(xd, context=None if context is None else context.to(device),
                context_mask=None if mask is None else mask.to(device))
        (gx,) = torch.autograd.grad((got * cot.to(device)).sum(), [xd])
        assert_close(got, want, TOL, "output")
        assert_close(gx, gx_ref, TOL, "grad x")


@pytest.mark.parametrize("B,H,I,J", [(2, 2, 200, 130), (1, 2, 1024, 1024), (2, 1, 300, 77), (1, 1, 40, 600)])
def test_reproducible_fused_backward(device, backward_path, B, H, I, J):
    """AMK_ATTN_BWD_DQ_REPRO: the one-pass backward without atomics -- dq stored directly (all keys in one
    workgroup: J = 77, 130 at 256 keys per workgroup) or as per-key-block partials summed in order -- gives
    bit-identical dq, dk, dv on every run, equal to the atomics path within rounding."""
    if backward_path != "fused":
        pytest.skip("one pass is enough: the test drives every variant itself")
    from amk import ops

    D = 64
    mk = lambda seed, T: seeded((B, T, H, D), seed).to(device).permute(0, 2, 1, 3)
    q, k, v, d_o = mk(1, I), mk(2, J), mk(3, J), mk(4, I)
    scale = D ** -0.5
    q, k, v, o, stats, scores = ops._attn_forward(q, k, v, None, None, scale, keep_scores=True)

    def run(stages, sc):
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (q, k, v))
        ops._attn_backward(q, k, v, o, stats, d_o, dq, dk, dv, None, None, scale, stages=stages, scores=sc)
        return dq, dk, dv

    ref = run(9, scores)                       # atomics
    for keys in (16, 32):
        for sc in (scores, None):
            a, b_ = run(73 | keys, sc), run(73 | keys, sc)
            for x, y, z, name in zip(a, b_, ref, ("dq", "dk", "dv")):
                assert torch.equal(x, y), (name, keys)
                assert_close(x, z, 3e-6, f"{name} keys bit {keys}")


def test_torch_deterministic_mode_selects_reproducible_backward(device, backward_path):
    """torch.use_deterministic_algorithms(True) -- PyTorch's switch for atomics-based backward kernels -- makes
    ops.attention take the reproducible dq path: two backward passes give bit-identical gradients."""
    if backward_path != "fused":
        pytest.skip("one pass")
    from amk import ops

    q, k, v = (seeded((2, 4, 1024, 64), s).to(device).requires_grad_(True) for s in (1, 2, 3))
    cot = seeded((2, 4, 1024, 64), 4).to(device)
    assert not ops.DETERMINISTIC_ATTENTION_BACKWARD
    torch.use_deterministic_algorithms(True)
    try:
        g1 = torch.autograd.grad((ops.attention(q, k, v, 0.125) * cot).sum(), [q, k, v])
        g2 = torch.autograd.grad((ops.attention(q, k, v, 0.125) * cot).sum(), [q, k, v])
    finally:
        torch.use_deterministic_algorithms(False)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


def test_core_causal_many_tiles_ragged(device):
    """Causal and arbitrary (I, J) masks over several 32-query tiles and key blocks, ragged ends: the one-pass backward
    takes the mask itself (csrc/attn_bwd_fused.hip, CAUSAL) instead of falling back to the two recompute kernels."""
    B, H, I, J = 2, 2, 300, 333
    g = torch.Generator().manual_seed(11)
    causal = torch.ones(I, J).triu(1).bool() | (torch.rand(I, J, generator=g) < 0.1)
    causal[17, :] = True   # a dead row
    km = torch.ones(B, J, dtype=torch.bool)
    km[1, 5::7] = False
    _core_case(device, B, H, I, J, key_mask=km, causal=causal, seed=12)
    _core_case(device, B, H, I, J, causal=causal, seed=13)
