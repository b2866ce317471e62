"""Pins the CPU oracle (oracle/ref_cpu.py) to the golden vectors that oracle/gen_golden.py
produced from the reference's own source files.  CPU only (-m "not gpu").

Tolerances: the oracle restates the same fp32 op sequence, so values agree to a few ulps;
indices (VQ codes, MoE / SwitchHead expert ids) must be bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.fixture_recipe import seeded, seeded_params
from util import assert_close, load_golden, weights_of

TIGHT = 2e-6


def _grads(out, cot, wrt):
    return torch.autograd.grad((out * cot).sum(), wrt, allow_unused=True)


@pytest.mark.parametrize(
    "variant", ["self", "self_keymask", "self_causal", "self_both", "self_deadrow", "cross", "cross_ctxmask"]
)
def test_softmax_attention_matches_reference(variant):
    fx = load_golden("softmax_attention")
    dim, h, d = (int(v) for v in fx["dims"])
    w = {k: v.clone().requires_grad_(True) for k, v in weights_of(fx).items()}
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    ctx = torch.from_numpy(fx["context"]).requires_grad_(True)
    cot = torch.from_numpy(fx["cot"])
    kw = {
        "self": {},
        "self_keymask": dict(context_mask=torch.from_numpy(fx["keymask"])),
        "self_causal": dict(causal_mask=torch.from_numpy(fx["causal"])),
        "self_both": dict(causal_mask=torch.from_numpy(fx["causal"]), context_mask=torch.from_numpy(fx["keymask"])),
        "self_deadrow": dict(causal_mask=torch.from_numpy(fx["dead"])),
        "cross": dict(context=ctx),
        "cross_ctxmask": dict(context=ctx, context_mask=torch.from_numpy(fx["ctxmask"])),
    }[variant]
    out = ref_cpu.softmax_attention(x, w, h, d, **kw)
    assert_close(out, fx[f"{variant}:out"], TIGHT, "out")
    names = sorted(w)
    wrt = [x] + ([ctx] if "context" in kw else []) + [w[n] for n in names]
    gs = _grads(out, cot, wrt)
    assert_close(gs[0], fx[f"{variant}:gx"], TIGHT, "grad x")
    off = 1
    if "context" in kw:
        assert_close(gs[1], fx[f"{variant}:gctx"], TIGHT, "grad context")
        off = 2
    for n, g in zip(names, gs[off:]):
        if f"{variant}:g:{n}" in fx:  # parameter gradients are stored for two of the variants
            assert_close(g, fx[f"{variant}:g:{n}"], TIGHT, f"grad {n}")


def test_softmax_attention_config1_seeded():
    """BASELINE.json configs[0]: dim 512, h 16, d 64, (B 2, T 128), weights from seeds."""
    fx = load_golden("softmax_attention_c1")
    dim, h, d, B, T = (int(v) for v in fx["dims"])
    s_w, s_x, s_c = (int(v) for v in fx["seeds"])
    shapes = {"q.0.weight": (h * d, dim), "kv.0.weight": (2 * h * d, dim), "W_o.weight": (dim, h * d), "W_o.bias": (dim,)}
    w = seeded_params(shapes, s_w)
    x = seeded((B, T, dim), s_x).requires_grad_(True)
    cot = seeded((B, T, dim), s_c)
    out = ref_cpu.softmax_attention(x, w, h, d)
    (gx,) = _grads(out, cot, [x])
    assert_close(out[:, ::4], fx["out_s4"], TIGHT, "out")
    assert_close(gx[:, ::4], fx["gx_s4"], TIGHT, "grad x")


def test_codebook_small_matches_reference():
    fx = load_golden("codebook_small")
    E = torch.from_numpy(fx["E"]).requires_grad_(True)
    z = torch.from_numpy(fx["z"]).requires_grad_(True)
    cot = torch.from_numpy(fx["cot"])
    zq, idx, loss = ref_cpu.codebook_forward(z, E, 0.25)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == tuple(fx["idx"].shape)
    assert np.array_equal(idx.numpy(), fx["idx"])          # bit-exact indices
    assert_close(zq, fx["zq"], TIGHT, "z_q")
    assert_close(loss, fx["loss"], TIGHT, "loss")
    gz, gE = torch.autograd.grad((zq * cot).sum() + float(fx["loss_weight"]) * loss, [z, E])
    assert_close(gz, fx["gz"], TIGHT, "grad z")
    assert_close(gE, fx["gE"], TIGHT, "grad codebook")
    assert_close(ref_cpu.codebook_margin(z.detach(), E.detach()), fx["margin"], 1e-4, "margin")
    assert_close(ref_cpu.indices_to_embeddings(idx, E.detach()), fx["emb"], TIGHT, "indices_to_embeddings")


def test_codebook_config3_indices_bit_exact():
    """K 8192, C 32, N 2x1024 (BASELINE.json configs[2] codebook), tensors from seeds."""
    fx = load_golden("codebook_c3")
    K, C, B, T = (int(v) for v in fx["dims"])
    s_e, s_z = (int(v) for v in fx["seeds"])
    E = seeded((K, C), s_e)
    z = seeded((B, T, C), s_z)
    zq, idx, loss = ref_cpu.codebook_forward(z, E, 0.25)
    assert np.array_equal(idx.numpy().astype(np.int16), fx["idx"])
    assert_close(loss, fx["loss"], TIGHT, "loss")
    assert abs(float(zq.double().sum()) - float(fx["zq_sum"])) < 1e-3


def test_vitvqgan_small_matches_reference():
    import json
    import os

    from util import GOLDEN

    fx = load_golden("vitvqgan_small")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["vitvqgan_small"]
    cfg = meta["cfg"]
    w = weights_of(fx)
    imgs = torch.from_numpy(fx["imgs"])
    rec, loss, idx = ref_cpu.vitvqgan_forward(imgs, w, cfg)
    assert np.array_equal(idx.numpy(), fx["idx"])
    assert_close(rec, fx["rec"], 1e-5, "reconstruction")
    assert_close(loss, fx["loss"], 1e-5, "codebook loss")
    z = ref_cpu.vitvqgan_encode_features(imgs, w, cfg)
    assert_close(z, fx["z"], 1e-5, "pre-quant features")
    dec = ref_cpu.vitvqgan_decode_embeds(ref_cpu.indices_to_embeddings(idx, w["codebook.embedding.weight"]), w, cfg)
    assert_close(dec, fx["dec"], 1e-5, "decode_indices")


def test_moe_small_matches_reference():
    fx = load_golden("moe_small")
    D, E, k = (int(v) for v in fx["dims"])
    w = {n: v.clone().requires_grad_(True) for n, v in weights_of(fx).items()}
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    out, sel = ref_cpu.moe_layer(x, w, E, k)
    assert np.array_equal(sel.numpy(), fx["sel"])          # bit-exact expert ids
    assert_close(out, fx["out"], TIGHT, "out")
    names = sorted(w)
    gs = _grads(out, torch.from_numpy(fx["cot"]), [x] + [w[n] for n in names])
    assert_close(gs[0], fx["gx"], TIGHT, "grad x")
    for n, g in zip(names, gs[1:]):
        if "g:" + n in fx:
            assert_close(g, fx["g:" + n], TIGHT, f"grad {n}")
        else:
            assert g is None or float(g.abs().max()) == 0.0


@pytest.mark.parametrize("variant", ["self", "self_keymask"])
def test_switchhead_small_matches_reference(variant):
    fx = load_golden("switchhead_small")
    dim, h, d, E, k = (int(v) for v in fx["dims"])
    w = {n: v.clone().requires_grad_(True) for n, v in weights_of(fx).items()}
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    kw = {} if variant == "self" else dict(context_mask=torch.from_numpy(fx["keymask"]))
    out, sel_v, sel_o = ref_cpu.switchhead_attention(x, w, h, d, E, k, **kw)
    assert np.array_equal(sel_v.numpy(), fx["sel_v"])
    assert np.array_equal(sel_o.numpy(), fx["sel_o"])
    assert_close(out, fx[f"{variant}:out"], TIGHT, "out")
    names = sorted(w)
    gs = _grads(out, torch.from_numpy(fx["cot"]), [x] + [w[n] for n in names])
    assert_close(gs[0], fx[f"{variant}:gx"], TIGHT, "grad x")
    for n, g in zip(names, gs[1:]):
        key = f"{variant}:g:{n}"
        if key in fx:
            assert_close(g, fx[key], TIGHT, f"grad {n}")
        else:  # W_d.0.weight: the reference gives it no gradient
            assert n == "W_d.0.weight" and g is None


def test_agent_small_matches_reference():
    fx = load_golden("agent_small")
    dim, h, d, agent_num = (int(v) for v in fx["dims"])
    w = {n: v.clone().requires_grad_(True) for n, v in weights_of(fx).items()}
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    out = ref_cpu.agent_attention(x, w, h, d, agent_num)
    assert_close(out, fx["out"], TIGHT, "out")
    names = sorted(w)
    gs = _grads(out, torch.from_numpy(fx["cot"]), [x] + [w[n] for n in names])
    assert_close(gs[0], fx["gx"], TIGHT, "grad x")
    for n, g in zip(names, gs[1:]):
        assert_close(g, fx["g:" + n], TIGHT, f"grad {n}")


def _meta(name):
    import json
    import os

    from util import GOLDEN

    return json.load(open(os.path.join(GOLDEN, "golden_meta.json")))[name]


def test_vit_small_matches_reference():
    fx = load_golden("vit_small")
    cfg = _meta("vit_small")["cfg"]
    w = {n: v.clone().requires_grad_(v.dtype.is_floating_point) for n, v in weights_of(fx).items()}
    logits = ref_cpu.vit_forward(torch.from_numpy(fx["imgs"]), w, cfg["patch_size"], cfg["n_heads"], cfg["d_head"], cfg["depth"])
    assert tuple(logits.shape) == (2, cfg["num_classes"])
    assert_close(logits, fx["logits"], 1e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]))
    assert_close(loss, fx["loss"], 1e-5, "loss")
    names = [n for n in sorted(w) if "g:" + n in fx]
    gs = torch.autograd.grad(loss, [w[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        assert_close(g, fx["g:" + n], 2e-5, f"grad {n}")


def test_vit_moe_small_matches_reference():
    fx = load_golden("vit_moe_small")
    cfg = _meta("vit_moe_small")["cfg"]
    w = {n: v.clone().requires_grad_(True) for n, v in weights_of(fx).items()}
    logits, sels = ref_cpu.vit_moe_forward(torch.from_numpy(fx["imgs"]), w, cfg["patch_size"], cfg["n_heads"], cfg["d_head"],
                                           cfg["depth"], cfg["n_experts"], cfg["sel_experts"])
    assert_close(logits, fx["logits"], 1e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(fx["labels"]))
    assert_close(loss, fx["loss"], 1e-5, "loss")
    names = sorted(w)
    gs = torch.autograd.grad(loss, [w[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        if "g:" + n in fx:
            assert_close(g, fx["g:" + n], 5e-5, f"grad {n}")
        else:
            assert g is None or float(g.abs().max()) == 0.0, n


@pytest.mark.parametrize("variant", ["plain", "ctxmask"])
def test_muse_decoder_small_matches_reference(variant):
    """BidirectionalDecoder of models/muse.py (BASELINE.json configs[4]): logits, CE loss, gradients."""
    fx = load_golden("muse_decoder_small")
    cfg = _meta("muse_decoder_small")["cfg"]
    w = {n: v.clone().requires_grad_(v.dtype.is_floating_point) for n, v in weights_of(fx).items()}
    ctx = torch.from_numpy(fx["context"]).requires_grad_(True)
    kw = {} if variant == "plain" else dict(context_mask=torch.from_numpy(fx["cmask"]))
    logits = ref_cpu.bidirectional_decoder(torch.from_numpy(fx["ids"]), ctx, w, cfg["n_heads"], cfg["d_head"], cfg["depth"], **kw)
    assert_close(logits, fx[f"{variant}:logits"], 1e-5, "logits")
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), torch.from_numpy(fx["tgt"]), ignore_index=-1)
    assert_close(loss, fx[f"{variant}:loss"], 1e-5, "loss")
    names = [n for n in sorted(w) if f"{variant}:g:{n}" in fx]
    gs = torch.autograd.grad(loss, [ctx] + [w[n] for n in names], allow_unused=True)
    assert_close(gs[0], fx[f"{variant}:gctx"], 2e-5, "grad context")
    for n, g in zip(names, gs[1:]):
        assert_close(g, fx[f"{variant}:g:{n}"], 5e-5, f"grad {n}")


def test_vqgan_codebook_oracle_matches_reference_fixture():
    """oracle.vqgan_codebook_forward against the reference's conv-VQGAN Codebook (models/vqgan.py:138-182)."""
    fx = load_golden("vqgan_codebook")
    E = torch.from_numpy(fx["E"]).requires_grad_(True)
    z = torch.from_numpy(fx["z"]).requires_grad_(True)
    zq, idx, loss = ref_cpu.vqgan_codebook_forward(z, E, float(fx["beta"]))
    assert torch.equal(idx, torch.from_numpy(fx["idx"]))
    assert_close(zq, fx["zq"], 2e-6, "zq")
    assert_close(loss, fx["loss"], 2e-6, "loss")
    gz, gE = torch.autograd.grad((zq * torch.from_numpy(fx["cot"])).sum() + float(fx["loss_weight"]) * loss, [z, E])
    assert_close(gz, fx["gz"], 2e-6, "gz")
    assert_close(gE, fx["gE"], 2e-6, "gE")
    B = z.shape[0]
    emb = ref_cpu.vqgan_indices_to_embeddings(idx.view(B, -1), E.detach())
    assert_close(emb, fx["emb"], 0.0, "emb")


def test_muse_generate_loop_follows_the_reference():
    """The oracle's restatement of the parallel decode loop (ref_cpu.muse_generate over ref_cpu.sampling_step and
    ref_cpu.bidirectional_decoder) against the fixture produced by the reference's own MUSE.generate
    (oracle/gen_golden.py:gen_muse_generate): the ids entering the decoder at every step and the final ids."""
    import json
    import os

    from util import GOLDEN

    fx = load_golden("muse_generate_small")
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))["muse_generate_small"]
    cfg = meta["cfg"]
    w = weights_of(fx)
    seen, final = ref_cpu.muse_generate(torch.from_numpy(fx["context"]), w, cfg["n_heads"], cfg["d_head"], cfg["depth"],
                                        cfg["num_patches"], cfg["codebook_size"], meta["timesteps"], torch.from_numpy(fx["gumbel"]))
    for t, ids in enumerate(seen):
        assert torch.equal(ids, torch.from_numpy(fx["ids_in"][t])), f"decoder input at step {t}"
    assert torch.equal(final, torch.from_numpy(fx["final_ids"]))
