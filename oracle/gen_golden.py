"""Generate tests/golden/*.npz by running the REFERENCE's own source files on the CPU.

Runs only in the build container (the reference checkout is not present on the GPU box):

    python oracle/gen_golden.py            # writes tests/golden/*.npz + golden_meta.json

The reference modules are loaded from /root/reference/models/*.py by path through a namespace
stub for the ``models`` package (its __init__ imports torchvision/cv2, which are not installed;
SURVEY.md section 8c).  Nothing from the reference is copied: the fixtures hold inputs,
parameters and the outputs / gradients / indices the reference computes for them.

One documented patch: the reference's ViT-VQGAN FeedForward cannot be constructed at HEAD
(models/vitvqgan.py:20-34 calls object.__init__ with SwiGLU-style keywords), so the generator
substitutes an FFN with the SwiGLU semantics those keywords describe.
Attention and codebook fixtures do not depend on it.
"""
import importlib
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("AMK_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_reference():
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    mods = {}
    for name in ("softmax_attention", "agent_attention", "switchhead_attention", "moe", "vitvqgan", "positional_encoding",
                 "transformer", "vit", "vqgan"):
        mods[name] = importlib.import_module(f"models.{name}")
    pkg.SwitchHeadAttention = mods["switchhead_attention"].SwitchHeadAttention
    pkg.MoELayer = mods["moe"].MoELayer
    mods["vit_moe"] = importlib.import_module("models.vit_moe")
    return mods


from fixture_recipe import randomize_, seeded  # noqa: E402


def np_state(module):
    return {"w:" + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads_of(out, cot, wrt):
    gs = torch.autograd.grad((out * cot).sum(), wrt, allow_unused=True)
    return [None if g is None else g.detach().numpy().copy() for g in gs]


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    clean = {k: v for k, v in arrays.items() if v is not None}
    np.savez_compressed(path, **clean)
    return os.path.getsize(path)


# ----------------------------------------------------------------------------------------
def gen_softmax_attention(mods, meta):
    SA = mods["softmax_attention"].SoftmaxAttention
    dim, h, d, B, T, J = 96, 2, 64, 2, 40, 77
    m = SA(dim, h, d)
    randomize_(m, 11)
    x = seeded((B, T, dim), 101).requires_grad_(True)
    ctx = seeded((B, J, dim), 102).requires_grad_(True)
    cot = seeded((B, T, dim), 103)
    causal = torch.ones(T, T).triu(1).bool()
    keymask = torch.ones(B, T, dtype=torch.bool)
    keymask[0, -9:] = False
    keymask[1, :5] = False
    ctxmask = torch.ones(B, J, dtype=torch.bool)
    ctxmask[:, -17:] = False
    # a causal mask with one fully masked query row (uniform softmax over all keys)
    dead = causal.clone()
    dead[3, :] = True
    params = [p for _, p in sorted(m.named_parameters())]
    pnames = [n for n, _ in sorted(m.named_parameters())]
    arrays = dict(np_state(m), x=x.detach().numpy(), context=ctx.detach().numpy(), cot=cot.numpy(),
                  causal=causal.numpy(), keymask=keymask.numpy(), ctxmask=ctxmask.numpy(), dead=dead.numpy(),
                  dims=np.array([dim, h, d]))
    variants = {
        "self": dict(),
        "self_keymask": dict(context_mask=keymask),
        "self_causal": dict(causal_mask=causal),
        "self_both": dict(causal_mask=causal, context_mask=keymask),
        "self_deadrow": dict(causal_mask=dead),
        "cross": dict(context=ctx),
        "cross_ctxmask": dict(context=ctx, context_mask=ctxmask),
    }
    for vname, kw in variants.items():
        out = m(x, **kw)
        wrt = [x] + ([ctx] if "context" in kw else []) + params
        gs = grads_of(out, cot, wrt)
        arrays[f"{vname}:out"] = out.detach().numpy()
        arrays[f"{vname}:gx"] = gs[0]
        off = 1
        if "context" in kw:
            arrays[f"{vname}:gctx"] = gs[1]
            off = 2
        if vname in ("self_both", "cross_ctxmask"):  # parameter gradients for two variants (fixture size)
            for n, g in zip(pnames, gs[off:]):
                arrays[f"{vname}:g:{n}"] = g
    meta["softmax_attention"] = dict(bytes=save("softmax_attention", **arrays), variants=list(variants))

    # BASELINE.json configs[0]: dim 512, h 16, d 64, (B 2, T 128); weights/inputs from seeds,
    # outputs stored for every 4th token.
    dim, h, d, B, T = 512, 16, 64, 2, 128
    m = SA(dim, h, d)
    randomize_(m, 12)
    x = seeded((B, T, dim), 201).requires_grad_(True)
    cot = seeded((B, T, dim), 202)
    out = m(x)
    gx, = grads_of(out, cot, [x])
    t0 = time.perf_counter()
    for _ in range(5):
        o2 = m(x)
        torch.autograd.grad((o2 * cot).sum(), [x] + list(m.parameters()))
    ms = (time.perf_counter() - t0) / 5 * 1e3
    meta["softmax_attention_c1"] = dict(
        bytes=save("softmax_attention_c1", out_s4=out.detach().numpy()[:, ::4], gx_s4=gx[:, ::4],
                   dims=np.array([dim, h, d, B, T]), seeds=np.array([12, 201, 202])),
        ref_cpu_ms_fwd_bwd=ms)


def gen_codebook(mods, meta):
    CB = mods["vitvqgan"].Codebook
    # small, everything stored
    K, C, B, T = 512, 32, 2, 24
    cb = CB(K, C)
    with torch.no_grad():
        cb.embedding.weight.copy_(seeded((K, C), 301))
    z = seeded((B, T, C), 302).requires_grad_(True)
    cot = seeded((B, T, C), 303)
    zq, idx, loss = cb(z)
    gz, gE = torch.autograd.grad((zq * cot).sum() + 3.0 * loss, [z, cb.embedding.weight])
    with torch.no_grad():
        zf = F.normalize(z, dim=-1).view(-1, C)
        en = F.normalize(cb.embedding.weight, dim=-1)
        dist = zf.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zf @ en.t()
        two = torch.topk(dist, 2, dim=1, largest=False).values
        margin = (two[:, 1] - two[:, 0]).numpy()
        emb = cb.indices_to_embeddings(idx)
    meta["codebook_small"] = dict(bytes=save(
        "codebook_small", E=cb.embedding.weight.detach().numpy(), z=z.detach().numpy(), cot=cot.numpy(),
        zq=zq.detach().numpy(), idx=idx.numpy(), loss=loss.detach().numpy(), gz=gz.numpy(), gE=gE.numpy(),
        margin=margin, emb=emb.detach().numpy(), loss_weight=np.array(3.0)))

    # config C3 shape: K 8192, C 32, N = 2*1024; codebook and z from seeds, indices stored
    K, C, B, T = 8192, 32, 2, 1024
    cb = CB(K, C)
    with torch.no_grad():
        cb.embedding.weight.copy_(seeded((K, C), 311))
    z = seeded((B, T, C), 312)
    with torch.no_grad():
        t0 = time.perf_counter()
        zq, idx, loss = cb(z)
        ms = (time.perf_counter() - t0) * 1e3
        zf = F.normalize(z, dim=-1).view(-1, C)
        en = F.normalize(cb.embedding.weight, dim=-1)
        dist = zf.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zf @ en.t()
        two = torch.topk(dist, 2, dim=1, largest=False).values
        margin = (two[:, 1] - two[:, 0]).numpy()
    meta["codebook_c3"] = dict(bytes=save(
        "codebook_c3", idx=idx.numpy().astype(np.int16), loss=loss.numpy(), margin=margin.astype(np.float32),
        zq_sum=zq.double().sum().numpy(), dims=np.array([K, C, B, T]), seeds=np.array([311, 312])),
        ref_cpu_ms_fwd=ms, min_margin=float(margin.min()))


def gen_vqgan_codebook(mods, meta):
    """Conv-VQGAN codebook (models/vqgan.py:138-182): (B, C, H, W) layout, C = 256 (README.md:246-249),
    beta on the codebook term, indices_to_embeddings without the l2-norm."""
    CB = mods["vqgan"].Codebook
    K, C, B, H = 512, 256, 2, 4
    cb = CB(K, C)
    with torch.no_grad():
        cb.embedding.weight.copy_(seeded((K, C), 331))
    z = seeded((B, C, H, H), 332).requires_grad_(True)
    cot = seeded((B, C, H, H), 333)
    zq, idx, loss = cb(z)
    gz, gE = torch.autograd.grad((zq * cot).sum() + 3.0 * loss, [z, cb.embedding.weight])
    with torch.no_grad():
        zf = F.normalize(z.permute(0, 2, 3, 1), dim=-1).reshape(-1, C)
        en = F.normalize(cb.embedding.weight, dim=-1)
        dist = zf.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zf @ en.t()
        two = torch.topk(dist, 2, dim=1, largest=False).values
        margin = (two[:, 1] - two[:, 0]).numpy()
        emb = cb.indices_to_embeddings(idx.view(B, H * H))
    meta["vqgan_codebook"] = dict(bytes=save(
        "vqgan_codebook", E=cb.embedding.weight.detach().numpy(), z=z.detach().numpy(), cot=cot.numpy(),
        zq=zq.detach().numpy(), idx=idx.numpy(), loss=loss.detach().numpy(), gz=gz.numpy(), gE=gE.numpy(),
        margin=margin, emb=emb.detach().numpy(), loss_weight=np.array(3.0), beta=np.array(cb.beta)),
        min_margin=float(margin.min()))


class _SwiGLU(nn.Module):
    def __init__(self, in_features, hidden_features, out_features, bias=True):
        super().__init__()
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        a, b = self.w12(x).chunk(2, dim=-1)
        return self.w3(F.silu(a) * b)


class _PatchedFeedForward(_SwiGLU):
    """Stands in for the unconstructible reference FeedForward (see module docstring)."""

    def __init__(self, in_features, hidden_features=None, bias=True):
        hidden = (int(hidden_features * 2 / 3) + 7) // 8 * 8
        super().__init__(in_features=in_features, hidden_features=hidden, out_features=in_features, bias=bias)


def gen_vitvqgan(mods, meta):
    vv = mods["vitvqgan"]
    vv.FeedForward = _PatchedFeedForward
    cfg = dict(dim=64, img_size=32, patch_size=4, n_heads=2, d_head=64, depth=2, mlp_dim=96, dropout=0.0)
    cbp = dict(codebook_size=256, codebook_dim=32)
    m = vv.ViTVQGAN(cfg, cbp)
    randomize_(m, 21)
    with torch.no_grad():
        m.codebook.embedding.weight.copy_(seeded((256, 32), 401))
        m.encoder.pos_enc.copy_(seeded(m.encoder.pos_enc.shape, 402, 0.5))
        m.decoder.pos_enc.copy_(seeded(m.decoder.pos_enc.shape, 403, 0.5))
    g = torch.Generator().manual_seed(404)
    imgs = torch.rand(2, 3, 32, 32, generator=g)
    rec, loss = m(imgs)
    idx = m.encode_imgs(imgs)
    total = F.mse_loss(rec, imgs) + F.l1_loss(rec, imgs) + loss
    names = [n for n, _ in sorted(m.named_parameters())]
    gs = torch.autograd.grad(total, [p for _, p in sorted(m.named_parameters())], allow_unused=True)
    with torch.no_grad():
        dec = m.decode_indices(idx)
        z = m.pre_quant(m.encoder(imgs))
        zf = F.normalize(z, dim=-1).view(-1, 32)
        en = F.normalize(m.codebook.embedding.weight, dim=-1)
        dist = zf.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1) - 2 * zf @ en.t()
        two = torch.topk(dist, 2, dim=1, largest=False).values
        margin = (two[:, 1] - two[:, 0]).numpy()
    arrays = dict(np_state(m), imgs=imgs.numpy(), rec=rec.detach().numpy(), loss=loss.detach().numpy(),
                  idx=idx.numpy(), dec=dec.numpy(), z=z.numpy(), margin=margin, total=total.detach().numpy())
    for n, gr in zip(names, gs):
        if gr is not None:
            arrays["g:" + n] = gr.numpy()
    meta["vitvqgan_small"] = dict(bytes=save("vitvqgan_small", **arrays), cfg=cfg, codebook=cbp,
                                  n_params=sum(p.numel() for p in m.parameters()))


def gen_moe(mods, meta):
    Moe = mods["moe"].MoELayer
    D, E, k, B, T = 64, 6, 2, 2, 10
    m = Moe(D, D, E, k)
    randomize_(m, 31)
    x = seeded((B, T, D), 501).requires_grad_(True)
    cot = seeded((B, T, D), 502)
    out = m(x)
    with torch.no_grad():
        logits = m.gate(x)
        wts, sel = torch.topk(logits, k)
        srt = torch.sort(logits, dim=-1, descending=True).values
        gap = (srt[..., k - 1] - srt[..., k]).numpy()
    params = [p for _, p in sorted(m.named_parameters())]
    pnames = [n for n, _ in sorted(m.named_parameters())]
    gs = grads_of(out, cot, [x] + params)
    arrays = dict(np_state(m), x=x.detach().numpy(), cot=cot.numpy(), out=out.detach().numpy(), sel=sel.numpy(),
                  gap=gap, gx=gs[0], dims=np.array([D, E, k]))
    for n, g in zip(pnames, gs[1:]):
        arrays["g:" + n] = g
    meta["moe_small"] = dict(bytes=save("moe_small", **arrays))


def gen_switchhead(mods, meta):
    SH = mods["switchhead_attention"].SwitchHeadAttention
    dim, h, d, E, k, B, T = 96, 2, 64, 5, 2, 2, 10
    m = SH(dim, h, d, num_experts=E, sel_experts=k)
    randomize_(m, 41)
    x = seeded((B, T, dim), 601).requires_grad_(True)
    cot = seeded((B, T, dim), 602)
    keymask = torch.ones(B, T, dtype=torch.bool)
    keymask[0, -3:] = False
    params = [p for _, p in sorted(m.named_parameters())]
    pnames = [n for n, _ in sorted(m.named_parameters())]
    arrays = dict(np_state(m), x=x.detach().numpy(), cot=cot.numpy(), keymask=keymask.numpy(),
                  dims=np.array([dim, h, d, E, k]))
    with torch.no_grad():
        arrays["sel_v"] = torch.topk(m.W_s(x), k).indices.numpy()
        arrays["sel_o"] = torch.topk(m.W_d(x), k).indices.numpy()
    for vname, kw in {"self": {}, "self_keymask": dict(context_mask=keymask)}.items():
        out = m(x, **kw)
        gs = grads_of(out, cot, [x] + params)
        arrays[f"{vname}:out"] = out.detach().numpy()
        arrays[f"{vname}:gx"] = gs[0]
        for n, g in zip(pnames, gs[1:]):
            arrays[f"{vname}:g:{n}"] = g  # W_d.0.weight has no gradient (None -> not stored)
    meta["switchhead_small"] = dict(bytes=save("switchhead_small", **arrays))


def gen_agent(mods, meta):
    AA = mods["agent_attention"].AgentAttention
    dim, h, d, agent_num, B, T = 96, 3, 64, 9, 2, 20   # pool = int(9**0.5) = 3 == h
    m = AA(dim, h, d, agent_num=agent_num)
    randomize_(m, 51)
    with torch.no_grad():
        m.bias1.fill_(0.3)
        m.bias2.fill_(-0.2)
    x = seeded((B, T, dim), 701).requires_grad_(True)
    cot = seeded((B, T, dim), 702)
    out = m(x)
    params = [p for _, p in sorted(m.named_parameters())]
    pnames = [n for n, _ in sorted(m.named_parameters())]
    gs = grads_of(out, cot, [x] + params)
    arrays = dict(np_state(m), x=x.detach().numpy(), cot=cot.numpy(), out=out.detach().numpy(), gx=gs[0],
                  dims=np.array([dim, h, d, agent_num]))
    for n, g in zip(pnames, gs[1:]):
        arrays["g:" + n] = g
    meta["agent_small"] = dict(bytes=save("agent_small", **arrays))


def gen_vit(mods, meta):
    cfg = dict(dim=64, image_size=32, patch_size=8, n_heads=2, d_head=64, depth=2, mlp_dim=128, dropout=0.0, num_classes=10)
    m = mods["vit"].ViT(**cfg)
    randomize_(m, 61)
    g = torch.Generator().manual_seed(801)
    imgs = torch.randn(2, 3, 32, 32, generator=g)
    labels = torch.tensor([3, 7])
    logits = m(imgs)
    loss = F.cross_entropy(logits, labels)
    names = [n for n, _ in sorted(m.named_parameters())]
    gs = torch.autograd.grad(loss, [p for _, p in sorted(m.named_parameters())], allow_unused=True)
    arrays = dict(np_state(m), imgs=imgs.numpy(), labels=labels.numpy(), logits=logits.detach().numpy(),
                  loss=loss.detach().numpy())
    for n, gr in zip(names, gs):
        if gr is not None and gr.numel():
            arrays["g:" + n] = gr.numpy()
    meta["vit_small"] = dict(bytes=save("vit_small", **arrays), cfg=cfg, n_params=sum(p.numel() for p in m.parameters()))


def gen_vit_moe(mods, meta):
    cfg = dict(dim=64, image_size=32, patch_size=8, n_heads=2, d_head=64, depth=2, n_experts=4, sel_experts=2,
               dropout=0.0, num_classes=10)
    m = mods["vit_moe"].ViTMoE(**cfg)
    randomize_(m, 71)
    g = torch.Generator().manual_seed(901)
    imgs = torch.randn(2, 3, 32, 32, generator=g)
    labels = torch.tensor([1, 8])
    logits = m(imgs)
    loss = F.cross_entropy(logits, labels)
    names = [n for n, _ in sorted(m.named_parameters())]
    gs = torch.autograd.grad(loss, [p for _, p in sorted(m.named_parameters())], allow_unused=True)
    arrays = dict(np_state(m), imgs=imgs.numpy(), labels=labels.numpy(), logits=logits.detach().numpy(),
                  loss=loss.detach().numpy())
    for n, gr in zip(names, gs):
        if gr is not None:
            arrays["g:" + n] = gr.numpy()
    meta["vit_moe_small"] = dict(bytes=save("vit_moe_small", **arrays), cfg=cfg,
                                 n_params=sum(p.numel() for p in m.parameters()))


def gen_muse_decoder(mods, meta):
    muse = importlib.import_module("models.muse")  # imports transformers' CLIP classes, fetches nothing
    cfg = dict(dim=64, codebook_size=256, n_heads=2, d_head=64, depth=2, mult=4, dropout=0.0, num_patches=16)
    m = muse.BidirectionalDecoder(**cfg)
    randomize_(m, 81)
    g = torch.Generator().manual_seed(1001)
    ids = torch.randint(0, cfg["codebook_size"] + 1, (2, 16), generator=g)
    ctx = seeded((2, 7, 64), 1002).requires_grad_(True)
    cmask = torch.ones(2, 7, dtype=torch.bool)
    cmask[1, -2:] = False
    tgt = torch.randint(0, cfg["codebook_size"], (2, 16), generator=g)
    tgt[0, ::3] = -1
    arrays = dict(np_state(m), ids=ids.numpy(), context=ctx.detach().numpy(), cmask=cmask.numpy(), tgt=tgt.numpy())
    names = [n for n, _ in sorted(m.named_parameters())]
    for vname, kw in {"plain": {}, "ctxmask": dict(context_mask=cmask)}.items():
        logits = m(ids, context=ctx, **kw)
        loss = F.cross_entropy(logits.transpose(1, 2), tgt, ignore_index=-1)
        gs = torch.autograd.grad(loss, [ctx] + [p for _, p in sorted(m.named_parameters())], allow_unused=True)
        arrays[f"{vname}:logits"] = logits.detach().numpy()
        arrays[f"{vname}:loss"] = loss.detach().numpy()
        arrays[f"{vname}:gctx"] = gs[0].numpy()
        if vname == "plain":  # parameter gradients once (they dominate the fixture size)
            for n, gr in zip(names, gs[1:]):
                if gr is not None:
                    arrays[f"{vname}:g:{n}"] = gr.numpy()
    # helper functions of the sampling loop (deterministic parts)
    lg = seeded((2, 5, 40), 1003)
    arrays["filter_in"] = lg.numpy()
    arrays["filter_out"] = muse.filter_logits(lg, p=0.9).numpy()
    tt = torch.linspace(0, 1, 18)
    arrays["cosine_t"] = tt.numpy()
    arrays["cosine_out"] = muse.cosine_schedule(tt).numpy()
    meta["muse_decoder_small"] = dict(bytes=save("muse_decoder_small", **arrays), cfg=cfg,
                                      n_params=sum(p.numel() for p in m.parameters()))


def gen_softmax_attention_bf16(mods, meta):
    """The reference's SoftmaxAttention under torch.autocast(bfloat16) -- its shipped training precision
    (cfg/vitvqgan.yaml:73; trainers/vitgqgan.py:149,170 run the forwards inside accelerator.autocast()): Linear and
    einsum in bf16, softmax in f32 (autocast's fp32 list), f32 parameters and gradients.  Stored next to the same
    module's f32 results, so that the fixture itself says what tolerance the mode supports: `err_out` / `err_gx` =
    max |autocast - f32| / max |f32| of the reference against ITSELF."""
    SA = mods["softmax_attention"].SoftmaxAttention
    dim, h, d, B, T, J = 128, 2, 64, 2, 96, 77
    m = SA(dim, h, d)
    randomize_(m, 21)
    x = seeded((B, T, dim), 301)
    ctx = seeded((B, J, dim), 302)
    cot = seeded((B, T, dim), 303)
    keymask = torch.ones(B, T, dtype=torch.bool)
    keymask[0, -9:] = False
    ctxmask = torch.ones(B, J, dtype=torch.bool)
    ctxmask[:, -17:] = False
    params = [p for _, p in sorted(m.named_parameters())]
    pnames = [n for n, _ in sorted(m.named_parameters())]
    arrays = dict(np_state(m), x=x.numpy(), context=ctx.numpy(), cot=cot.numpy(), keymask=keymask.numpy(),
                  ctxmask=ctxmask.numpy(), dims=np.array([dim, h, d]))
    variants = {"self": dict(), "self_keymask": dict(context_mask=keymask), "cross_ctxmask": dict(context=ctx, context_mask=ctxmask)}
    errs = {}
    for vname, kw in variants.items():
        res = {}
        for mode in ("f32", "bf16"):
            xr = x.clone().requires_grad_(True)
            kw2 = dict(kw)
            wrt = [xr]
            if "context" in kw:
                kw2["context"] = ctx.clone().requires_grad_(True)
                wrt.append(kw2["context"])
            if mode == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = m(xr, **kw2)
            else:
                out = m(xr, **kw2)
            gs = torch.autograd.grad((out.float() * cot).sum(), wrt + params)
            res[mode] = (out.detach().float(), [g.detach().float() for g in gs])
        (o32, g32), (o16, g16) = res["f32"], res["bf16"]
        arrays[f"{vname}:out"] = o16.numpy()
        arrays[f"{vname}:out_f32"] = o32.numpy()
        arrays[f"{vname}:gx"] = g16[0].numpy()
        arrays[f"{vname}:gx_f32"] = g32[0].numpy()
        off = 2 if "context" in kw else 1
        if "context" in kw:
            arrays[f"{vname}:gctx"] = g16[1].numpy()
            arrays[f"{vname}:gctx_f32"] = g32[1].numpy()
        for n, a, b in zip(pnames, g16[off:], g32[off:]):
            arrays[f"{vname}:g:{n}"] = a.numpy()
            arrays[f"{vname}:g32:{n}"] = b.numpy()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        errs[vname] = dict(err_out=rel(o16, o32), err_gx=rel(g16[0], g32[0]),
                           err_gparams=max(rel(a, b) for a, b in zip(g16[off:], g32[off:])))
    meta["softmax_attention_bf16"] = dict(bytes=save("softmax_attention_bf16", **arrays), variants=list(variants),
                                          reference_autocast_vs_reference_f32=errs)


def gen_muse_generate(mods, meta):
    """The parallel decode loop itself: the reference's MUSE.generate (models/muse.py:180-239) called as an unbound
    function on a namespace holding the reference's BidirectionalDecoder, a synthetic-context text encoder (SURVEY.md
    section 8c: CLIP needs a download) and an identity `vq` -- so the loop that runs is the reference's own text:
    cosine schedule, argsort / scatter masking, the two decoder passes, guidance, softmax, filter_logits,
    F.gumbel_softmax, gather.  Stored: the ids entering the decoder at every step (recorded by the decoder wrapper),
    the final ids, and the Gumbel noise the loop drew (re-drawn from the same seed: F.gumbel_softmax is the loop's
    only consumer of the global generator)."""
    muse = importlib.import_module("models.muse")
    cfg = dict(dim=64, codebook_size=256, n_heads=2, d_head=64, depth=2, mult=4, dropout=0.0, num_patches=16)
    dec = muse.BidirectionalDecoder(**cfg)
    randomize_(dec, 91)
    dec.eval()
    b, timesteps, seed = 2, 6, 4242
    ctx = seeded((b, 7, 64), 1102)
    seen = []

    def decoder(ids, context=None, context_mask=None):
        seen.append(ids.clone())
        return dec(ids, context=context, context_mask=context_mask)

    ns = types.SimpleNamespace(vq=types.SimpleNamespace(num_patches=cfg["num_patches"], decode_indices=lambda ids: ids),
                               text_encoder=lambda texts, device=None: (ctx, ctx), mask_token_id=cfg["codebook_size"],
                               decoder=decoder)
    torch.manual_seed(seed)
    with torch.no_grad():
        final_ids = muse.MUSE.generate(ns, ["a", "b"], timesteps=timesteps, device="cpu")
    assert len(seen) == 2 * timesteps and all(torch.equal(seen[2 * t], seen[2 * t + 1]) for t in range(timesteps))
    torch.manual_seed(seed)
    noise = torch.stack([-torch.empty(b, cfg["num_patches"], cfg["codebook_size"]).exponential_().log() for _ in range(timesteps)])
    arrays = dict(np_state(dec), context=ctx.numpy(), final_ids=final_ids.numpy(), gumbel=noise.numpy(),
                  ids_in=torch.stack([seen[2 * t] for t in range(timesteps)]).numpy())
    meta["muse_generate_small"] = dict(bytes=save("muse_generate_small", **arrays), cfg=cfg, timesteps=timesteps, seed=seed)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    mods = load_reference()
    meta_path = os.path.join(OUT, "golden_meta.json")
    if len(sys.argv) > 2 and sys.argv[1] == "--only":   # add / refresh single fixtures, keep the rest
        meta = json.load(open(meta_path))
        for name in sys.argv[2:]:
            globals()["gen_" + name](mods, meta)
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        return
    meta = dict(torch=torch.__version__, threads=torch.get_num_threads(), cpu_count=os.cpu_count(),
                reference=REF, generated_by="oracle/gen_golden.py")
    gen_vqgan_codebook(mods, meta)
    gen_softmax_attention(mods, meta)
    gen_codebook(mods, meta)
    gen_vitvqgan(mods, meta)
    gen_moe(mods, meta)
    gen_switchhead(mods, meta)
    gen_agent(mods, meta)
    gen_vit(mods, meta)
    gen_vit_moe(mods, meta)
    gen_muse_decoder(mods, meta)
    gen_muse_generate(mods, meta)
    gen_softmax_attention_bf16(mods, meta)
    with open(os.path.join(OUT, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    total = sum(v["bytes"] for v in meta.values() if isinstance(v, dict) and "bytes" in v)
    print(json.dumps(meta, indent=1, sort_keys=True))
    print(f"total fixture bytes: {total}")


if __name__ == "__main__":
    main()
