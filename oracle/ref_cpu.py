"""CPU oracle: a fresh restatement of the reference's attention / MoE / VQ arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under ``attention-models_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may call
it, and only as the checker / the timed baseline -- never as a product path.

Each function states the reference lines it follows (paths relative to /root/reference).  The
op sequence is the reference's (same associations, same fill constants, same reduction axes)
but the code is written from scratch as pure functions over explicit weights, in plain
PyTorch fp32 on the CPU.  It is pinned by ``tests/golden/*.npz``, generated in the build
container from the reference's own source files by ``oracle/gen_golden.py``
(``tests/test_oracle_golden.py`` is the check).  The reference ships no tests or golden
vectors of its own, so those fixtures are the pin.
"""
import torch
import torch.nn.functional as F

FILL = -1e9  # masked_fill value used by the reference (softmax_attention.py:67,71)


# ------------------------------------------------------------------ attention core
def attention_core(q, k, v, scale, key_mask=None, causal_mask=None):
    """models/softmax_attention.py:62-76 on (B,h,T,d) tensors.

    key_mask: bool (B,J), True = keep; causal_mask: bool (I,J), True = masked.
    """
    s = torch.matmul(q * scale, k.transpose(-1, -2))
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], FILL)
    if causal_mask is not None:
        s = s.masked_fill(causal_mask, FILL)
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v)


def _split_heads(t, h, d):
    B, T, _ = t.shape
    return t.view(B, T, h, d).permute(0, 2, 1, 3)


def softmax_attention(x, w, num_heads, dim_head, context=None, causal_mask=None, context_mask=None):
    """SoftmaxAttention.forward (models/softmax_attention.py:48-82), dropout 0.

    w: {'q.0.weight', 'kv.0.weight', 'W_o.weight', 'W_o.bias'}.
    """
    src = x if context is None else context
    B, I, _ = x.shape
    J = src.shape[1]
    q = _split_heads(x @ w["q.0.weight"].t(), num_heads, dim_head)
    kv = (src @ w["kv.0.weight"].t()).view(B, J, 2, num_heads, dim_head)
    k = kv[:, :, 0].permute(0, 2, 1, 3)
    v = kv[:, :, 1].permute(0, 2, 1, 3)
    o = attention_core(q, k, v, dim_head ** -0.5, context_mask, causal_mask)
    o = o.permute(0, 2, 1, 3).reshape(B, I, num_heads * dim_head)
    return o @ w["W_o.weight"].t() + w["W_o.bias"]


# ------------------------------------------------------------------ agent attention
def agent_attention(x, w, num_heads, dim_head, agent_num=47):
    """AgentAttention.forward (models/agent_attention.py:49-80), dropout 0.

    w: {'qkv.weight', 'W_o.weight', 'W_o.bias', 'bias1', 'bias2', 'dwc.1.weight', 'dwc.1.bias'}.
    Requires num_heads == int(agent_num ** 0.5).
    """
    B, T, _ = x.shape
    h, d = num_heads, dim_head
    pool = int(agent_num ** 0.5)
    scale = d ** -0.5
    qkv = (x @ w["qkv.weight"].t()).view(B, T, 3, h, d)
    q = qkv[:, :, 0].permute(0, 2, 1, 3)
    k = qkv[:, :, 1].permute(0, 2, 1, 3)
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    # pool the (t, h) plane of q per channel to (pool, pool); result indexed (b, h', t', d)
    agents = F.adaptive_avg_pool2d(q.permute(0, 3, 2, 1), (pool, pool)).permute(0, 3, 2, 1)
    s1 = torch.matmul(agents * scale, k.transpose(-1, -2)) + w["bias1"]
    v_agent = torch.matmul(torch.softmax(s1, dim=-1), v)
    s2 = torch.matmul(q * scale, agents.transpose(-1, -2)) + w["bias2"]
    o = torch.matmul(torch.softmax(s2, dim=-1), v_agent)
    conv = F.conv2d(v.permute(0, 3, 1, 2), w["dwc.1.weight"], w["dwc.1.bias"], padding=1, groups=d)
    o = o + conv.permute(0, 2, 3, 1)
    o = o.permute(0, 2, 1, 3).reshape(B, T, h * d)
    return o @ w["W_o.weight"].t() + w["W_o.bias"]


# ------------------------------------------------------------------ top-k routing / MoE
def topk_route(logits, k):
    """torch.topk(logits, k) + sigmoid of the selected logits (models/moe.py:27-29)."""
    vals, ids = torch.topk(logits, k)
    return torch.sigmoid(vals), ids


def moe_layer(x, w, num_experts, sel_experts):
    """MoELayer.forward (models/moe.py:23-38).

    w: {'gate.weight', 'gate.bias', 'experts.{e}.weight', 'experts.{e}.bias'}.
    Returns (out, selected_experts int64 (B,T,k)).  Experts accumulate in ascending id.
    """
    B, T, D = x.shape
    logits = x @ w["gate.weight"].t() + w["gate.bias"]
    gatew, sel = topk_route(logits, sel_experts)
    out = torch.zeros(B, T, D)
    for e in range(num_experts):
        hit = sel == e                                   # (B,T,k)
        if not hit.any():
            continue
        rows = hit.any(-1)                               # a token selects an expert at most once
        coef = (gatew * hit).sum(-1)[rows]               # its weight for this expert
        y = x[rows] @ w[f"experts.{e}.weight"].t() + w[f"experts.{e}.bias"]
        out[rows] += coef[:, None] * y
    return out, sel


# ------------------------------------------------------------------ SwitchHead attention
def switchhead_attention(x, w, num_heads, dim_head, num_experts, sel_experts,
                         causal_mask=None, context_mask=None):
    """SwitchHeadAttention.forward, self-attention (models/switchhead_attention.py:58-116).

    w: {'q.0.weight','k.0.weight','W_s.0.weight','W_d.0.weight','experts_v.{e}.weight',
        'experts_out.{e}.weight'}.  Returns (out, sel_v (B,T,h,k), sel_out (B,T,h,k)).
    moe_out ignores its gate weights (only the indices route), as the reference does.
    """
    B, T, D = x.shape
    h, d, E, k = num_heads, dim_head, num_experts, sel_experts
    q = _split_heads(x @ w["q.0.weight"].t(), h, d)
    kk = _split_heads(x @ w["k.0.weight"].t(), h, d)

    gate_v, sel_v = topk_route((x @ w["W_s.0.weight"].t()).view(B, T, h, E), k)
    v = torch.zeros(B, T, h, d)
    for e in range(E):
        hit = sel_v == e                                 # (B,T,h,k)
        if not hit.any():
            continue
        coef = (gate_v * hit).sum(-1)                    # (B,T,h), 0 where not routed
        y = x @ w[f"experts_v.{e}.weight"].t()           # (B,T,d)
        v = v + coef[..., None] * y[:, :, None, :] * hit.any(-1)[..., None]
    v = v.permute(0, 2, 1, 3)

    o = attention_core(q, kk, v, d ** -0.5, context_mask, causal_mask)  # (B,h,T,d)
    o = o.permute(0, 2, 1, 3)                                            # (B,T,h,d)

    sel_o = torch.topk((x @ w["W_d.0.weight"].t()).view(B, T, h, E), k).indices
    out = torch.zeros(B, T, h, D)
    for e in range(E):
        hit = (sel_o == e).any(-1)                       # (B,T,h)
        if not hit.any():
            continue
        y = o @ w[f"experts_out.{e}.weight"].t()         # (B,T,h,D)
        out = out + y * hit[..., None]
    return out.sum(dim=-2), sel_v, sel_o


# ------------------------------------------------------------------ VQ codebook
def l2norm(t):
    """models/vitvqgan.py:16-17: F.normalize(x, p=2, dim=-1)."""
    return t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def codebook_distances(z, codebook):
    """(sum z^2 + sum e^2) - 2 z e^T on normalised rows (models/vitvqgan.py:152-159)."""
    zn = l2norm(z).reshape(-1, z.shape[-1])
    en = l2norm(codebook)
    return (zn.pow(2).sum(1, keepdim=True) + en.pow(2).sum(1)) - 2.0 * (zn @ en.t())


def codebook_forward(z, codebook, beta=0.25):
    """Codebook.forward (models/vitvqgan.py:151-171): (z_q straight-through, idx, loss)."""
    zn = l2norm(z)
    d = codebook_distances(z, codebook)
    idx = torch.argmin(d, dim=1).view(z.shape[:-1])
    zq = l2norm(codebook[idx])
    loss = beta * torch.mean((zq.detach() - zn) ** 2) + torch.mean((zq - zn.detach()) ** 2)
    out = zn + (zq - zn).detach()
    return out, idx, loss


def codebook_margin(z, codebook):
    """Gap between the smallest and second-smallest distance per row (tie diagnostics)."""
    d = codebook_distances(z, codebook)
    two = torch.topk(d, 2, dim=1, largest=False).values
    return two[:, 1] - two[:, 0]


def indices_to_embeddings(idx, codebook):
    """models/vitvqgan.py:173-176."""
    return l2norm(codebook[idx])


def vqgan_codebook_forward(z_bchw, codebook, beta=0.25):
    """Conv-VQGAN Codebook.forward (models/vqgan.py:148-176): channels-first input, the same normalised
    nearest-neighbour lookup, but beta weighs the CODEBOOK term and the indices come back flat (B*H*W)."""
    z = l2norm(z_bchw.permute(0, 2, 3, 1))
    d = codebook_distances(z, codebook)
    idx = torch.argmin(d, dim=1)
    zq = l2norm(codebook[idx]).view(z.shape)
    loss = torch.mean((zq.detach() - z) ** 2) + beta * torch.mean((zq - z.detach()) ** 2)
    out = z + (zq - z).detach()
    return out.permute(0, 3, 1, 2), idx, loss


def vqgan_indices_to_embeddings(idx_bt, codebook):
    """models/vqgan.py:178-182: raw (un-normalised) rows, 'b (h w) d -> b d h w' with h = w = sqrt(T)."""
    e = codebook[idx_bt]
    side = int(e.shape[1] ** 0.5)
    return e.view(e.shape[0], side, side, -1).permute(0, 3, 1, 2)


# ------------------------------------------------------------------ ViT-VQGAN scaffolding
def _ln(x, w, prefix):
    return F.layer_norm(x, x.shape[-1:], w[prefix + ".weight"], w[prefix + ".bias"])


def _sub(w, prefix):
    n = len(prefix) + 1
    return {k[n:]: v for k, v in w.items() if k.startswith(prefix + ".")}


def _patchify(img, p):
    B, C, H, W = img.shape
    t = img.view(B, C, H // p, p, W // p, p)
    return t.permute(0, 2, 4, 3, 5, 1).reshape(B, (H // p) * (W // p), p * p * C)


def _unpatchify(tok, p, grid):
    B = tok.shape[0]
    return tok.view(B, grid, grid, p, p, -1).permute(0, 5, 1, 3, 2, 4).reshape(B, -1, grid * p, grid * p)


def _swiglu(x, w):
    a, b = (x @ w["w12.weight"].t() + w["w12.bias"]).chunk(2, dim=-1)
    return (F.silu(a) * b) @ w["w3.weight"].t() + w["w3.bias"]


def _vit_blocks(x, w, prefix, depth, h, d):
    for i in range(depth):
        lw = _sub(w, f"{prefix}.layers.{i}")
        x = x + softmax_attention(_ln(x, lw, "norm1"), _sub(lw, "self_attn"), h, d)
        x = x + _swiglu(_ln(x, lw, "norm2"), _sub(lw, "feed_forward"))
    return x


def vitvqgan_encode_features(imgs, w, cfg):
    """ViTEncoder.forward + pre_quant (models/vitvqgan.py:100-108, 192)."""
    p, h, d, depth = cfg["patch_size"], cfg["n_heads"], cfg["d_head"], cfg["depth"]
    x = _patchify(imgs, p)
    x = _ln(x, w, "encoder.to_patch_embedding.1")
    x = x @ w["encoder.to_patch_embedding.2.weight"].t() + w["encoder.to_patch_embedding.2.bias"]
    x = _ln(x, w, "encoder.to_patch_embedding.3")
    x = _ln(w["encoder.pos_enc"] + x, w, "encoder.pre_norm")
    x = _vit_blocks(x, w, "encoder.encoder", depth, h, d)
    return x @ w["pre_quant.weight"].t() + w["pre_quant.bias"]


def vitvqgan_decode_embeds(zq, w, cfg):
    """post_quant + ViTDecoder.forward (models/vitvqgan.py:127-137, 194-195)."""
    p, h, d, depth = cfg["patch_size"], cfg["n_heads"], cfg["d_head"], cfg["depth"]
    x = zq @ w["post_quant.weight"].t() + w["post_quant.bias"]
    x = _ln(x + w["decoder.pos_enc"], w, "decoder.pre_norm")
    x = _vit_blocks(x, w, "decoder.decoder", depth, h, d)
    x = x @ w["decoder.fc.weight"].t() + w["decoder.fc.bias"]
    return _unpatchify(x, p, cfg["img_size"] // p)


def vitvqgan_forward(imgs, w, cfg, beta=0.25):
    """ViTVQGAN.forward (models/vitvqgan.py:190-196) with the SwiGLU FFN decision of
    SURVEY.md section 8c.  Returns (reconstruction, loss, indices)."""
    z = vitvqgan_encode_features(imgs, w, cfg)
    zq, idx, loss = codebook_forward(z, w["codebook.embedding.weight"], beta)
    return vitvqgan_decode_embeds(zq, w, cfg), loss, idx


# ------------------------------------------------------------------ ViT / ViTMoE scaffolding
def _geglu_ffn(x, w):
    """transformer.FeedForward (models/transformer.py:22-43); inner width may be 0 (ViT quirk)."""
    hcat = x @ w["ff.0.weight"].t()
    val, gate = hcat.chunk(2, dim=-1)
    hid = gate * F.gelu(val)
    hid = F.layer_norm(hid, hid.shape[-1:], w["ff.2.gamma"], w["ff.2.beta"])
    return hid @ w["ff.3.weight"].t()


def _patch_embed(imgs, w, patch):
    x = _patchify(imgs, patch)
    x = _ln(x, w, "to_patch_embedding.1")
    x = x @ w["to_patch_embedding.2.weight"].t() + w["to_patch_embedding.2.bias"]
    return _ln(x, w, "to_patch_embedding.3")


def vit_forward(imgs, w, patch, n_heads, d_head, depth):
    """ViT.forward (models/vit.py:52-69) incl. the mult=dropout quirk: the per-layer FFN has the
    width the checkpoint says (0 at dropout 0.0) and encoder.feed_forward is never applied."""
    x = _patch_embed(imgs, w, patch)
    cls = w["class_token"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + w["pos_enc"]
    for i in range(depth):
        lw = _sub(w, f"encoder.layers.{i}")
        xn = F.layer_norm(x, x.shape[-1:], lw["norm1.gamma"], lw["norm1.beta"])
        x = softmax_attention(xn, _sub(lw, "self_attn"), n_heads, d_head) + x
        xn = F.layer_norm(x, x.shape[-1:], lw["norm2.gamma"], lw["norm2.beta"])
        x = _geglu_ffn(xn, _sub(lw, "feed_forward")) + x
    return x[:, 0] @ w["final_fc.weight"].t() + w["final_fc.bias"]


def vit_moe_forward(imgs, w, patch, n_heads, d_head, depth, n_experts, sel_experts):
    """ViTMoE.forward (models/vit_moe.py:89-107).  Returns (logits, [per-layer (sel_v, sel_o, sel_moe)])."""
    x = _patch_embed(imgs, w, patch)
    cls = w["class_token"].expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1) + w["pos_enc"]
    sels = []
    for i in range(depth):
        lw = _sub(w, f"encoder.layers.{i}")
        a, sv, so = switchhead_attention(_ln(x, lw, "norm1"), _sub(lw, "self_attn"), n_heads, d_head, n_experts, sel_experts)
        x = a + x
        m, sm = moe_layer(_ln(x, lw, "norm2"), _sub(lw, "moe"), n_experts, sel_experts)
        x = m + x
        sels.append((sv, so, sm))
    x = _ln(x, w, "norm")
    return x[:, 0] @ w["class_embed.weight"].t() + w["class_embed.bias"], sels


# ------------------------------------------------------------------ masked-token decoder (Muse)
def _gamma_ln(x, w, prefix):
    return F.layer_norm(x, x.shape[-1:], w[prefix + ".gamma"], w[prefix + ".beta"])


def bidirectional_decoder(ids, context, w, n_heads, d_head, depth, context_mask=None):
    """BidirectionalDecoder.forward (models/muse.py:88-96) over transformer.Decoder
    (models/transformer.py:87-135): per layer self-attention (no causal mask), cross-attention
    onto `context`, GEGLU feed-forward, each pre-normed with a residual."""
    x = w["token_emb.weight"][ids] + w["pos_enc"]
    for i in range(depth):
        lw = _sub(w, f"decoder.layers.{i}")
        x = softmax_attention(_gamma_ln(x, lw, "norm1"), _sub(lw, "self_attn"), n_heads, d_head) + x
        x = softmax_attention(_gamma_ln(x, lw, "norm2"), _sub(lw, "cross_attn"), n_heads, d_head,
                              context=context, context_mask=context_mask) + x
        x = _geglu_ffn(_gamma_ln(x, lw, "norm3"), _sub(lw, "feed_forward")) + x
    x = _gamma_ln(x, w, "final_norm")
    return x @ w["linear.weight"].t()


# ------------------------------------------------------------------ masked-token decoder (MaskGit)
def maskgit_transformer(ids, w, n_heads, d_head, depth):
    """BiDirectionalTransformer.forward (models/maskgit.py:80-91) over transformer.Encoder
    (models/transformer.py:46-76): embedding + positions -> LN -> depth x (self-attention, GEGLU FFN; pre-LN,
    residual) -> LN -> logits.  The reference file itself cannot be imported here (it imports cv2), so this
    composition is restated from its text; its parts are the ones the other fixtures pin."""
    x = w["input_proj.weight"][ids] + w["pos_enc"]
    x = _gamma_ln(x, w, "init_norm")
    for i in range(depth):
        lw = _sub(w, f"decoder.layers.{i}")
        x = softmax_attention(_gamma_ln(x, lw, "norm1"), _sub(lw, "self_attn"), n_heads, d_head) + x
        x = _geglu_ffn(_gamma_ln(x, lw, "norm2"), _sub(lw, "feed_forward")) + x
    x = _gamma_ln(x, w, "final_norm")
    return x @ w["linear.weight"].t()


def sampling_step(logits, ids, mask, gumbel, tau, null_logits=None, cfg_scale=3.0, p=0.9, unmasked_score=None):
    """One step of the parallel decode as the reference writes it (models/muse.py:211-236,
    models/maskgit.py:255-272), with the Gumbel noise given: CFG combine, softmax, filter_logits (top
    ceil((1-p) V) logits, -inf elsewhere), argmax of softmax((filtered + g) / tau), probability gather,
    masked id update.  Returns (new ids, scores)."""
    import math

    s = logits if null_logits is None else null_logits + cfg_scale * (logits - null_logits)
    probs = torch.softmax(s, dim=-1)
    k = math.ceil((1 - p) * s.shape[-1])
    val, ind = s.topk(k, dim=-1)
    filtered = torch.full_like(s, float("-inf")).scatter_(2, ind, val)
    y = torch.softmax((filtered + gumbel) / tau, dim=-1)   # tau = 0: all NaN, argmax 0 (the reference's last step)
    pred = y.argmax(dim=-1)
    ids = ids.clone()
    ids[mask] = pred[mask]
    scores = probs.gather(2, pred.unsqueeze(-1)).squeeze(-1)
    if unmasked_score is not None:
        scores = scores.masked_fill(~mask, unmasked_score)
    return ids, scores


def muse_generate(ctx, w, n_heads, d_head, depth, num_patches, mask_token_id, timesteps, gumbel):
    """MUSE.generate (models/muse.py:180-239) restated on the functions above, with the loop's Gumbel noise given
    (timesteps, B, n, V).  Returns (ids entering the decoder at each step, final ids)."""
    import math

    b, n = ctx.shape[0], num_patches
    ids = torch.full((b, n), mask_token_id, dtype=torch.long)
    scores = torch.zeros(b, n)
    mask = torch.zeros(b, n, dtype=torch.bool)
    seen = []
    for step, t in enumerate(torch.linspace(0, 1, timesteps)):
        steps_until_x0 = timesteps - 1 - step
        n_masked = max(int((torch.cos(t * math.pi / 2) * n).item()), 1)
        mask.scatter_(1, torch.argsort(scores, dim=-1)[:, :n_masked], True)
        ids = ids.masked_fill(mask, mask_token_id)
        seen.append(ids.clone())
        logits = bidirectional_decoder(ids, ctx, w, n_heads, d_head, depth)
        null_logits = bidirectional_decoder(ids, torch.zeros_like(ctx), w, n_heads, d_head, depth)
        ids, scores = sampling_step(logits, ids, mask, gumbel[step], 1 * (steps_until_x0 / timesteps), null_logits=null_logits)
        mask = torch.zeros_like(mask)
    return seen, ids
