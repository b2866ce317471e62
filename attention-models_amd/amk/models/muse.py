"""Masked-token image generator over frozen ViT-VQGAN codes (BASELINE.json configs[4]; reference:
models/muse.py:57-239).  The transformer stack runs on the HIP attention kernels: bidirectional
self-attention over the 1024 image tokens and cross-attention onto the 77 text positions
(I = 1024, J = 77).

What differs from the reference, and why: its TextEncoder downloads a CLIP text tower
(``CLIPTextModel.from_pretrained``, models/muse.py:41-42) -- a network fetch that is unavailable
here -- so ``MUSE`` takes the tower's OUTPUT (``text_hidden`` (B, 77, 768), CLIP's last hidden
state) instead of strings and keeps only the trainable part, ``project_embeds``.  Everything
downstream follows the reference: cosine mask schedule, classifier-free-guidance drop, the
18-step confidence-based parallel decode with CFG scale 3, top-(1-p) logit filter and Gumbel
sampling, including its quirks (temperature reaches 0 on the last step).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .transformer import Decoder, LayerNorm


def cosine_schedule(t):
    return torch.cos(t * math.pi / 2)


def filter_logits(logits, p=0.9):
    """Keep the ceil((1-p) * n_classes) largest logits per position, -inf elsewhere."""
    k = math.ceil((1 - p) * logits.shape[-1])
    val, ind = logits.topk(k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(2, ind, val)
    return out


class BidirectionalDecoder(nn.Module):
    """token_emb (K+1 entries: the last id is the mask token) + learned positions ->
    Decoder (self-attn without causal mask, cross-attn onto the context, GEGLU FFN) -> LN -> logits."""

    def __init__(self, dim, codebook_size, n_heads, d_head, depth, mult, dropout, num_patches):
        super().__init__()
        self.token_emb = nn.Embedding(codebook_size + 1, dim)
        self.pos_enc = nn.Parameter(torch.randn(1, num_patches, dim))
        self.decoder = Decoder(dim=dim, n_heads=n_heads, d_head=d_head, depth=depth, mult=mult, dropout=dropout)
        self.final_norm = LayerNorm(dim)
        self.linear = nn.Linear(dim, codebook_size, bias=False)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(module):
        if isinstance(module, nn.Linear):
            nn.init.trunc_normal_(module.weight, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            nn.init.trunc_normal_(module.weight, std=0.02)

    def forward(self, img_token_indices, context=None, context_mask=None):
        x = self.token_emb(img_token_indices) + self.pos_enc
        x = self.decoder(dec_in=x, context=context, context_mask=context_mask)
        return self.linear(self.final_norm(x))


class MUSE(nn.Module):
    def __init__(self, dim, vq, text_dim=768, n_heads=8, d_head=64, depth=6, mult=4, embeds_drop_prob=0.9, dropout=0.0,
                 fused_sampling=True):
        super().__init__()
        self.fused_sampling = fused_sampling  # generate(): the per-step sampling chain as one kernel
        self.project_embeds = nn.Linear(text_dim, dim)  # text_encoder.project_embeds in the reference
        self.vq = vq
        codebook_size = vq.codebook.codebook_size
        self.mask_token_id = codebook_size
        self.decoder = BidirectionalDecoder(dim, codebook_size, n_heads, d_head, depth, mult, dropout, vq.num_patches)
        self.ignore_index = -1
        self.embeds_drop_prob = embeds_drop_prob
        self.vq.requires_grad_(False)

    def _context(self, text_hidden):
        if not torch.is_tensor(text_hidden):
            raise TypeError("MUSE here takes CLIP's last hidden state (B, 77, 768); the CLIP tower itself "
                            "(CLIPTextModel.from_pretrained) needs a download that is not available")
        return self.project_embeds(text_hidden)

    def fill_mask(self, image_tokens):
        """Cosine schedule: a random timestep per image decides how many tokens become the mask id."""
        B, T = image_tokens.shape
        t = torch.rand(B, device=image_tokens.device)
        n_masked = (T * cosine_schedule(t).clip(0)).round().clamp(min=1)
        order = torch.rand(B, T, device=image_tokens.device).argsort(dim=-1)
        mask = order < n_masked.unsqueeze(-1)
        return image_tokens.masked_fill(mask, self.mask_token_id), image_tokens.masked_fill(~mask, self.ignore_index)

    def forward(self, text_hidden, imgs):
        ctx = self._context(text_hidden)
        with torch.no_grad():
            tokens = self.vq.encode_imgs(imgs)
        inp, tgt = self.fill_mask(tokens)
        keep = torch.rand((ctx.shape[0], 1, 1), device=ctx.device) < self.embeds_drop_prob
        logits = self.decoder(inp, context=ctx * keep)
        return F.cross_entropy(logits.transpose(1, 2), tgt, ignore_index=self.ignore_index)

    @torch.no_grad()
    def generate(self, text_hidden, timesteps=18, gumbel=None):
        ctx = self._context(text_hidden)
        ids = parallel_decode(self.decoder, ctx, self.mask_token_id, self.vq.num_patches, timesteps,
                              fused_sampling=self.fused_sampling, gumbel=gumbel)
        return self.vq.decode_indices(ids)


@torch.no_grad()
def parallel_decode(decoder, ctx, mask_token_id, n, timesteps=18, fused_sampling=True, gumbel=None, trace=None):
    """The masked-token parallel decode of MUSE.generate (models/muse.py:180-236): per step, re-mask the least
    confident tokens (cosine schedule), run the decoder with and without the text context, sample every masked
    token.  gumbel: explicit noise (timesteps, B, n, V) in place of the sampler's own (tests replay the reference's
    draw); trace: a list that receives the ids entering the decoder at every step.  Returns the final ids (B, n)."""
    dev = ctx.device
    B = ctx.shape[0]
    ids = torch.full((B, n), mask_token_id, dtype=torch.long, device=dev)
    scores = torch.zeros(B, n, device=dev)
    mask = torch.zeros(B, n, dtype=torch.bool, device=dev)
    null_ctx = torch.zeros_like(ctx)
    for step, t in enumerate(torch.linspace(0, 1, timesteps, device=dev)):
        steps_until_x0 = timesteps - 1 - step
        n_masked = max(int((cosine_schedule(t) * n).item()), 1)
        low = torch.argsort(scores, dim=-1)[:, :n_masked]          # least confident tokens
        mask.scatter_(1, low, True)
        ids = ids.masked_fill(mask, mask_token_id)
        if trace is not None:
            trace.append(ids.clone())
        logits = decoder(ids, context=ctx)                          # two decoder passes per step:
        null_logits = decoder(ids, context=null_ctx)                # conditional + unconditional
        temperature = 1 * (steps_until_x0 / timesteps)
        noise = gumbel[step] if gumbel is not None else None
        if fused_sampling:
            # CFG (scale 3) + softmax + top-(1-p) filter + Gumbel-argmax + score gather + masked write:
            # one pass over the logits (csrc/sample.hip) instead of the op chain below
            scores = ops.sample_step(logits, ids, mask=mask, null_logits=null_logits, cfg_scale=3.0,
                                     tau=temperature, p=0.9, gumbel=noise)
        else:
            scaled = null_logits + 3 * (logits - null_logits)       # classifier-free guidance, scale 3
            probs = F.softmax(scaled, dim=-1)
            filt = filter_logits(scaled, p=0.9)
            if noise is None:
                pred = F.gumbel_softmax(filt, tau=temperature, hard=False, dim=-1).argmax(dim=-1)
            else:
                pred = F.softmax((filt + noise) / temperature, dim=-1).argmax(dim=-1)
            ids[mask] = pred[mask]
            scores = probs.gather(2, pred.unsqueeze(-1)).squeeze(-1)
        mask = torch.zeros_like(mask)
    return ids
