"""MaskGIT: unconditional masked-token image generator over frozen ViT-VQGAN codes (BASELINE.json
configs[4] names "Muse/MaskGIT decoder"; reference: models/maskgit.py:51-287).  The bidirectional
transformer is ``transformer.Encoder`` -- pre-LN self-attention (no mask) + GEGLU FFN -- on the HIP
attention kernels; the per-step sampling chain of ``generate`` is one kernel (csrc/sample.hip).

The reference class does not run at HEAD: ``fill_mask`` calls ``torch.random(b)``
(a module), ``.cuda()`` is hard-coded, the file imports cv2 and ``generate`` writes debug JPEGs.  What
is built here is what those lines evidently mean: ``torch.rand(b)`` timesteps for the cosine schedule,
tensors on the input's device, no image dumps.  State-dict keys follow the reference
(``bidirectional_transformer.{input_proj,pos_enc,init_norm,decoder,final_norm,linear}``).
Parity: the transformer stack is pinned through its parts (Encoder / SoftmaxAttention / FFN goldens);
the class itself cannot be imported from the reference here (cv2), so its composition is checked
against the oracle's restatement only -- "parity unpinned" for the composition.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .muse import cosine_schedule, filter_logits
from .transformer import Encoder, LayerNorm


class BiDirectionalTransformer(nn.Module):
    """token embedding (vocab + 1: the last id is the mask token) + learned positions -> LN -> Encoder
    -> LN -> logits over the vocabulary (models/maskgit.py:51-91)."""

    def __init__(self, dim, vocab_size=8192, num_patches=256, n_heads=8, d_head=64, dec_depth=6, mult=4, dropout=0.1):
        super().__init__()
        self.input_proj = nn.Embedding(vocab_size + 1, dim)
        self.pos_enc = nn.Parameter(nn.init.trunc_normal_(torch.zeros(1, num_patches, dim), 0.0, 0.02))
        self.mask_token_id = vocab_size
        self.init_norm = LayerNorm(dim)
        self.decoder = Encoder(dim=dim, n_heads=n_heads, d_head=d_head, depth=dec_depth, mult=mult, dropout=dropout)
        self.final_norm = LayerNorm(dim)
        self.linear = nn.Linear(dim, vocab_size, bias=False)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):  # the reference's weights_init: truncated normal (0, 0.02) for Linear and Embedding
        if isinstance(m, (nn.Linear, nn.Embedding)):
            nn.init.trunc_normal_(m.weight.data, 0.0, 0.02)

    def forward(self, x):
        x = self.input_proj(x) + self.pos_enc
        return self.linear(self.final_norm(self.decoder(self.init_norm(x))))


class MaskGitTransformer(nn.Module):
    def __init__(self, dim, vq, vocab_size=8192, n_heads=8, d_head=64, dec_depth=6, mult=4, dropout=0.1,
                 fused_sampling=True):
        super().__init__()
        self.vq = vq
        self.bidirectional_transformer = BiDirectionalTransformer(
            dim=dim, vocab_size=vocab_size, num_patches=vq.num_patches, n_heads=n_heads, d_head=d_head,
            dec_depth=dec_depth, mult=mult, dropout=dropout)
        self.mask_token_id = vocab_size
        self.fused_sampling = fused_sampling
        self.vq.eval()
        self.vq.requires_grad_(False)

    def fill_mask(self, x):
        """Cosine schedule (models/maskgit.py:118-133): a uniform timestep per image decides how many of its
        tokens become the mask id; the loss ignores (-1) the others."""
        b, n = x.shape
        timesteps = torch.rand(b, device=x.device)
        num_masked = (cosine_schedule(timesteps) * n).clamp(min=1.0).int()
        order = torch.rand(x.shape, device=x.device).argsort(dim=-1)
        mask = order < num_masked.unsqueeze(-1)
        return x.masked_fill(mask, self.mask_token_id), x.masked_fill(~mask, -1), mask

    def forward(self, imgs):
        with torch.no_grad():
            x = self.vq.encode_imgs(imgs)
        x, tgt, mask = self.fill_mask(x)
        output = self.bidirectional_transformer(x)
        if not self.training:  # (models/maskgit.py:176-185) fill the masked positions greedily and decode
            pred = output.argmax(dim=-1)
            x = torch.where(mask, pred, x)
            return self.vq.decode_indices(x)
        return F.cross_entropy(output.transpose(1, 2), tgt, ignore_index=-1)

    @torch.no_grad()
    def generate(self, batch=1, timesteps=18, device=None):
        """Confidence-based parallel decode from an all-mask canvas (models/maskgit.py:193-287 without the
        in-painting / JPEG-dump branches): per step re-mask the least confident tokens (cosine schedule),
        one transformer pass, top-(1-p) filtered Gumbel sampling at an annealed temperature."""
        n = self.vq.num_patches
        dev = device or self.bidirectional_transformer.pos_enc.device
        ids = torch.full((batch, n), self.mask_token_id, dtype=torch.long, device=dev)
        scores = torch.zeros(batch, n, device=dev)
        mask = torch.zeros(batch, n, dtype=torch.bool, device=dev)
        for step, t in enumerate(torch.linspace(0, 1, timesteps, device=dev)):
            steps_until_x0 = timesteps - 1 - step
            n_masked = max(int((cosine_schedule(t) * n).item()), 1)
            low = torch.argsort(scores, dim=-1)[:, :n_masked]
            mask.scatter_(1, low, True)
            x = ids.masked_fill(mask, self.mask_token_id)
            logits = self.bidirectional_transformer(x)
            temperature = 1 * (steps_until_x0 / timesteps)
            if self.fused_sampling:
                scores = ops.sample_step(logits, ids, mask=mask, tau=temperature, p=0.9, unmasked_score=1.0)
            else:
                probs = F.softmax(logits, dim=-1)
                pred = F.gumbel_softmax(filter_logits(logits, p=0.9), tau=temperature, hard=False, dim=-1).argmax(dim=-1)
                ids[mask] = pred[mask]
                scores = probs.gather(2, pred.unsqueeze(-1)).squeeze(-1).masked_fill(~mask, 1.0)
            mask = torch.zeros_like(mask)
        return self.vq.decode_indices(ids)
